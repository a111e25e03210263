#!/bin/bash
# bound-pass chunk size sweep on the bench workload: bash tools/chunk_sweep2.sh "0 150 100 600"
for c in $1; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --chunk $c 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunk', $c, j['value'], j['ms_per_step'], {k:round(v,1) for k,v in j['stage_ms_per_step'].items() if k.startswith('adc')})"
done
