#!/bin/bash
# launch-group size sweep on the bench workload: bash tools/qb_sweep2.sh "0 7 15 20 34"
for qb in $1; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --query-batch $qb 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('query_batch', $qb, j['value'], j['ms_per_step'], {k:round(v) for k,v in j['stage_ms_per_step'].items()})"
done
