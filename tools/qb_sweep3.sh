#!/bin/bash
# launch-group size sweep on a shard: bash tools/qb_sweep3.sh <gallery> "<batches>"
for qb in $2; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gallery $1 --query-batch $qb 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gallery', $1, 'query_batch', $qb, j['value'], j['ms_per_step'], j['rank1_hits'], {k:round(v,1) for k,v in j['stage_ms_per_step'].items()})"
done
