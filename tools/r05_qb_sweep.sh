#!/bin/bash
# latents per launch group in the default (overlapped) schedule: bench.py --query-batch N, one box, in this order, twice
cd "$(dirname "$0")/.."
O=gpurun_out/r05_qb_sweep; mkdir -p $O
for pass in 1 2; do
for qb in ${QBS:-0 12 17 25 34 50}; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alone --query-batch $qb > $O/qb${qb}_p$pass.json 2> $O/qb${qb}_p$pass.err
  python - <<PY
import json
d=json.loads(open('$O/qb${qb}_p$pass.json').read().strip().splitlines()[-1])
print($qb, $pass, d['value'], d['ms_per_step'], {k:v for k,v in d['stage_ms_per_step'].items() if k in ('adc_bound_ms','adc_refine_ms','tex_tail_ms','cands_ms','minu_graph_ms')})
PY
done; done | tee $O/summary.txt
