#!/bin/bash
# one box: the bound pass's gallery chunk and its CU share at the round's launch-group size (default schedule, 100 x 100k)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_misc_sweep; mkdir -p $O
run() { tag=$1; shift; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alone "$@" > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1])
print('$tag', d['value'], d['ms_per_step'], {k:v for k,v in d['stage_ms_per_step'].items() if k in ('adc_bound_ms','adc_refine_ms','tex_tail_ms','cands_ms','minu_graph_ms')})
PY
}
for pass in 1 2; do
run default_p$pass
run chunk360_p$pass --chunk 360
run chunk1440_p$pass --chunk 1440
run cus96_p$pass --bound-cus 96
run cus160_p$pass --bound-cus 160
done | tee $O/summary.txt
