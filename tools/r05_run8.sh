#!/bin/bash
# CU split of the overlapped schedule at the faster candidate kernel
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run8; mkdir -p $O
for bc in 128 160 192 128 160; do timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alone --bound-cus $bc > $O/b_$bc.json 2>> $O/err.txt; python -c "
import json; d=json.loads(open('$O/b_$bc.json').read().strip().splitlines()[-1]); print($bc, d['value'], d['ms_per_step'], {k:d['stage_ms_per_step'][k] for k in ('adc_bound_ms','adc_refine_ms','tex_tail_ms','cands_ms','minu_graph_ms')})"; done
