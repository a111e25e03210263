#!/bin/bash
# bound pass: the row groups of a gallery chunk on ONE XCD (default) against round-robin chunks (AFIS_MF_NO_XCD_MAP=1): parity subset, step time, FETCH_SIZE of the pass
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r05_xcd_map; mkdir -p $O
REPO=$PWD
#timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rowmax or bound or matrix_core or scores_small or launch_groups or schedule" > $O/t.log 2>&1; echo "pytest rc $?" >> $O/t.log; tail -3 $O/t.log
run() { tag=$1; shift; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1])
print('$tag', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['alone_on_the_chip']['avg_launch_ms'], {k:v for k,v in d['stage_ms_per_step'].items() if k in ('adc_bound_ms','adc_refine_ms','tex_tail_ms','cands_ms','minu_graph_ms')})
PY
}
for pass in; do
run xcd_p$pass
AFIS_MF_NO_XCD_MAP=1 run rr_p$pass
done | tee $O/summary.txt
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alone --bound-cus 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_xcd -- $B > /dev/null 2>&1
AFIS_MF_NO_XCD_MAP=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_rr -- $B > /dev/null 2>&1
B="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alone"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_xcd128 -- $B > /dev/null 2>&1
for t in xcd rr xcd128; do echo "== $t"; python $REPO/tools/pmc_summary.py "$O/pmc_$t/**/*counter_collection.csv" 2>&1 | grep -A1 "adc_mfma\|tex_refine"; done | tee $O/pmc.txt
rm -rf $O/pmc_xcd $O/pmc_rr $O/pmc_xcd128
