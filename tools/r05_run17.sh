#!/bin/bash
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r05_run17; mkdir -p $O
REPO=$PWD
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --bound-cus 0 > $O/bench_b2b.json 2> $O/bench_b2b.err
python - <<PY
import json
for t in ('bench','bench_b2b'):
    d=json.loads(open('$O/'+t+'.json').read().strip().splitlines()[-1]); print(t, d['value'], d['ms_per_step'], d['rank1_hits'], d['stage_ms_per_step'])
PY
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alone --bound-cus 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- $B > /dev/null 2>&1
python $REPO/tools/pmc_summary.py "$O/pmc_f/**/*counter_collection.csv" "$O/pmc_w/**/*counter_collection.csv" 2>&1 | grep -A2 "adc_mfma\|tex_refine\|graph_texture" | tee $O/pmc.txt
rm -rf $O/pmc_f $O/pmc_w
