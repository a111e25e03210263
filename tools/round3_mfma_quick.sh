#!/bin/bash
# quick A/B of adc_variant 9: parity subset, timed bench, per-kernel stats.  bash tools/round3_mfma_quick.sh <tag> [bench args]
TAG=${1:-r03q}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rowmax_bit_exact or scores_small or bound_and_refine or matrix_core" > $OUT/parity.log 2>&1; tail -4 $OUT/parity.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --variant 9 "$@" > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
python $REPO/tools/rocprof_summary.py $(find $OUT/stats -name "*.db" | head -1) $OUT/kernel_stats.csv
find $OUT/stats -type f -delete 2>/dev/null
python - <<PY
import csv, json
for r in csv.DictReader(open('$OUT/kernel_stats.csv')):
    if float(r['total_ms']) > 1: print(r['kernel'][:36].ljust(36), r['calls'], r['total_ms'], r['avg_ms'], r['min_ms'], r['max_ms'], r['pct'])
j = json.loads(open('$OUT/bench_profiled.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['stage_ms_per_step'], j['rank1_hits'])
PY
