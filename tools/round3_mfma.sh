#!/bin/bash
# adc_variant 9 bring-up: MFMA lane-map check, parity tests, bench with the refine counters, timed bench, kernel stats.
TAG=${1:-r03b}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
timeout 120 tools/ubench/mfma_layout > $OUT/mfma_layout.json 2>&1; cat $OUT/mfma_layout.json
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "${2:-rowmax_bit_exact or scores_small or bound_and_refine or matrix_core}" > $OUT/parity.log 2>&1; tail -15 $OUT/parity.log
timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --variant 9 --refine-stats --gallery 20000 --queries 16 > $OUT/bench_v9_stats.json 2> $OUT/bench_v9_stats.err; tail -c 1200 $OUT/bench_v9_stats.json; tail -3 $OUT/bench_v9_stats.err
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --variant 9 > $OUT/bench_v9.json 2> $OUT/bench_v9.err; tail -c 700 $OUT/bench_v9.json; tail -3 $OUT/bench_v9.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --variant 9 > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
python $REPO/tools/rocprof_summary.py $(find $OUT/stats -name "*.db" | head -1) $OUT/kernel_stats.csv
find $OUT/stats -type f -delete 2>/dev/null
cut -d, -f1-7,10-14 $OUT/kernel_stats.csv | head -14
