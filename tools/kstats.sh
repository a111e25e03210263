#!/bin/bash
# per-kernel times of one bench step under rocprofv3 (kernel trace + stats only):  bash tools/kstats.sh <tag> [bench args]
TAG=${1:-k}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
DB=$(find $OUT/stats -name "*.db" | head -1)
python $REPO/tools/rocprof_summary.py $DB $OUT/kernel_stats.csv
find $OUT/stats -type f ! -name "*.db" -delete 2>/dev/null
cut -d, -f1-7,10-14 $OUT/kernel_stats.csv | head -12
