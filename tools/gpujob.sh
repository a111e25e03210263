#!/bin/bash
# One parameterised job script for the GPU box (replaces the per-experiment r05_run*.sh / *_sweep*.sh files of earlier rounds: their outputs are in profiles/).
#   bash tools/gpujob.sh bench   <tag> [bench args]          one bench.py line            -> gpurun_out/<tag>.json
#   bash tools/gpujob.sh sweep   <tag> <option> "<values>" [bench args]   bench.py once per value of a bench option (e.g. --bound-cus "0 96 128 160", --query-batch "20 34 50")
#   bash tools/gpujob.sh kstats  <tag> [bench args]          per-kernel times under rocprofv3 --kernel-trace --stats -> gpurun_out/<tag>/kernel_stats.csv
#   bash tools/gpujob.sh pmc     <tag> "<counters>" [bench args]          one counter pass (counters only: never with --sys-trace etc.) -> gpurun_out/<tag>/pmc_summary.txt
#   bash tools/gpujob.sh ab      <tag> <lib A> <lib B> [lib_ab.py args]    interleaved A/B of two builds of the library on one box (tools/lib_ab.py)
#   bash tools/gpujob.sh parity  <tag> <seed> <Q> <G> [env assignments]   tools/parity_sweep.py (AFIS_SWEEP_WORKLOAD=wide|structured, AFIS_SWEEP_DUP=0|10|30)
# Run from the repo root through gpurun, e.g.  gpurun --timeout 900 -- 'bash tools/gpujob.sh sweep cus --bound-cus "0 96 128" --workload structured'
set -u
MODE=${1:?mode}; TAG=${2:?tag}; shift 2
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
line() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(d["value"], d["ms_per_step"], d["stage_ms_per_step"], d.get("stage_ms_per_step_back_to_back"))
PY
}
case $MODE in
  bench)  python bench.py --no-cpu-baseline "$@" > $OUT/$TAG.json 2> $OUT/$TAG.err; line $OUT/$TAG.json ;;
  sweep)  OPT=$1; VALS=$2; shift 2
          for v in $VALS; do python bench.py --no-cpu-baseline --no-alone $OPT $v "$@" > $OUT/${TAG}_$v.json 2> $OUT/${TAG}_$v.err; echo -n "$OPT $v: "; line $OUT/${TAG}_$v.json; done ;;
  kstats) mkdir -p $OUT/$TAG; cd /tmp; export TMPDIR=/tmp
          rocprofv3 --kernel-trace --stats -d $OUT/$TAG/stats -o stats -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alone "$@" > $OUT/$TAG/bench_profiled.json 2> $OUT/$TAG/bench_profiled.err
          python $REPO/tools/rocprof_summary.py $(find $OUT/$TAG/stats -name "*.db" | head -1) $OUT/$TAG/kernel_stats.csv; rm -rf $OUT/$TAG/stats
          cut -d, -f1-7,10-14 $OUT/$TAG/kernel_stats.csv | cut -c1-220 | head -12 ;;
  pmc)    CTRS=$1; shift; mkdir -p $OUT/$TAG; cd /tmp; export TMPDIR=/tmp
          rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/$TAG/pmc -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --bound-cus 0 "$@" > /dev/null 2>&1
          python $REPO/tools/pmc_summary.py "$OUT/$TAG/pmc/**/*counter_collection.csv" > $OUT/$TAG/pmc_summary.txt; rm -rf $OUT/$TAG/pmc; head -30 $OUT/$TAG/pmc_summary.txt ;;
  ab)     python tools/lib_ab.py "$@" > $OUT/$TAG.txt 2>&1; tail -20 $OUT/$TAG.txt ;;
  parity) SEED=$1; Q=$2; G=$3; shift 3; env "$@" python tools/parity_sweep.py $SEED $Q $G > $OUT/$TAG.txt 2>&1; tail -4 $OUT/$TAG.txt ;;
  *) echo "unknown mode $MODE"; exit 2 ;;
esac
