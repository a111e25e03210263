#!/bin/bash
# Round profile bundle, run on the GPU box from the repo root:  bash tools/profile_round.sh <tag>
# 1. plain bench (the JSON line)          2. the SAME command under rocprofv3 --kernel-trace --stats
# 3./4. PMC passes for HBM traffic of the ADC kernel (FETCH_SIZE and WRITE_SIZE in separate passes, counters only)
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
python bench.py --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $OLDPWD/bench.py --steps 2 --warmup 1 > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
B="python $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.json 2> /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.json 2> /dev/null
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.json 2> /dev/null
# keep the merged output small: drop raw traces, keep the counter CSVs and the stats db
find $OUT -name "*kernel_trace.csv" -delete
ls -R $OUT | head -40
tail -c 600 $OUT/bench.json
