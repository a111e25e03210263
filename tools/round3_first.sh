#!/bin/bash
# Round 3, first GPU call: the whole -m gpu suite (new: virtual-shard configs[3]/[4], driver-form bench, un-normalised descriptors),
# the VALU-issue micro-benchmark, the FETCH_SIZE calibration, the default bench with the CPU thread curve, the per-rank shard workloads.
TAG=${1:-r03a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_tests.log
tail -3 $OUT/gpu_tests.log
timeout 300 tools/ubench/valu_peak > $OUT/valu_peak.json 2> $OUT/valu_peak.err
timeout 600 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
for g in 12500 25000 50000; do
  timeout 600 python bench.py --steps 3 --warmup 1 --gallery $g --no-cpu-baseline > $OUT/bench_shard_$g.json 2> $OUT/bench_shard_$g.err
done
cd /tmp && export TMPDIR=/tmp
C=$REPO/tools/ubench/fetch_calib
for spec in "0 8192 0 stream" "1 32768 268435456 gather_32g" "1 86 1073741824 gather_86m"; do
  set -- $spec
  timeout 300 $C $1 $2 $3 > $OUT/calib_$4.json 2>> $OUT/calib.err
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_$4_fetch -- $C $1 $2 $3 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/calib_$4_rdreq -- $C $1 $2 $3 > /dev/null 2>&1
  python $REPO/tools/pmc_summary.py "$OUT/calib_$4_fetch/**/*counter_collection.csv" "$OUT/calib_$4_rdreq/**/*counter_collection.csv" > $OUT/calib_$4_pmc.txt
  rm -rf $OUT/calib_$4_fetch $OUT/calib_$4_rdreq
done
cd $REPO
tail -c 1500 $OUT/bench.json; cat $OUT/calib_*_pmc.txt | head -40
