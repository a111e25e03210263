#!/bin/bash
# FETCH_SIZE on an 8-byte-per-lane coalesced stream (the recomputation kernel's record stream), against the 16-byte stream of round 3's calibration
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r05_run11; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in 0 2; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$mode -- $GRAFT_REPO_ROOT/tools/ubench/fetch_calib $mode 8192 > $O/known_$mode.json 2> $O/err_$mode.txt
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py "$O/pmc_$mode/**/*counter_collection.csv" > $O/pmc_$mode.txt; cat $O/known_$mode.json $O/pmc_$mode.txt
done
rm -rf $O/pmc_0 $O/pmc_2
