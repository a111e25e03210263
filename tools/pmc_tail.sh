#!/bin/bash
# SQ counters of the non-ADC kernels (candidate + graph kernels) on a 10k x 8 probe workload, counters only (two passes):
#   bash tools/pmc_tail.sh <tag>
TAG=${1:-pmc_tail}
OUT=$PWD/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --gallery 10000 --queries 8 --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/a -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/b -- $B > /dev/null 2>&1
find $OUT -name "*kernel_trace.csv" -delete
python $REPO/tools/pmc_summary.py "$OUT/**/*counter_collection.csv" > $OUT/summary.txt
grep -v "adc_rowmax\|lut_build\|k_fuse\|rocclr" $OUT/summary.txt | head -80
