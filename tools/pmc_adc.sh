#!/bin/bash
# SQ counters of the ADC kernel for a given variant on the small probe workload: bash tools/pmc_adc.sh <variant>
V=${1:-1}
OUT=$PWD/gpurun_out/pmc_adc_v$V
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $OLDPWD/bench.py --gallery 10000 --queries 4 --steps 1 --warmup 0 --no-cpu-baseline --variant $V"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/a -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b -- $B > /dev/null 2>&1
find $OUT -name "*kernel_trace.csv" -delete
