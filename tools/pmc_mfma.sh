#!/bin/bash
# SQ counters of adc_variant 9's kernels: bash tools/pmc_mfma.sh <tag> [bench args]
TAG=${1:-pmc9}; shift
OUT=$PWD/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --gallery 25000 --queries 16 --steps 1 --warmup 0 --no-cpu-baseline --variant 9 $@"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/a -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/b -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/c -- $B > /dev/null 2>&1
python $REPO/tools/pmc_summary.py "$OUT/a/**/*counter_collection.csv" "$OUT/b/**/*counter_collection.csv" "$OUT/c/**/*counter_collection.csv" > $OUT/pmc_summary.txt
python - <<PY
import csv,glob
t={}
for f in glob.glob("$OUT/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][:40]; t.setdefault(k,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
for k,v in t.items(): print(k, len(v), "avg ms %.3f"%(sum(v)/len(v)))
PY
rm -rf $OUT/a $OUT/b $OUT/c
grep -A26 "k_adc_mfma\|k_tex_refine" $OUT/pmc_summary.txt | head -70
