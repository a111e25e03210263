#!/bin/bash
# Round-6 profile bundle, run on the GPU box from the repo root:  bash tools/profile_round6.sh <tag> [big]
# = round 5's bundle (bench, the same command under rocprofv3 --kernel-trace --stats, PMC passes — counters only, separate runs —, refine statistics, the wide workload, the per-rank
# shards, single latents) + the STRUCTURED workload at three shares of repeated code vectors, and its per-kernel times.   Summarised by: python tools/collect_profiles_r05.py <tag> r06
TAG=${1:-r06}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --bound-cus 0 > $OUT/bench_back_to_back.json 2> $OUT/bench_back_to_back.err
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --refine-stats > $OUT/bench_refine_stats.json 2> $OUT/bench_refine_stats.err
for d in 10 0 30; do python bench.py --workload structured --dup $d --steps 3 --warmup 1 --no-cpu-baseline --refine-stats > $OUT/bench_structured_dup$d.json 2> $OUT/bench_structured_dup$d.err; done
python bench.py --workload wide --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_wide.json 2> $OUT/bench_wide.err
for g in 12500 25000 50000; do python bench.py --steps 3 --warmup 1 --gallery $g --no-cpu-baseline > $OUT/bench_shard_$g.json 2> $OUT/bench_shard_$g.err; done
python bench.py --queries 1 --gallery 10000 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/latency_1x10k.json 2> $OUT/latency_1x10k.err
python bench.py --queries 1 --gallery 100000 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/latency_1x100k.json 2> $OUT/latency_1x100k.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alone > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
python $REPO/tools/rocprof_summary.py $(find $OUT/stats -name "*.db" | head -1) $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $OUT/stats0 -o stats -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --bound-cus 0 > $OUT/bench_profiled_b2b.json 2> $OUT/bench_profiled_b2b.err
python $REPO/tools/rocprof_summary.py $(find $OUT/stats0 -name "*.db" | head -1) $OUT/kernel_stats_back_to_back.csv
rocprofv3 --kernel-trace --stats -d $OUT/statss -o stats -- python $REPO/bench.py --workload structured --dup 10 --steps 2 --warmup 1 --no-cpu-baseline --bound-cus 0 > $OUT/bench_profiled_structured.json 2> $OUT/bench_profiled_structured.err
python $REPO/tools/rocprof_summary.py $(find $OUT/statss -name "*.db" | head -1) $OUT/kernel_stats_structured.csv
# the counter passes characterise each kernel ALONE (one stream, kernels back to back): what runs beside a kernel in the default schedule would be counted into it
B="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --bound-cus 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq3 -- $B > /dev/null 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT/stats -type f ! -name "*.db" -delete 2>/dev/null
python $REPO/tools/pmc_summary.py "$OUT/pmc_fetch/**/*counter_collection.csv" "$OUT/pmc_write/**/*counter_collection.csv" > $OUT/pmc_hbm_summary.txt
python $REPO/tools/pmc_summary.py "$OUT/pmc_sq1/**/*counter_collection.csv" "$OUT/pmc_sq2/**/*counter_collection.csv" "$OUT/pmc_sq3/**/*counter_collection.csv" > $OUT/pmc_sq_summary.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3 $OUT/stats $OUT/stats0 $OUT/statss
cd $REPO
if [ "$2" = "big" ]; then python bench.py --queries 8 --gallery 1000000 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_8x1M.json 2> $OUT/bench_8x1M.err; fi
ls $OUT; tail -c 1500 $OUT/bench.json
