#!/bin/bash
# round 5, third GPU run: the whole GPU suite on the re-cut libraries, the side-stream wait experiments inside the real search, per-kernel stats of the wide workload
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run3; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
# (a) no per-group wait, final wait polls the context's stream only: round 4's hang, now bounded
AFIS_NO_GROUP_WAIT=1 AFIS_SEARCH_TIMEOUT_S=25 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alone > $O/wait_a.json 2> $O/wait_a.err; echo "wait_a rc $?"; tail -2 $O/wait_a.err
# (b) no per-group wait, final wait polls all three streams
AFIS_NO_GROUP_WAIT=1 AFIS_FINAL_WAIT=1 AFIS_SEARCH_TIMEOUT_S=25 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alone > $O/wait_b.json 2> $O/wait_b.err; echo "wait_b rc $?"; tail -2 $O/wait_b.err
# (c) shipped: per-group wait on the side streams
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alone > $O/wait_c.json 2> $O/wait_c.err; echo "wait_c rc $?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_wide -o wide -- python $GRAFT_REPO_ROOT/bench.py --workload wide --steps 2 --warmup 1 --no-cpu-baseline --no-alone > $GRAFT_REPO_ROOT/$O/prof_wide.json 2> $GRAFT_REPO_ROOT/$O/prof_wide.err; echo "prof wide rc $?"
cd $GRAFT_REPO_ROOT
find $O/prof_wide -name "*kernel_stats*" | head; f=$(find $O/prof_wide -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/wide_kernel_stats.csv && head -20 $O/wide_kernel_stats.csv
find $O/prof_wide -name "*.db" -delete; find $O/prof_wide -name "*trace.csv" -size +20M -delete
python - <<'PY'
import json
for n in ("wait_a", "wait_b", "wait_c"):
    try:
        d = json.loads(open(f"gpurun_out/r05_run3/{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["rank1_hits"])
    except Exception as e:
        print(n, "failed", e)
PY
