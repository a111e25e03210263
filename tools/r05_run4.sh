#!/bin/bash
# round 5, fourth GPU run: per-kernel stats of the wide workload; where the small-class candidate kernel's time goes (occupancy probe AFIS_RT_GRID, barrier / GEMM-only ablations,
# kernels back to back); round 4's hanging wait form under a kill timer, for the record
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run4; mkdir -p $O
bash tools/kstats.sh r05_run4/wide --workload wide --bound-cus 0 > $O/kstats_wide.txt 2>&1; tail -14 $O/kstats_wide.txt
bash tools/kstats.sh r05_run4/headline --bound-cus 0 > $O/kstats_headline.txt 2>&1; tail -14 $O/kstats_headline.txt
export AFIS_AB_OPTS=bound_cus=0
for g in 256 512 768 1024 2048; do echo "grid $g"; AFIS_RT_GRID=$g timeout 300 python tools/lib_ab.py 20000 20 2>&1 | tail -1; done > $O/occupancy.txt 2>&1
cat $O/occupancy.txt
timeout 600 python tools/lib_ab.py 20000 20 tools/exp/libafis_mc1.so tools/exp/libafis_mc2.so > $O/ablate.txt 2>&1; cat $O/ablate.txt
unset AFIS_AB_OPTS
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alone > $O/bench_shipped_wait.json 2> $O/bench_shipped_wait.err; echo "shipped rc $?"
AFIS_SEARCH_TIMEOUT_S=0 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alone > $O/bench_blocking_all.json 2> $O/bench_blocking_all.err; echo "blocking waits on all three streams rc $?"
AFIS_SEARCH_TIMEOUT_S=0 AFIS_WAIT_CTX_SYNC_ONLY=1 timeout -s KILL 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alone > $O/bench_ctx_sync_only.json 2> $O/bench_ctx_sync_only.err; echo "hipStreamSynchronize of the context's stream only rc $? (137 = killed after 150 s: the hang)"
python - <<'PY'
import json
for n in ("bench_shipped_wait", "bench_blocking_all", "bench_ctx_sync_only"):
    try:
        d = json.loads(open(f"gpurun_out/r05_run4/{n}.json").read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], d["rank1_hits"])
    except Exception as e:
        print(n, "no line:", e)
PY
