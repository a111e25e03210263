#!/bin/bash
# full GPU suite + headline / wide bench at the current kernels
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run7; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_headline.json 2> $O/bench_headline.err; echo "headline rc $?"
timeout 900 python bench.py --workload wide --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_wide.json 2> $O/bench_wide.err; echo "wide rc $?"
python - <<'PY'
import json
for n in ("headline", "wide"):
    d = json.loads(open(f"gpurun_out/r05_run7/bench_{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["stage_ms_per_step"], d["stage_ms_per_step_back_to_back"], d["rank1_hits"])
PY
