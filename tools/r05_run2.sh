#!/bin/bash
# round 5, second GPU run: the reproducer of the side-stream wait, the new parity tests, the off-envelope sweep, wide bench with the adaptive schedule
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run2; mkdir -p $O
bash tools/repro/run.sh > $O/side_stream_hang.txt 2>&1; cat $O/side_stream_hang.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "off_envelope or shape_classes or edge_shapes or stage_lists or schedule or allocates" > $O/gpu_tests_new.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests_new.log; tail -15 $O/gpu_tests_new.log
timeout 1500 python tools/offenv_sweep.py 1 80 250 $O/offenv_sweep_seed1.json > $O/offenv_sweep_seed1.log 2>&1; echo "offenv rc $?"; tail -3 $O/offenv_sweep_seed1.log
timeout 900 python bench.py --workload wide --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_wide.json 2> $O/bench_wide.err; echo "wide rc $?"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_headline.json 2> $O/bench_headline.err; echo "headline rc $?"
python - <<'PY'
import json
for n in ("headline", "wide"):
    try:
        d = json.loads(open(f"gpurun_out/r05_run2/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["config"]["schedule"][:60], d["stage_ms_per_step"], d["minutiae_candidate_tasks"]["fallback_share"], d["ranks"], d["distinct_devices"], d["rccl_ranks"])
    except Exception as e:
        print(n, "failed", e)
PY
