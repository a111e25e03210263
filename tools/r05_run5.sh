#!/bin/bash
# round 5, fifth GPU run: the large class with a second key block (parity, wide bench), phase shares of the candidate kernel, new-vs-round-4 library on the headline shapes
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "off_envelope or shape_classes or edge_shapes or stage_lists or c_abi or container" > $O/gpu_tests_sel.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests_sel.log; tail -6 $O/gpu_tests_sel.log
timeout 300 python tools/phase_probe.py > $O/phase_probe.txt 2>&1; head -12 $O/phase_probe.txt
AFIS_AB_OPTS=bound_cus=0 timeout 600 python tools/lib_ab.py 20000 20 tools/exp/libafis_r04.so > $O/ab_headline_b2b.txt 2>&1; cat $O/ab_headline_b2b.txt
timeout 900 python bench.py --workload wide --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_wide.json 2> $O/bench_wide.err; echo "wide rc $?"
timeout 1500 python tools/offenv_sweep.py 2 80 250 $O/offenv_sweep_seed2.json > $O/offenv_sweep_seed2.log 2>&1; echo "offenv rc $?"; tail -2 $O/offenv_sweep_seed2.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_run5/bench_wide.json").read().strip().splitlines()[-1])
print("wide", d["value"], d["ms_per_step"], d["stage_ms_per_step"], d["minutiae_candidate_tasks"], d["rank1_hits"])
PY
