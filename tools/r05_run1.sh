#!/bin/bash
# round 5, first GPU run: parity of the shape-class candidate kernel, headline and wide bench, old-vs-new library on the wide shapes
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run1; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -5 $O/gpu_tests.log
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_headline.json 2> $O/bench_headline.err; echo "headline rc $?"
timeout 900 python bench.py --workload wide --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_wide.json 2> $O/bench_wide.err; echo "wide rc $?"
AFIS_AB_WORKLOAD=wide timeout 900 python tools/lib_ab.py 20000 8 tools/exp/libafis_r04.so > $O/ab_wide.txt 2>&1; echo "ab wide rc $?"
timeout 600 python tools/lib_ab.py 20000 8 tools/exp/libafis_r04.so > $O/ab_headline.txt 2>&1; echo "ab headline rc $?"
cat $O/ab_wide.txt $O/ab_headline.txt
python - <<'PY'
import json
for n in ("headline", "wide"):
    try:
        d = json.loads(open(f"gpurun_out/r05_run1/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["stage_ms_per_step"], d["minutiae_candidate_tasks"], d["roofline"].get("measured_clock_ghz"), d["roofline"]["frac"], d["rank1_hits"])
    except Exception as e:
        print(n, "failed", e)
PY
