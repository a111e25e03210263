#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run15; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['rank1_hits'], d['cpu_baseline']['value'], d['roofline']['frac'], d['roofline']['alone_on_the_chip'])"
timeout 400 python bench.py --workload wide --no-cpu-baseline > $O/bench_wide.json 2> $O/bench_wide.err; echo "wide rc $?"; python -c "
import json; d=json.loads(open('$O/bench_wide.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
