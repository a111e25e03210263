#!/bin/bash
# the whole GPU suite at the end-of-round code, log kept for profiles/
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run12; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -25 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
