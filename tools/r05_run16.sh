#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run16; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
bash tools/profile_round5.sh r05_bundle5 > $O/bundle.log 2>&1; echo "bundle rc $?"
python -c "
import json; d=json.loads(open('gpurun_out/r05_bundle5/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['rank1_hits'], d['roofline']['frac'], d['roofline']['alone_on_the_chip']['frac'])"
