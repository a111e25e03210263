#!/bin/bash
# candidate kernel: batched key pass, 8-wide ranking, carry-chain hit mask, split tickets — parity subset, then new vs round 4 on the headline shapes (kernels back to back), phase shares
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "off_envelope or shape_classes or edge_shapes or stage_lists or degenerate or scores_small" > $O/gpu_tests_sel.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests_sel.log; tail -6 $O/gpu_tests_sel.log
AFIS_AB_OPTS=bound_cus=0 timeout 600 python tools/lib_ab.py 20000 20 tools/exp/libafis_r04.so > $O/ab_headline_b2b.txt 2>&1; cat $O/ab_headline_b2b.txt
bash tools/build_phase_lib.sh > /dev/null 2>&1; timeout 300 python tools/phase_probe.py > $O/phase_probe.txt 2>&1; head -12 $O/phase_probe.txt
AFIS_AB_WORKLOAD=wide AFIS_AB_OPTS=bound_cus=0 timeout 600 python tools/lib_ab.py 20000 8 > $O/ab_wide.txt 2>&1; cat $O/ab_wide.txt
