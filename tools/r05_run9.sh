#!/bin/bash
# recomputation kernel with four lanes per item: parity subset, then new vs round 4 on the headline shapes (kernels back to back)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rowmax or bound or scores_small or matrix_core or texture" > $O/gpu_tests_sel.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests_sel.log; tail -6 $O/gpu_tests_sel.log
AFIS_AB_OPTS=bound_cus=0 timeout 600 python tools/lib_ab.py 20000 20 tools/exp/libafis_r04.so > $O/ab_headline_b2b.txt 2>&1; cat $O/ab_headline_b2b.txt
