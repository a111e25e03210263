#!/bin/bash
# bound pass A/B: libafis_hip.so (working tree) against tools/exp variants, alone on the chip and on 128 CUs; parity subset first
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab_bound; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rowmax or bound or matrix_core or scores_small" > $O/t.log 2>&1; echo "pytest rc $?" >> $O/t.log; tail -3 $O/t.log
AFIS_AB_OPTS=bound_cus=0 timeout 600 python tools/lib_ab.py 50000 20 $(ls tools/exp/libafis_*.so) > $O/ab_alone.txt 2>&1; cut -c1-30,120-215 $O/ab_alone.txt
AFIS_AB_OPTS=bound_cus=128 timeout 600 python tools/lib_ab.py 50000 20 $(ls tools/exp/libafis_*.so) > $O/ab_128.txt 2>&1; cut -c1-30,120-215 $O/ab_128.txt
