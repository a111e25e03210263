#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "outlasts or properties" > $O/t.log 2>&1; echo "pytest rc $?" >> $O/t.log; tail -5 $O/t.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "correspond or all_templates or c_abi or cli" > $O/t2.log 2>&1; echo "pytest rc $?" >> $O/t2.log; tail -3 $O/t2.log
