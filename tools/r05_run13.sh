#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run13; mkdir -p $O
export AFIS_DESTROY_TRACE=1
EARLY_CLOSE=1 TAPS=1 timeout 200 python tools/repro/timeout_recovery.py > $O/t2.log 2>&1; echo "rc $?" >> $O/t2.log; cat $O/t2.log
timeout 300 python tools/repro/timeout_recovery.py > $O/t1.log 2>&1; echo "rc $?" >> $O/t1.log; cat $O/t1.log
