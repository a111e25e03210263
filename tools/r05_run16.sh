#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run16; mkdir -p $O
AFIS_AB_OPTS=bound_cus=0 timeout 600 python tools/lib_ab.py 20000 20 tools/exp/libafis_mc3.so > $O/ab.txt 2>&1; cat $O/ab.txt | cut -c1-330
