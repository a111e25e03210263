#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run16; mkdir -p $O
AFIS_AB_OPTS=bound_cus=0 timeout 600 python tools/lib_ab.py 20000 20 tools/exp/libafis_rot1.so tools/exp/libafis_rot2.so tools/exp/libafis_rot3.so > $O/ab.txt 2>&1; cut -c1-30,140-210,330-420 $O/ab.txt
AFIS_AB_OPTS=bound_cus=128 timeout 600 python tools/lib_ab.py 50000 20 tools/exp/libafis_rot1.so tools/exp/libafis_rot2.so tools/exp/libafis_rot3.so > $O/ab128.txt 2>&1; cut -c1-30,140-210,330-420 $O/ab128.txt
