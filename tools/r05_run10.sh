#!/bin/bash
# bound pass: timing-only ablations of the round's three ideas (wrong results by construction; one box, libraries interleaved, kernels back to back)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run10; mkdir -p $O
AFIS_ABLATE_SKIP_TEXTURE_TAIL=1 AFIS_AB_OPTS=bound_cus=0 timeout 600 python tools/lib_ab.py 50000 20 tools/exp/libafis_mf1.so tools/exp/libafis_mf5.so tools/exp/libafis_mf10.so tools/exp/libafis_mf11.so > $O/ablate_b2b.txt 2>&1; cat $O/ablate_b2b.txt
AFIS_ABLATE_SKIP_TEXTURE_TAIL=1 AFIS_AB_OPTS=bound_cus=128 timeout 600 python tools/lib_ab.py 50000 20 tools/exp/libafis_mf1.so tools/exp/libafis_mf10.so tools/exp/libafis_mf11.so > $O/ablate_128.txt 2>&1; cat $O/ablate_128.txt
