#!/bin/bash
# `match -ldir` under rocprofv3 --kernel-trace, n times: which kernel of which search call is slow in the slow runs (tools/ldir_repeat.sh found the search
# stage bimodal: one call's minutiae candidates 818 ms instead of 149)
# usage: tools/ldir_trace.sh <work dir of tools/cli_scale_r04.py run with AFIS_CLI_KEEP=1> <n> <out dir under gpurun_out>
W=$1; N=${2:-4}; O=$3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in $(seq 1 $N); do
  mkdir -p $W/out_$i $R/$O/run_$i
  AFIS_MATCH_TIMING=2 rocprofv3 --kernel-trace --output-format csv -d $R/$O/run_$i -- $R/msu-latentafis_amd/csrc/match -ldir $W/lat -g $W/gallery.afisgal -s $W/out_$i/ -c $R/tests/golden/codebook_EmbeddingSize_96_stride_16_subdim_6.dat -d 0 2> $R/$O/run_$i/stderr.log >/dev/null
  grep -E "of search" $R/$O/run_$i/stderr.log
  rm -rf $W/out_$i
done
