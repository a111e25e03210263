#!/bin/bash
# `match -ldir` on one container several times in a row: is the search stage's wall time stable, and is it the device's time?
# usage: tools/ldir_repeat.sh <work dir of tools/cli_scale_r04.py run with AFIS_CLI_KEEP=1> <n> [env assignments for match ...]
W=$1; N=${2:-5}; shift; shift
for i in $(seq 1 $N); do
  mkdir -p $W/out_$i
  env AFIS_MATCH_TIMING=2 "$@" msu-latentafis_amd/csrc/match -ldir $W/lat -g $W/gallery.afisgal -s $W/out_$i/ -c tests/golden/codebook_EmbeddingSize_96_stride_16_subdim_6.dat -d 0 2>&1 >/dev/null | grep -E "timing|device|alloc"
  rm -rf $W/out_$i
done
