#!/bin/bash
# A/B builds: tools/build_variant.sh <name> <source.hip> "<extra flags>"  ->  tools/exp/libafis_<name>.so = the current objects with ONE source rebuilt with extra flags
# (compared on one box by tools/lib_ab.py; tools/exp/ is git-ignored but travels with gpurun)
set -e
NAME=$1; SRC=$2; EXTRA=$3
cd "$(dirname "$0")/../msu-latentafis_amd/csrc"
make -s -j8 libafis_hip.so
BASE=${SRC%.hip}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
case $BASE in
  adc_mfma) FL="$FL -fno-honor-nans -fno-slp-vectorize";;
  adc_refine|graph) FL="$FL -fno-slp-vectorize";;
esac
mkdir -p ../../tools/exp
/opt/rocm/bin/hipcc $FL $EXTRA -c $SRC -o /tmp/variant_${NAME}.o
OBJS=""
for o in adc.o adc_mfma.o adc_refine.o graph.o minu.o pq_encode.o afis_api.o afis_gallery.o afis_search.o template_io.o; do
  if [ "$o" = "$BASE.o" ]; then OBJS="$OBJS /tmp/variant_${NAME}.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ../../tools/exp/libafis_${NAME}.so
echo built tools/exp/libafis_${NAME}.so
