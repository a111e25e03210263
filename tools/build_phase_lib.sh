#!/bin/bash
# builds tools/libafis_phase.so: the library with -DAFIS_PHASE_TIMING in graph.hip / minu.hip (see tools/phase_probe.py)
set -e
cd "$(dirname "$0")/../msu-latentafis_amd/csrc"
make -s -j8 libafis_hip.so libafis_hip_test.so
for f in graph minu; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DAFIS_PHASE_TIMING -c $f.hip -o /tmp/${f}_ph.o 2>/dev/null; done
hipcc --offload-arch=gfx950 -shared -fPIC adc.o adc_mfma_exp.o adc_direct.o adc_refine.o /tmp/graph_ph.o /tmp/minu_ph.o pq_encode.o afis_api_exp.o afis_gallery.o afis_search_exp.o afis_taps.o template_io.o -o ../../tools/libafis_phase.so
