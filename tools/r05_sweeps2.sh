#!/bin/bash
# parity sweeps repeated at the FINAL kernels (after the large class took rolled templates of up to 512 minutiae and the readback went through the pinned buffer)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_sweeps2; mkdir -p $O
for seed in 5 6 7; do timeout 1500 python tools/offenv_sweep.py $seed 80 250 $O/offenv_seed$seed.json >> $O/offenv.log 2>&1; done
for seed in 141 142; do timeout 900 python tools/parity_sweep.py $seed 16 12000 >> $O/headline.log 2>&1; done
for seed in 151; do AFIS_SWEEP_WORKLOAD=wide timeout 1200 python tools/parity_sweep.py $seed 12 8000 >> $O/wide.log 2>&1; done
for seed in 161; do timeout 900 python tools/shape_sweep.py $seed 24 60 >> $O/shapes.log 2>&1; done
grep -h "^seed\|pairs_with_any" $O/*.log | cut -c1-400
