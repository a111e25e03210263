#!/bin/bash
# round-5 parity sweeps at the final kernels (GPU box; the oracle runs on the host's threads): headline shapes, wide shapes, latent shapes, off-envelope shapes, tie_mode 0
cd "$(dirname "$0")/.."
O=gpurun_out/r05_sweeps; mkdir -p $O
for seed in 101 102 103 104; do timeout 900 python tools/parity_sweep.py $seed 16 12000 >> $O/headline.log 2>&1; done
for seed in 111 112; do timeout 900 python tools/parity_sweep.py $seed 8 12000 tie0 >> $O/tie0.log 2>&1; done
for seed in 121 122 123; do AFIS_SWEEP_WORKLOAD=wide timeout 1200 python tools/parity_sweep.py $seed 12 8000 >> $O/wide.log 2>&1; done
for seed in 131 132 133; do timeout 900 python tools/shape_sweep.py $seed 24 60 >> $O/shapes.log 2>&1; done
for seed in 3 4; do timeout 1500 python tools/offenv_sweep.py $seed 80 250 $O/offenv_seed$seed.json >> $O/offenv.log 2>&1; done
grep -h "^seed\|vs tie" $O/*.log
