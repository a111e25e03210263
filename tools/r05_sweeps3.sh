#!/bin/bash
# more volume at the final kernels (HEAD): off-envelope, wide and headline shapes, tie_mode 0
cd "$(dirname "$0")/.."
O=gpurun_out/r05_sweeps3; mkdir -p $O
for seed in 8 9 10 11; do timeout 1500 python tools/offenv_sweep.py $seed 80 250 $O/offenv_seed$seed.json >> $O/offenv.log 2>&1; done
for seed in 152 153 154; do AFIS_SWEEP_WORKLOAD=wide timeout 1200 python tools/parity_sweep.py $seed 12 8000 >> $O/wide.log 2>&1; done
for seed in 143 144 145 146; do timeout 900 python tools/parity_sweep.py $seed 16 12000 >> $O/headline.log 2>&1; done
for seed in 113; do timeout 900 python tools/parity_sweep.py $seed 8 12000 tie0 >> $O/tie0.log 2>&1; done
for seed in 162 163; do timeout 900 python tools/shape_sweep.py $seed 24 60 >> $O/shapes.log 2>&1; done
grep -h "^seed\|vs tie" $O/*.log | cut -c1-200; grep -h -o '"pairs_with_any_differing_bit": [0-9]*' $O/offenv.log
