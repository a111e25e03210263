#!/bin/bash
# volume at the round's last state (bound pass with the operand prefetch, launch groups of about five million pairs): new seeds
cd "$(dirname "$0")/.."
O=gpurun_out/r05_sweeps4; mkdir -p $O
for seed in 12 13; do timeout 1500 python tools/offenv_sweep.py $seed 80 250 $O/offenv_seed$seed.json >> $O/offenv.log 2>&1; done
for seed in 155; do AFIS_SWEEP_WORKLOAD=wide timeout 1200 python tools/parity_sweep.py $seed 12 8000 >> $O/wide.log 2>&1; done
for seed in 147 148 149; do timeout 900 python tools/parity_sweep.py $seed 16 12000 >> $O/headline.log 2>&1; done
for seed in 164; do timeout 900 python tools/shape_sweep.py $seed 24 60 >> $O/shapes.log 2>&1; done
grep -h "^seed\|vs tie" $O/*.log | cut -c1-200; grep -h -o '"pairs_with_any_differing_bit": [0-9]*' $O/offenv.log
