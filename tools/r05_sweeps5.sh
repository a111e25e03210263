#!/bin/bash
# volume at the round's very last state (values in the compact list only, bound-pass chunks per XCD): new seeds
cd "$(dirname "$0")/.."
O=gpurun_out/r05_sweeps5; mkdir -p $O
for seed in 14; do timeout 1500 python tools/offenv_sweep.py $seed 80 250 $O/offenv_seed$seed.json >> $O/offenv.log 2>&1; done
for seed in 156; do AFIS_SWEEP_WORKLOAD=wide timeout 1200 python tools/parity_sweep.py $seed 12 8000 >> $O/wide.log 2>&1; done
for seed in 150; do timeout 900 python tools/parity_sweep.py $seed 16 12000 >> $O/headline.log 2>&1; done
for seed in 165; do timeout 900 python tools/shape_sweep.py $seed 24 60 >> $O/shapes.log 2>&1; done
grep -h "^seed\|vs tie" $O/*.log | cut -c1-200; grep -h -o '"pairs_with_any_differing_bit": [0-9]*' $O/offenv.log
