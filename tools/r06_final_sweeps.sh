#!/bin/bash
# The parity sweeps at the round's last kernels (run on the GPU box from the repo root; summarised into profiles/r06_parity_sweeps.json by hand-off to tools/r06_collect_sweeps.py)
O=gpurun_out/r06_sweeps; mkdir -p $O
python tools/parity_sweep.py 2061 16 12000 tie0 > $O/headline.txt 2>&1
AFIS_SWEEP_WORKLOAD=wide python tools/parity_sweep.py 2062 8 12000 > $O/wide.txt 2>&1
python tools/offenv_sweep.py 2063 80 250 $O/offenv.json > $O/offenv.txt 2>&1
python tools/shape_sweep.py 2064 24 60 > $O/shapes.txt 2>&1
AFIS_SWEEP_WORKLOAD=structured AFIS_SWEEP_DUP=10 python tools/parity_sweep.py 2065 67 3000 > $O/structured_dup10.txt 2>&1
AFIS_SWEEP_WORKLOAD=structured AFIS_SWEEP_DUP=30 python tools/parity_sweep.py 2066 20 3000 > $O/structured_dup30.txt 2>&1
AFIS_SWEEP_WORKLOAD=structured AFIS_SWEEP_DUP=0 python tools/parity_sweep.py 2067 20 3000 tie0 > $O/structured_dup0_tie0.txt 2>&1
for f in $O/*.txt; do echo "== $f"; tail -4 $f | cut -c1-400; done
