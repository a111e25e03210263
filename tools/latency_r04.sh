#!/bin/bash
# Single-latent latency (BASELINE.json configs[1]; One2List_matching, matcher.cpp:216-337): 1 latent vs 10k / 100k resident templates.
set -x
mkdir -p gpurun_out/lat
for QG in "1 10000" "1 100000" "4 100000" "16 100000"; do
  set -- $QG
  python bench.py --queries $1 --gallery $2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/lat/q$1_g$2.json 2> gpurun_out/lat/q$1_g$2.err
  tail -c 1500 gpurun_out/lat/q$1_g$2.json
done
