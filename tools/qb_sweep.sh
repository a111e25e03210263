#!/bin/bash
# latents per launch group (query_batch): QB="4 8 16" GALLERY=n
for qb in ${QB:-4 8 16}; do
  python bench.py --gallery ${GALLERY:-100000} --query-batch $qb --no-cpu-baseline --steps 2 2>&1 | tail -1 > /tmp/line.json; python - $qb <<'PY'
import sys, json
d = json.load(open("/tmp/line.json")); print(d["config"]["gallery"], "query_batch", sys.argv[1], d["value"], d["ms_per_step"], d["stage_ms_per_step"])
PY
done
