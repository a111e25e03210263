#!/bin/bash
# ADC chunk size (gallery templates per workgroup) against shard size: prints gallery, chunk, q/s, ms/step, adc ms
for g in ${GALLERIES:-12500 25000}; do for c in ${CHUNKS:-64 128 256 512}; do
  python bench.py --gallery $g --chunk $c --no-cpu-baseline --steps 3 2>&1 | tail -1 > /tmp/line.json
  python - "$c" <<'PY'
import sys, json
d = json.load(open("/tmp/line.json")); print(d["config"]["gallery"], sys.argv[1], d["value"], d["ms_per_step"], d["stage_ms_per_step"]["adc_ms"])
PY
done; done
