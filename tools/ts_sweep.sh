#!/bin/bash
# ADC tile_share (chunks taken against the same LUT tile by the blocks that follow one another on an XCD) x chunk size:  CFGS="share:chunk ..." GALLERY=n
for cfg in ${CFGS:-1:0 4:0}; do
  s=${cfg%%:*}; c=${cfg##*:}
  python bench.py --gallery ${GALLERY:-100000} --tile-share $s --chunk $c --no-cpu-baseline --steps 3 2>&1 | tail -1 > /tmp/line.json; python - $s $c <<'PY'
import sys, json
d = json.load(open("/tmp/line.json")); print(d["config"]["gallery"], "tile_share", sys.argv[1], "chunk", sys.argv[2], d["value"], d["ms_per_step"], d["stage_ms_per_step"]["adc_ms"])
PY
done
