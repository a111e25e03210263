#!/bin/bash
# which CU masks does the runtime honour?  each probe under its own timeout (a mask that leaves a queue without CUs would hang)
probe() { name=$1; mask=$2; echo "== $name $mask"; AFIS_CU_MASK=$mask timeout 120 python bench.py --gallery 20000 --queries 8 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; l=sys.stdin.read().strip(); 
try:
    j=json.loads(l); print(j['ms_per_step'], j['stage_ms_per_step']['adc_bound_ms'], j['stage_ms_per_step']['tex_tail_ms'])
except Exception as e: print('FAILED/TIMEOUT', l[-200:])"; }
Z=0; F=ffffffff
probe all $F,$F,$F,$F,$F,$F,$F,$F
probe low128 $F,$F,$F,$F,$Z,$Z,$Z,$Z
probe hi128 $Z,$Z,$Z,$Z,$F,$F,$F,$F
probe hi64 $Z,$Z,$Z,$Z,$Z,$Z,$F,$F
probe mid128 $Z,$Z,$F,$F,$F,$F,$Z,$Z
