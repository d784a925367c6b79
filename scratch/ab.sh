#!/bin/bash
# A/B of two builds of the library inside one box: alternates bench.py runs and prints images/s and ms per step.
# usage: bash scratch/ab.sh [rounds]
R=${1:-3}
L=$PWD/vision-transformer_amd/lib
for r in $(seq $R); do
  for v in new old; do
    lib=$L/libvitm_gfx950.so; [ $v = old ] && lib=$L/libvitm_gfx950_old.so
    VITM_GFX950_LIB=$lib timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['frac'] if d.get('roofline') else None)
" || exit 1
  done
done
