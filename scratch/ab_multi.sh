#!/bin/bash
# A/B/C...: alternates bench.py over every library lib/libvitm_gfx950*.so, R rounds, one box.
R=${1:-2}
L=$PWD/vision-transformer_amd/lib
for r in $(seq $R); do
  for lib in $L/libvitm_gfx950*.so; do
    VITM_GFX950_LIB=$lib timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$(basename $lib)', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['avg_launch_ms'])
" || exit 1
  done
done
