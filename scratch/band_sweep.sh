#!/bin/bash
# Forward time against the weight-band size of the big-tile GEMM's tile order (VITM_PP_BAND_KB), one box.
for kb in 2560 800 1200 1600 2000 3200 4800 2560; do
  VITM_PP_BAND_KB=$kb timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('band_kb $kb', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['avg_launch_ms'])
" || exit 1
done
