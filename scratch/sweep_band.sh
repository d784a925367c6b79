#!/bin/bash
# Forward rate against the column-band size of the big-tile GEMM's tile order (VITM_PP_BAND_KB; 0 = plain row-major).
for rep in 1 2; do
for kb in 2560 0 1024 1536 2048 3072 4096 6144; do
  VITM_PP_BAND_KB=$kb timeout -k 10 120 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('band_kb $kb', d['value'], 'img/s', d['ms_per_step'], 'ms  fc1', round(d['roofline']['avg_launch_ms']*1e3,1), 'us')
"
done
done
