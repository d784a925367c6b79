#!/bin/bash
# A/B of builds of the library on the training step: scratch/ab_train.sh <a.so> <b.so> ... (two rounds over all of them)
set -e
for i in 1 2; do
  for lib in "$@"; do
    VITM_GFX950_LIB=$PWD/$lib timeout -k 10 120 python -u bench.py --mode train --dtype bf16 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', d['value'], 'img/s', d['ms_per_step'], 'ms')
"
  done
done
