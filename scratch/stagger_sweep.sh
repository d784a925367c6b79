#!/bin/bash
# Forward time against a start stagger of the big-tile GEMM's workgroups (experiment), one box.
run() {
  env "$@" timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['avg_launch_ms'])
" || exit 1
}
run VITM_PP_STAGGER=0
run VITM_PP_STAGGER=10000 VITM_PP_STAGGER_CLASSES=2
run VITM_PP_STAGGER=20000 VITM_PP_STAGGER_CLASSES=2
run VITM_PP_STAGGER=10000 VITM_PP_STAGGER_CLASSES=4
run VITM_PP_STAGGER=5000 VITM_PP_STAGGER_CLASSES=4
run VITM_PP_STAGGER=5000 VITM_PP_STAGGER_CLASSES=8
run VITM_PP_STAGGER=10000 VITM_PP_STAGGER_CLASSES=4 VITM_PP_STAGGER_MINT=2
run VITM_PP_STAGGER=0
