#!/bin/bash
# Throughput of the other benchmark configurations and a batch sweep of ViT-B/16 (run on the GPU box).
set -e
for args in "--config vit_tiny_cifar --batch 64" "--config vit_tiny_cifar --batch 64 --graph" "--config vit_tiny_cifar --batch 1024" "--config vit_tiny_cifar --batch 1024 --graph" \
            "--config vit_s16 --batch 256" "--config vit_s16 --batch 256 --graph" "--config vit_b16_384 --batch 64" "--config vit_l16 --batch 128" \
            "--config vit_b16 --batch 64" "--config vit_b16 --batch 128" "--config vit_b16 --batch 256 --graph" "--config vit_b16 --batch 332" "--config vit_b16 --batch 512"; do
  timeout -k 10 200 python -u bench.py $args --steps 20 --warmup 5 --no-cpu-baseline | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d['config']
        print('$args', '->', d['value'], 'img/s', d['ms_per_step'], 'ms', d['whole_forward'])
"
done
