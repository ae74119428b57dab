#!/bin/bash
# the last layer without its K / V projection (AG_LAST_KV_SKIP) on / off: bench.py headline and ViT-large, interleaved on one box
for i in 1 2 3; do
  for k in 0 1; do
    AG_LAST_KV_SKIP=$k python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']; print('kv_skip=$k', d['value'], d['ms_per_step'], {n:(v['avg_us'], v['launches']) for n,v in k.items()})"
  done
done
for k in 0 1; do
  AG_LAST_KV_SKIP=$k python bench.py --workload vit_large --no-cpu-baseline --no-secondary --steps 5 --warmup 2 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('vit_large kv_skip=$k', d['value'], d['ms_per_step'])"
done
