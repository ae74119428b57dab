#!/bin/bash
# round 5: attn_stream3_kernel on / off — the attention kernel alone (tools/attn_bench.py) and the headline step (bench.py), interleaved on one box
for i in 1 2; do
  for s3 in 1 0; do
    echo "AG_ATTN_STREAM3=$s3 $(AG_ATTN_STREAM3=$s3 ATTN_ONLY=vit_base python tools/attn_bench.py 2>&1 | tail -1)"
  done
done
for i in 1 2; do
  for s3 in 1 0; do
    AG_ATTN_STREAM3=$s3 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']; print('stream3=$s3', d['value'], d['ms_per_step'], {n:v['avg_us'] for n,v in k.items()})"
  done
done
