#!/bin/bash
# round 5: attention kernel variants (lib/libautognothi_hip_<name>.so built by tools/build_variant.sh), interleaved on one box:
# the kernel alone (tools/attn_bench.py) and the headline step (bench.py)
LIBS=("$@")
for i in 1 2; do
  for lib in "${LIBS[@]}"; do
    echo "$lib $(GB_LIB=$lib ATTN_ONLY=vit_base python tools/attn_bench.py 2>&1 | tail -1)"
  done
done
for i in 1 2; do
  for lib in "${LIBS[@]}"; do
    AG_HIP_LIB=$PWD/autognothi_amd/lib/$lib python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']; print('$lib', d['value'], d['ms_per_step'], {n:v['avg_us'] for n,v in k.items()})"
  done
done
