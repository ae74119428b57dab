#!/bin/bash
# interleaved same-box A/B of bench.py between libraries: tools/r5_ab2.sh <lib1.so> <lib2.so>  (files in autognothi_amd/lib/)
for i in 1 2 3; do
  for lib in "$@"; do
    AG_HIP_LIB=$PWD/autognothi_amd/lib/$lib python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']; print('$lib', d['value'], d['ms_per_step'], {n:v['avg_us'] for n,v in k.items()})"
  done
done
