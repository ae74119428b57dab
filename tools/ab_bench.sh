#!/bin/bash
# same-box A/B of bench.py between builds of the library: tools/ab_bench.sh <lib1.so> <lib2.so> ... [-- bench args...]
# (interleaved runs, one box; the guide's rule 24: never rank builds across devices)
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for i in 1 2 3; do
  for lib in "${LIBS[@]}"; do
    AG_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 "$@" 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']; print('$lib', d['value'], d['ms_per_step'], {n:v['avg_us'] for n,v in k.items()})"
  done
done
