#!/bin/bash
# bench.py per-class kernel times under GEMM tile-order / store-policy knobs (same box, one pass each + a repeat of the default)
run() { env "$@" python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']; print('$*', d['value'], {n:v['avg_us'] for n,v in k.items() if n.startswith('gemm<bias') and 'f32' not in n})"; }
run AG_X=0
for g in 1 2 3 4 6 12; do run AG_GEMM_NGRP=$g; done
run AG_GEMM_NT=0
run AG_GEMM_NT=1
run AG_GEMM_WFIT_MB=1.6
run AG_GEMM_WFIT_MB=3.2
run AG_X=0
