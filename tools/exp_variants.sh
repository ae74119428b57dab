#!/bin/bash
# same-box timing of the four encoder GEMM shapes under experiment builds (tools/build_variant.sh)
for i in 1 2; do
for lib in libautognothi_hip.so "$@"; do
  echo -n "$lib: "; GB_M=${GB_M:-302592} GB_LIB=$lib python tools/gemm_shapes.py 2>&1 | tail -1
done
done
