#!/bin/bash
# Build the C-ABI shared library for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../lib
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
OBJS=()
PIDS=()
for src in gemm.hip gemm_tn.hip gemm_big.hip side_mlp.hip probe.hip attention.hip cls_last.hip sampler.hip elementwise.hip shapley.hip train.hip train_fused.hip encoder.cpp capi.cpp; do
  obj="$OUT/${src%.*}.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ common.h -nt "$obj" ] || [ ../../include/autognothi_hip.h -nt "$obj" ]; then
    echo "hipcc $src"
    rm -f "$obj"          # a failed compile must not leave a stale object behind
    $HIPCC $FLAGS -x hip -c "$src" -o "$obj" &
    PIDS+=($!)
  fi
  OBJS+=("$obj")
done
# test infrastructure: the same library with the two earlier large-M GEMM generations compiled in (gemm_big.hip -DAG_REF_KERNELS):
# the parity tests check the shipped stream kernel bit for bit against them (tests/test_gpu_gemm_ring.py loads this file beside
# the shipped one); nothing under autognothi_amd/ loads it
REFOBJ="$OUT/gemm_big_ref.o"
if [ ! -f "$REFOBJ" ] || [ gemm_big.hip -nt "$REFOBJ" ] || [ common.h -nt "$REFOBJ" ] || [ ../../include/autognothi_hip.h -nt "$REFOBJ" ]; then
  echo "hipcc gemm_big.hip -DAG_REF_KERNELS"
  rm -f "$REFOBJ"
  $HIPCC $FLAGS -DAG_REF_KERNELS -x hip -c gemm_big.hip -o "$REFOBJ" &
  PIDS+=($!)
fi
for pid in "${PIDS[@]}"; do wait "$pid"; done   # (set -e: the first failed compile aborts the build)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libautognothi_hip.so" "${OBJS[@]}"
REFOBJS=()
for o in "${OBJS[@]}"; do
  if [ "$o" = "$OUT/gemm_big.o" ]; then REFOBJS+=("$REFOBJ"); else REFOBJS+=("$o"); fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libautognothi_hip_ref.so" "${REFOBJS[@]}"
echo "built $OUT/libautognothi_hip.so (+ libautognothi_hip_ref.so: parity reference kernels, tests only)"
