#!/bin/bash
# A/B builds: tools/build_variant.sh <name> <source.hip> "<extra hipcc flags>"  ->  autognothi_amd/lib/<name>_libautognothi_hip.so
# (the named source recompiled with the flags, every other object taken from the regular build; pick it with GB_LIB=<name>_libautognothi_hip.so
#  in the tools or AG_HIP_LIB=<path> anywhere)
set -euo pipefail
NAME=$1; SRC=$2; EXTRA=${3:-}
cd "$(dirname "$0")/../autognothi_amd/csrc"
OUT=../lib
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
base=${SRC%.*}
$HIPCC $FLAGS $EXTRA -x hip -c "$SRC" -o "$OUT/${NAME}_${base}.o"
OBJS=()
for src in gemm.hip gemm_tn.hip gemm_big.hip side_mlp.hip probe.hip attention.hip sampler.hip elementwise.hip shapley.hip train.hip train_fused.hip encoder.cpp capi.cpp; do
  b=${src%.*}
  if [ "$b" == "$base" ]; then OBJS+=("$OUT/${NAME}_${base}.o"); else OBJS+=("$OUT/$b.o"); fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/${NAME}_libautognothi_hip.so" "${OBJS[@]}"
echo "built $OUT/${NAME}_libautognothi_hip.so"
