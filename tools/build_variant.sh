#!/bin/bash
# tools/build_variant.sh <name> <file.hip> <-Dflags...>: lib/libautognothi_hip_<name>.so = the shipped objects with <file.hip> recompiled under the flags (A/B builds)
set -euo pipefail
cd "$(dirname "$0")/../autognothi_amd/csrc"
name=$1; src=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable "$@" -x hip -c "$src" -o "../lib/${src%.*}_$name.o"
OBJS=()
for o in gemm gemm_tn gemm_big side_mlp probe attention cls_last sampler elementwise shapley train train_fused encoder capi; do
  if [ "$o" = "${src%.*}" ]; then OBJS+=("../lib/${o}_$name.o"); else OBJS+=("../lib/$o.o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "../lib/libautognothi_hip_$name.so" "${OBJS[@]}"
echo "built lib/libautognothi_hip_$name.so"
