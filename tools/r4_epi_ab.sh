#!/bin/bash
# round-4 A/B on one box: base (committed 47ee3c3 kernels) vs the packed-FMA / staged-GELU epilogue
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_gpu_gemm_ring.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -5
for lib in base_libautognothi_hip.so libautognothi_hip.so; do
  for cfg in "gelu 1 3072" "bias 1 2304"; do
    set -- $cfg
    echo "== timeline $lib epi=$1 fold=$2 N=$3"
    GB_LIB=$lib GB_EPI=$1 GB_FOLD=$2 GB_N=$3 GB_K=768 python tools/gemm_timeline.py 2>&1 | head -3
  done
done
bash tools/ab_bench.sh autognothi_amd/lib/base_libautognothi_hip.so autognothi_amd/lib/libautognothi_hip.so
