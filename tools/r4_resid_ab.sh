#!/bin/bash
# round 4: residual epilogue (packed statistics sums, RLDS image swizzle on (row >> 1) & 7): parity, timeline, same-box bench A/B vs the previous build
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_gpu_gemm_ring.py tests/test_gpu_kernels.py tests/test_gpu_gemm_split.py -q -x 2>&1 | tail -4
for lib in base_libautognothi_hip.so libautognothi_hip.so; do
  for k in 768 3072; do
    echo "== timeline $lib resid K=$k"
    GB_LIB=$lib GB_EPI=resid GB_N=768 GB_K=$k python tools/gemm_timeline.py 2>&1 | grep -v amdgpu.ids | head -2
  done
done
bash tools/ab_bench.sh autognothi_amd/lib/base_libautognothi_hip.so autognothi_amd/lib/libautognothi_hip.so
