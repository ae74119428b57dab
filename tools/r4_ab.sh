#!/bin/bash
# Round-4 same-box A/B runs (gpurun -- 'bash tools/r4_ab.sh <what>'); the numbers of profiles/HISTORY.md §10 come from here.
#   epi    base (lib/base_libautognothi_hip.so = the previous build) vs current: parity, stamped tile timelines (fc1 + GELU, QKV), bench A/B
#   split  ag_gemm_resid_split: parity, small-batch forward with AG_GEMM_SPLIT on / off alternating, stamped timeline
#   resid  residual epilogue: parity, stamped timelines at K = 768 / 3072, bench A/B
R=$GRAFT_REPO_ROOT; cd $R
WHAT=${1:-epi}
timeline() {  # lib epi fold N K
  echo "== timeline $1 epi=$2 fold=$3 N=$4 K=$5"
  GB_LIB=$1 GB_EPI=$2 GB_FOLD=$3 GB_N=$4 GB_K=$5 python tools/gemm_timeline.py 2>&1 | grep -v amdgpu.ids | head -2
}
case $WHAT in
epi)
  python -m pytest tests/test_gpu_gemm_ring.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -4
  for lib in base_libautognothi_hip.so libautognothi_hip.so; do timeline $lib gelu 1 3072 768; timeline $lib bias 1 2304 768; done
  bash tools/ab_bench.sh autognothi_amd/lib/base_libautognothi_hip.so autognothi_amd/lib/libautognothi_hip.so ;;
split)
  python -m pytest tests/test_gpu_gemm_split.py -q -x 2>&1 | tail -4
  for b in 1 4; do for sp in 1 0 1 0; do
    AG_GEMM_SPLIT=$sp timeout 250 python bench.py --steps 30 --warmup 5 --batch $b --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('split=$sp B=$b', d['value'], 'fwd/s', d['ms_per_step'], 'ms/step', {n:v['avg_us'] for n,v in d['roofline']['kernels'].items()})"
  done; done
  timeline libautognothi_hip.so gelu 1 3072 768; timeline libautognothi_hip.so bias 1 2304 768 ;;
resid)
  python -m pytest tests/test_gpu_gemm_ring.py tests/test_gpu_kernels.py tests/test_gpu_gemm_split.py -q -x 2>&1 | tail -4
  for lib in base_libautognothi_hip.so libautognothi_hip.so; do timeline $lib resid 0 768 768; timeline $lib resid 0 768 3072; done
  bash tools/ab_bench.sh autognothi_amd/lib/base_libautognothi_hip.so autognothi_amd/lib/libautognothi_hip.so ;;
esac
