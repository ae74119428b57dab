#!/bin/bash
# round 4: the split fc2 (parity + small-batch sweep, split on / off on one box), the epilogue store granularity A/B
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_gpu_gemm_split.py -q -x 2>&1 | tail -5
for b in 1 4; do
  for sp in 1 0 1 0; do
    AG_GEMM_SPLIT=$sp timeout 250 python bench.py --steps 30 --warmup 5 --batch $b --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('split=$sp B=$b', d['value'], 'fwd/s', d['ms_per_step'], 'ms/step', {n:v['avg_us'] for n,v in d['roofline']['kernels'].items()})"
  done
done
for lib in libautognothi_hip.so store16_libautognothi_hip.so; do
  for cfg in "gelu 1 3072" "bias 1 2304"; do
    set -- $cfg
    echo "== timeline $lib epi=$1 fold=$2 N=$3"
    GB_LIB=$lib GB_EPI=$1 GB_FOLD=$2 GB_N=$3 GB_K=768 python tools/gemm_timeline.py 2>&1 | grep -v amdgpu.ids | head -2
  done
done
bash tools/ab_bench.sh autognothi_amd/lib/libautognothi_hip.so autognothi_amd/lib/store16_libautognothi_hip.so
python -m pytest tests/test_gpu_rccl.py -q -x 2>&1 | tail -5
