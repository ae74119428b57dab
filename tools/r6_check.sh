#!/bin/bash
# round-6 dev run: the new tests, the nt-A experiment, the grouped-dW A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cls_last.py tests/test_gpu_gemm_ex.py tests/test_gpu_training16.py tests/test_gpu_graph.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r6_tests1.log
R6_ROUNDS=5 R6_VARIANTS="base:AG_GEMM_PIPE=0;ntA:AG_GEMM_PIPE=0,LIB=ntA;hoistHi:AG_GEMM_PIPE=0,LIB=hh" timeout 600 python tools/r6_gemm_ab.py > gpurun_out/r6_ntA_ab.log 2>&1
rm -f gpurun_out/r6_dw_group_ab.log
for g in 0 8 0 8 4 16; do
  echo "== AG_TRAIN_DW_GROUP=$g" >> gpurun_out/r6_dw_group_ab.log
  AG_TRAIN_DW_GROUP=$g STEPS=36 timeout 400 python tools/train_step_bench.py duo_bert_base vit_base froyo_vit_base >> gpurun_out/r6_dw_group_ab.log 2>&1
done
tail -15 gpurun_out/r6_tests1.log; cat gpurun_out/r6_ntA_ab.log; grep -v amdgpu gpurun_out/r6_dw_group_ab.log | tail -40
