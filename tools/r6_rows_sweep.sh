#!/bin/bash
cd "$(dirname "$0")/.."
rm -f gpurun_out/r6_rows_sweep.log
for r in 1536 3072 768 1536 3072; do
  echo "== AG_TARGET_ROWS=$r" >> gpurun_out/r6_rows_sweep.log
  AG_TARGET_ROWS=$r STEPS=72 timeout 400 python tools/train_step_bench.py duo_bert_base vit_base froyo_vit_base 2>&1 | grep -v amdgpu | cut -c1-150 >> gpurun_out/r6_rows_sweep.log
done
cat gpurun_out/r6_rows_sweep.log
