#!/bin/bash
# round 6: CUs per XCD of the target forward's GEMM in the two-stream epoch, with the grouped dW products
cd "$(dirname "$0")/.."
rm -f gpurun_out/r6_part_sweep.log
for c in auto 12 16 20 24 28; do
  echo "== AG_TRAIN_PARTITION=$c" >> gpurun_out/r6_part_sweep.log
  if [ $c = auto ]; then unset AG_TRAIN_PARTITION; else export AG_TRAIN_PARTITION=$c; fi
  STEPS=36 timeout 400 python tools/train_step_bench.py duo_bert_base vit_base froyo_vit_base 2>&1 | grep -v amdgpu | cut -c1-140 >> gpurun_out/r6_part_sweep.log
done
cat gpurun_out/r6_part_sweep.log
