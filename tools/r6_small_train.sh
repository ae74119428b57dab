#!/bin/bash
# round 6: the config-5 training step at 2 / 4 images per GPU (the 8-GPU strong-scaling shard): side streams on / off, one / two streams
cd "$(dirname "$0")/.."
rm -f gpurun_out/r6_small_train.log
for tb in 2 4; do for part in 0 auto; do for minrows in 1024 0; do
  echo "== TB=$tb AG_TRAIN_PARTITION=$part AG_TRAIN_SIDE_MIN_ROWS=$minrows" >> gpurun_out/r6_small_train.log
  if [ $part = auto ]; then unset AG_TRAIN_PARTITION; else export AG_TRAIN_PARTITION=$part; fi
  TB=$tb AG_TRAIN_SIDE_MIN_ROWS=$minrows STEPS=48 timeout 300 python tools/train_step_bench.py duo_bert_base froyo_vit_base 2>&1 | grep -v amdgpu | cut -c1-150 >> gpurun_out/r6_small_train.log
done; done; done
cat gpurun_out/r6_small_train.log
