#!/bin/bash
# explainer training step (ms/step) of three workloads at several ring-kernel thresholds (AG_GEMM_BIG_MIN_TILES, read per call)
for wl in vit_base duo_bert_base froyo_vit_base; do for mt in 48 90 130 200; do
  echo "$wl TB=${TB:-8} min_tiles=$mt: $(WL=$wl AG_GEMM_BIG_MIN_TILES=$mt python3 tools/train_step_profile.py 2>&1 | tail -1)"
done; done
