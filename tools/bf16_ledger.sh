#!/bin/bash
# bf16 error ledger under each knob (one process per setting: the knobs are read at import / first call)
mkdir -p gpurun_out
OUT=gpurun_out/bf16_ledger.jsonl
LOG=gpurun_out/bf16_ledger.log
rm -f $OUT $LOG
run() { echo "### $*" >> $LOG; env "$@" python tools/bf16_ledger.py $TAG --json $OUT >> $LOG 2>&1; }
TAG=vit_base_l12
run AG_NOP=1
run AG_LN_FOLD=0
TAG=bert_base_l12
run AG_NOP=1
run AG_BERT_LN_FOLD=0
run AG_BERT_PRUNE=0
run AG_LN_FOLD=0
cat $LOG
