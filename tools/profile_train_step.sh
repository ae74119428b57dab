#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of the bf16 explainer training step of one bench workload, one stream
# usage: profile_train_step.sh <workload> [images per step]
R=$GRAFT_REPO_ROOT; WL=${1:-duo_bert_base}; export TB=${2:-8} STEPS=${STEPS:-12} AG_TRAIN_PARTITION=${AG_TRAIN_PARTITION:-0}
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/train_step_bench.py $WL 2>/dev/null | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trainstep_$WL -- python3 $R/tools/train_step_bench.py $WL > $R/gpurun_out/trainstep_$WL.log 2>&1
find $R/gpurun_out/trainstep_$WL -name "*kernel_trace.csv" -delete
S=$(find $R/gpurun_out/trainstep_$WL -name "*kernel_stats.csv" | head -1)
tail -1 $R/gpurun_out/trainstep_$WL.log
python3 $R/tools/summarize_kernel_stats.py $S | head -${LINES_OUT:-45}
