#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of the explainer training step of one bench workload.
# usage: profile_train.sh <workload> [images per step]
R=$GRAFT_REPO_ROOT; export WL=${1:-vit_base}; export TB=${2:-8}
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/train_step_profile.py > $R/gpurun_out/train_$WL.plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/train_$WL -- python3 $R/tools/train_step_profile.py > $R/gpurun_out/train_$WL.log 2>&1
find $R/gpurun_out/train_$WL -name "*kernel_trace.csv" -delete
S=$(find $R/gpurun_out/train_$WL -name "*kernel_stats.csv" | head -1)
echo "== $WL TB=$TB: $(tail -1 $R/gpurun_out/train_$WL.plain.log) (unprofiled)  $(tail -1 $R/gpurun_out/train_$WL.log) (profiled)"
python3 $R/tools/summarize_kernel_stats.py $S | head -${3:-28}
