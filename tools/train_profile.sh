#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of the explainer training step (TB images per step, MIXED=1 for bf16 GEMM operands)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
export TB=${TB:-32} AG_TRAIN_BF16=${MIXED:-1}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -- python3 $R/tools/train_step_profile.py > $R/gpurun_out/prof_train.log 2>&1
f=$(find $R/gpurun_out/prof_train -name "*kernel_stats.csv" | head -1)
find $R/gpurun_out/prof_train -name "*kernel_trace.csv" -delete
grep "ms/step" $R/gpurun_out/prof_train.log
head -25 $f | awk -F'","' '{printf "%-90s %6s %12s %10s %s\n", substr($1,2,90), $2, $3, $4, $5}'
