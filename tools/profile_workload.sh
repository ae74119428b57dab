#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of the timed hot path of one bench workload.  usage: profile_workload.sh <workload> [batch]
R=$GRAFT_REPO_ROOT; W=${1:-ltt_vit_base}; B=${2:-48}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$W -- python3 $R/bench.py --workload $W --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_$W.log 2>&1
find $R/gpurun_out/prof_$W -name "*kernel_trace.csv" -delete
tail -1 $R/gpurun_out/prof_$W.log | cut -c1-200
