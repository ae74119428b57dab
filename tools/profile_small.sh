#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of the masked forward at a small batch, planner off / on.  usage: profile_small.sh <outdir> <point> ...
R=$GRAFT_REPO_ROOT; OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
for mode in off auto; do
  if [ $mode = off ]; then export AG_WS_ROUTE=0; else unset AG_WS_ROUTE; fi
  for pt in "$@"; do
    tag=${pt//:/_}
    FP_STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_${mode}_$tag -- python3 $R/bench.py --workload ${pt%%:*} --batch $(echo $pt | cut -d: -f2) --masks $(echo $pt: | cut -d: -f3 | sed 's/^$/0/') --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $R/$OUT/prof_${mode}_$tag.log 2>&1
    find $R/$OUT/prof_${mode}_$tag -name "*kernel_trace.csv" -delete
    f=$(find $R/$OUT/prof_${mode}_$tag -name "*kernel_stats.csv" | head -1)
    echo "== $mode $pt"; python3 $R/tools/summarize_kernel_stats.py $f 14
  done
done
