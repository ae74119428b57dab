cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
FP_STEPS=30 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b1 -- python3 $R/tools/fwd_points.py vit_base:1 > $R/gpurun_out/prof_b1.log 2>&1
find $R/gpurun_out/prof_b1 -name "*kernel_trace.csv" -delete
S=$(find $R/gpurun_out/prof_b1 -name "*kernel_stats.csv" | head -1)
tail -1 $R/gpurun_out/prof_b1.log | cut -c1-600
python3 $R/tools/summarize_kernel_stats.py $S 40
