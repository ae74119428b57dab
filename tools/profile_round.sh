#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes (FETCH_SIZE / WRITE_SIZE) of the
# default bench command.  Outputs land in gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into profiles/.
set -u
R=$GRAFT_REPO_ROOT; TAG=${1:-final}; shift || true
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-secondary $*"   # the timed hot path only: secondary lines reuse the same kernels at other shapes
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}/stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}/fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}/write -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${TAG}/sq -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/prof_${TAG}/tcc -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_tcc.log 2>&1
# keep the merge small: the per-dispatch traces are large
find $R/gpurun_out/prof_${TAG} -name "*kernel_trace.csv" -delete
python3 $R/tools/summarize_profile.py $R/gpurun_out/prof_${TAG} $R/gpurun_out/prof_${TAG}_summary.json
tail -2 $R/gpurun_out/prof_${TAG}_stats.log
