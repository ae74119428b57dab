#!/bin/bash
# round-6: the whole GPU suite + the default bench line of the current build
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r6_gpu_tests.log
timeout 1500 python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err
cat gpurun_out/r6_gpu_tests.log; tail -c 2500 gpurun_out/r6_bench_default.json; tail -5 gpurun_out/r6_bench_default.err
