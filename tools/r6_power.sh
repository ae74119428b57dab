#!/bin/bash
# round 6: is the large-M GEMM clock-limited?  time / effective clock / MFMA busy of the shipped kernel on random and on all-zero operands
cd "$(dirname "$0")/.."
R6_ROUNDS=5 R6_VARIANTS="random:AG_GEMM_RLDS=1;zero:DATA=zero" timeout 600 python tools/r6_gemm_ab.py > gpurun_out/r6_zero_ab.log 2>&1
cat gpurun_out/r6_zero_ab.log
for s in qkv fc1 fc2; do
  GB_ZERO=0 bash tools/clock_pmc.sh libautognothi_hip.so $s 2>&1 | tail -1 | sed 's/^/random: /' | tee -a gpurun_out/r6_clock_pmc.txt
  GB_ZERO=1 bash tools/clock_pmc.sh libautognothi_hip.so $s 2>&1 | tail -1 | sed 's/^/zero:   /' | tee -a gpurun_out/r6_clock_pmc.txt
done
find gpurun_out -name "*kernel_trace.csv" -path "*clk_*" -delete; find gpurun_out -name "*counter_collection.csv" -path "*clk_*" -delete
