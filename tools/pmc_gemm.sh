#!/bin/bash
# usage: tools/pmc_gemm.sh <variant> <shape> <outdir-suffix>   (run on the GPU box via gpurun)
R=$GRAFT_REPO_ROOT; V=$1; S=$2; TAG=$3
cd /tmp && export TMPDIR=/tmp
export AG_GEMM_VARIANT=$V GB_ONLY=$S
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -- python3 $R/tools/gemm_bench.py > $R/gpurun_out/pmc_${TAG}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${TAG}_b -- python3 $R/tools/gemm_bench.py > $R/gpurun_out/pmc_${TAG}_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/pmc_${TAG}_c -- python3 $R/tools/gemm_bench.py > $R/gpurun_out/pmc_${TAG}_c.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${TAG}_d -- python3 $R/tools/gemm_bench.py > $R/gpurun_out/pmc_${TAG}_d.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_${TAG}_e -- python3 $R/tools/gemm_bench.py > $R/gpurun_out/pmc_${TAG}_e.log 2>&1
tail -3 $R/gpurun_out/pmc_${TAG}_a.log
