#!/bin/bash
# Run on the GPU box: SQ counters of the masked-attention kernel at the bench shape (ViT-base R=1536, T=197), separate passes
# (8 SQ slots per pass), for the VALU-vs-MFMA issue statement of DESIGN.md.  Output: gpurun_out/<tag>_attn_pmc.txt
R=$GRAFT_REPO_ROOT; TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
export ATTN_ONLY=vit_base
bash $R/tools/pmc_script.sh tools/attn_bench.py attn_bf16_kernel \
  "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES" \
  "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  > $R/gpurun_out/${TAG}_attn_pmc.txt 2>&1
cat $R/gpurun_out/${TAG}_attn_pmc.txt
