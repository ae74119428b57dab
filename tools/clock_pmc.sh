#!/bin/bash
# effective shader clock + wave-cycle counters of one GEMM shape under a library build:
#   tools/clock_pmc.sh <lib.so> <shape>   (GPU box; prints per-kernel mean duration, GRBM_GUI_ACTIVE/8/duration, wave-cycle split)
R=$GRAFT_REPO_ROOT; LIB=$1; S=${2:-fc2}
cd /tmp && export TMPDIR=/tmp
export AG_HIP_LIB=$R/autognothi_amd/lib/$LIB GB_ONLY=$S GB_M=${GB_M:-302592}
D=$R/gpurun_out/clk_${LIB%%_lib*}_$S
rm -rf $D
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $D -- python3 $R/tools/gemm_bench.py > $D.log 2>&1
python3 - <<PY
import csv, glob, collections
kt = glob.glob("$D/*/*kernel_trace.csv")[0]
cc = glob.glob("$D/*/*counter_collection.csv")[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    if "gemm_ring" in r["Kernel_Name"] or "gemm_line" in r["Kernel_Name"] or "gemm_stream" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if "gemm_ring" in r["Kernel_Name"] or "gemm_line" in r["Kernel_Name"] or "gemm_stream" in r["Kernel_Name"]:
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
ids = [i for i in cnt if i in dur][3:]
n = len(ids)
us = sum(dur[i] for i in ids) / n
m = {k: sum(cnt[i][k] for i in ids) / n for k in cnt[ids[0]]}
ghz = m["GRBM_GUI_ACTIVE"] / 8 / us / 1e3
print("$LIB $S: %d launches, %.1f us, eff clock %.3f GHz, cycles/launch %.3e, wave-cycles %.3e (wait_any %.2f, wait_inst %.2f, active_inst %.2f), mfma busy %.3f of SIMD cycles" % (
    n, us, ghz, m["GRBM_GUI_ACTIVE"] / 8, m["SQ_WAVE_CYCLES"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
    m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
