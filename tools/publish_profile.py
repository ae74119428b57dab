#!/usr/bin/env python3
"""Copy a profiling round from gpurun_out/ into profiles/ and refresh profiles/traffic.json (what bench.py reports as
roofline.traffic).  usage: publish_profile.py <tag> <workload> <batch> <masks>   e.g.  r01_h vit_base 48 32"""
import glob, json, os, shutil, sys

tag, workload, batch, masks = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
suffix = f"{workload}_B{batch}_K{masks}"
stats = glob.glob(os.path.join(go, f"prof_{tag}", "stats", "*", "*kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(pr, f"{tag}_kernel_stats_{suffix}.csv"))
summ = json.load(open(os.path.join(go, f"prof_{tag}_summary.json")))
json.dump(summ, open(os.path.join(pr, f"{tag}_pmc_summary_{suffix}.json"), "w"), indent=1)
# bench.py kernel labels <- profiled kernel names (bf16 ViT hot path: LN-folded QKV / fc1+GELU, residual GEMMs with row stats)
# (<EPI, VAR, RLDS, HT>: the whole-tile instantiations, HT = false — the half-height-tail ones only serve launches of at most one round)
names = {"gemm<bias>": "gemm_stream_kernel<0, 1, false, false>", "gemm<bias+gelu>": "gemm_stream_kernel<1, 1, false, false>",
         "gemm<bias+residual>": "gemm_stream_kernel<2, 2, true, false>", "layernorm": "layernorm_kernel<unsigned short, unsigned short>"}
for lab, var in (("gemm<bias>", "0, 1"), ("gemm<bias+gelu>", "1, 1"), ("gemm<bias+residual>", "2, 2")):
    # (AG_GEMM_RLDS=0 / AG_GEMM_STREAM=0 / builds before round 3)
    for old in (f"gemm_stream_kernel<{var}, true>", f"gemm_stream_kernel<{var}, false>", f"gemm_stream_kernel<{var}>", f"gemm_line_kernel<{var}>", f"gemm_ring_kernel<{var}, false>"):
        if names[lab] not in summ and old in summ:
            names[lab] = old
attn = [k for k in summ if k.startswith("attn_stream3_kernel")] or [k for k in summ if k.startswith("attn_bf16_kernel")]
if attn:
    names["masked_attention"] = attn[0]
sys.path.insert(0, root)
import bench  # noqa: E402  (kernel_source_sha16: the counters are only quoted next to numbers of the same kernel build)
out = {"workload": workload, "batch": batch, "masks": masks, "precision": "bf16", "kernel_source_sha16": bench.kernel_source_sha16(),
       "source": f"profiles/{tag}_pmc_summary_{suffix}.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; "
                 "bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 per MI355X_MICROARCH.md HBM section)",
       "traffic_bytes_per_launch": {}, "l2_hit_rate": {}, "mfma_busy_frac": {}}
for label, k in names.items():
    if k in summ:
        # a class = its whole-tile instantiation + its half-height-tail one (<..., true>: the launches of at most one round, e.g. layer 0's
        # QKV on the B distinct inputs): the launch-weighted mean over both, i.e. over the same launches as before the tail existed
        parts = [summ[k]] + ([summ[k[:-len("false>")] + "true>"]] if k.endswith(", false>") and k[:-len("false>")] + "true>" in summ else [])
        for f, cnt in (("traffic_bytes_per_launch", "FETCH_SIZE"), ("l2_hit_rate", "TCC_HIT_sum"), ("mfma_busy_frac", "SQ_VALU_MFMA_BUSY_CYCLES")):
            have = [(q[f], q[cnt]["n"]) for q in parts if f in q and cnt in q]
            if have:
                out[f][label] = sum(v * n for v, n in have) / sum(n for _, n in have)
json.dump(out, open(os.path.join(pr, "traffic.json"), "w"), indent=1)
print(json.dumps(out["traffic_bytes_per_launch"]))
