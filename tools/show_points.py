"""print tools/fwd_points.py / tools/ws_calib.sh output compactly"""
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        try:
            d = json.loads(l)
        except Exception:
            continue
        cl = d["classes"]
        g = lambda k: (cl.get(k, {}).get("avg_us", 0), cl.get(k, {}).get("n", 0))
        print(f'{d.get("tag",""):10s} {d["point"]:16s} eager {d["eager_fwd_s"]:9.1f} graph {d["graph_fwd_s"]:9.1f} ms {d["eager_ms"]:.3f}/{d["graph_ms"]:.3f} '
              f'frac {d["exec_frac_graph"]:.3f} | qkv {g("gemm<bias>")[0]:6.1f} fc1 {g("gemm<bias+gelu>")[0]:6.1f} res {g("gemm<bias+residual>")[0]:6.1f} '
              f'att {g("masked_attention")[0]:5.1f} ln {g("layernorm")[0]:4.1f}x{g("layernorm")[1]}')
