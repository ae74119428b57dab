"""bf16-vs-fp32 error ledger of the masked forward (measurement aid; the fp32 mode is the one pinned to the reference at 1e-4).

For one input and its K fixture masks — row 0 replaced by the all-visible mask, row 1 by the least-visible mask of the set —
prints, per encoder depth, the relative L2 error of the bf16 hidden state against the fp32 one for the all-visible row, the
least-visible row and the rest, then the error of the head outputs.  Run once per knob setting (AG_LN_FOLD, AG_BERT_LN_FOLD,
AG_BERT_PRUNE are read at import: e.g. `AG_LN_FOLD=0 python tools/bf16_ledger.py vit_base_l12`).

usage: python tools/bf16_ledger.py vit_base_l12|bert_base_l12 [--json out.json]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from util import build_case  # noqa: E402


def rel(a, b, dim):
    return ((a - b).double().pow(2).sum(dim).sqrt() / b.double().pow(2).sum(dim).sqrt().clamp_min(1e-30)).cpu().numpy()


def main():
    tag = sys.argv[1]
    from autognothi_amd import _lib as L, engine, ops
    dev = torch.device("cuda:0")
    c = build_case(tag)
    vit = c["meta"]["kind"] == "vit"
    srg = c["surrogate"].to(dev)
    model = srg.vit if vit else srg.bert
    cfg = model.config
    xs = torch.from_numpy(c["xs"]).to(dev)[:1]
    masks = torch.from_numpy(c["masks"])[:c["K"]].clone()
    vis = masks.sum(1)
    lo = int(vis.argmin())
    masks[1] = masks[lo].clone()
    masks[0] = 1
    vis = masks.sum(1).numpy()
    masks = masks.to(dev)
    bits = ops.pack_mask(masks)
    rows, t = bits.shape[0], c["P"] + 1
    n_layers = cfg.num_hidden_layers
    knobs = {k: os.environ.get(k, "") for k in ("AG_LN_FOLD", "AG_BERT_LN_FOLD", "AG_BERT_PRUNE", "AG_F32_STREAM")}
    out = dict(tag=tag, knobs=knobs, visible=vis.tolist(), layers=[])
    print(f"== {tag} knobs {knobs}; visible players: row0 {vis[0]} row1 {vis[1]} others {sorted(vis[2:].tolist())}")
    kind = L.AG_MASK_VIT_MUL if vit else L.AG_MASK_BERT_ADD
    hid = {}
    with torch.no_grad():
        for l in sorted(set([1, 2, 3, 4, 6, 8, 10, n_layers - 1, n_layers])):
            for prec in ("fp32", "bf16"):
                engine.set_precision(prec)
                dt = engine.get_precision()
                h0 = model.embed(xs, dt) if vit else model.embed(xs, None, dt)
                enc = engine.PackedEncoder(list(model.encoder.layers)[:l], kind, t, cfg.hidden_size, cfg.intermediate_size,
                                           cfg.num_attention_heads, cfg.layer_norm_eps)
                hid[prec] = enc.forward(h0, rows, rows, bits, False, dt).float()
                if prec == "bf16":   # the all-visible row alone: small-GEMM path (no ring kernel, no LayerNorm fold)
                    alone = enc.forward(h0, 1, 1, bits[:1].contiguous(), False, dt).float()
            e_cls = rel(hid["bf16"][:, 0], hid["fp32"][:, 0], -1)
            e_all = rel(hid["bf16"].reshape(rows, -1), hid["fp32"].reshape(rows, -1), -1)
            e_alone = rel(alone.reshape(1, -1), hid["fp32"][:1].reshape(1, -1), -1)[0]
            e_alone_cls = rel(alone[:, 0], hid["fp32"][:1, 0], -1)[0]
            rec = dict(l=l, cls_vis=float(e_cls[0]), cls_masked=float(e_cls[1]), cls_rest=float(np.median(e_cls[2:])),
                       all_vis=float(e_all[0]), all_masked=float(e_all[1]), all_rest=float(np.median(e_all[2:])),
                       alone_all=float(e_alone), alone_cls=float(e_alone_cls))
            out["layers"].append(rec)
            print(f"layer {l:2d}: rel L2 err  CLS: visible {e_cls[0]:.2e} least-visible {e_cls[1]:.2e} rest(med) {np.median(e_cls[2:]):.2e} | "
                  f"all tokens: {e_all[0]:.2e} {e_all[1]:.2e} {np.median(e_all[2:]):.2e} | visible row alone (small path): all {e_alone:.2e} cls {e_alone_cls:.2e}")
        # heads, production path (cls_only_last; BERT: pruning unless AG_BERT_PRUNE=0)
        res = {}
        for prec in ("fp32", "bf16"):
            engine.set_precision(prec)
            v, _ = c["recipe"].fw_surrogate(srg, xs, masks)
            va, _ = c["recipe"].fw_surrogate(srg, xs, masks[:1])
            res[prec] = (v.float().cpu().numpy(), va.float().cpu().numpy())
        err = np.abs(res["bf16"][0] - res["fp32"][0]).max(1)
        err_alone = float(np.abs(res["bf16"][1] - res["fp32"][1]).max())
        spread = float(res["fp32"][0].max(0).max() - res["fp32"][0].min(0)[res["fp32"][0].max(0).argmax()])
        out["v_s"] = dict(visible=float(err[0]), masked=float(err[1]), rest_median=float(np.median(err[2:])), rest_max=float(err[2:].max()),
                          visible_alone=err_alone, spread_top_class=spread)
        print(f"v_s |bf16 - fp32| max over classes: visible row {err[0]:.2e} least-visible {err[1]:.2e} rest median {np.median(err[2:]):.2e} "
              f"max {err[2:].max():.2e} | visible row alone {err_alone:.2e} | spread of v_s over masks (top class) {spread:.3f}")
        order = np.argsort(vis)
        print("   err by visible count:", " ".join(f"{vis[i]}:{err[i]:.1e}" for i in order))
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "a") as f:
            f.write(json.dumps(out) + "\n")


if __name__ == "__main__":
    main()
