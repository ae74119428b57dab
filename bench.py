#!/usr/bin/env python3
"""bench.py — masked-forwards/sec of the K-mask surrogate forward (BASELINE.json metric) on MI355X.

A "step" = one pass of the hot path over one batch of synthetic inputs resident in HBM:
  device mask sampler (mask_shapley_new, B*K rows) -> masked ViT-base surrogate forward for the
  B*K rows (K masks share each input's embeddings / layer-0 LN+QKV) -> v_s [B*K, C] on device.
Workload (config.workload) = BASELINE.json configs[1]: vit_base_imagenette_vanilla, K=32, bf16.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload vit_base|bert_base|vit_large|vit_tiny]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: rows shard by image (each rank owns B images x all K masks; weights replicated; masks
come from per-rank device generators) — no data-path collective, weak scaling.  Rank 0 prints ONE
JSON line.  `roofline` is measured live with hipEvents around every launch of the dominant kernel
inside the timed region; `cpu_baseline` times the torch-CPU port of the reference path (oracle/torch_port.py)
on a bounded sample on the host cores (rank 0, N=1).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from autognothi_amd import _lib as L  # noqa: E402
from autognothi_amd import engine, ops  # noqa: E402
from autognothi_amd.recipes import get_recipe  # noqa: E402
from autognothi_amd.utils import synth  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (2.5 PF)
PEAK_F32_TFLOPS = 157.3     # f32-input MFMA peak, same guide

WORKLOADS = {
    # name: (recipe kind, experiment params, BASELINE K)
    "vit_base": ("vanilla_vit", dict(attention_probs_dropout_prob=0.1, explainer_attn_num_layers=1, explainer_head_hidden_size=3072,
                                     explainer_normalize=True, hidden_dropout_prob=0.1, hidden_size=768, intermediate_size=3072,
                                     layer_norm_eps=1e-12, num_attention_heads=12, num_hidden_layers=12, num_labels=10,
                                     img_channels=3, img_px_size=224, img_patch_size=16), 32),
    "vit_large": ("vanilla_vit", dict(attention_probs_dropout_prob=0.1, explainer_attn_num_layers=1, explainer_head_hidden_size=4096,
                                      explainer_normalize=True, hidden_dropout_prob=0.1, hidden_size=1024, intermediate_size=4096,
                                      layer_norm_eps=1e-12, num_attention_heads=16, num_hidden_layers=24, num_labels=10,
                                      img_channels=3, img_px_size=224, img_patch_size=16), 64),
    "vit_tiny": ("vanilla_vit", dict(attention_probs_dropout_prob=0.1, explainer_attn_num_layers=1, explainer_head_hidden_size=768,
                                     explainer_normalize=True, hidden_dropout_prob=0.1, hidden_size=192, intermediate_size=768,
                                     layer_norm_eps=1e-12, num_attention_heads=3, num_hidden_layers=12, num_labels=10,
                                     img_channels=3, img_px_size=224, img_patch_size=16), 4),
    "bert_base": ("vanilla_bert", dict(attention_probs_dropout_prob=0.1, explainer_attn_num_layers=1, explainer_head_hidden_size=3072,
                                       explainer_normalize=True, hidden_dropout_prob=0.1, hidden_size=768, intermediate_size=3072,
                                       layer_norm_eps=1e-12, max_position_embeddings=128, num_attention_heads=12,
                                       num_hidden_layers=12, num_labels=2, pad_token_id=0, type_vocab_size=2, vocab_size=30522), 32),
}
# LTT (ladder side network): the shipped ladder width (96 = 12 heads of 8, 384 intermediate,
# experiments/bert_base_tayp_ltt/.hparams.json) on the base backbones; fw_surrogate = frozen backbone + ladder 0
_LTT = dict(explainer_s_attn_num_layers=1, explainer_s_head_hidden_size=3072, s_attn_hidden_size=96, s_attn_intermediate_size=384)
WORKLOADS["ltt_vit_base"] = ("ltt_vit", dict({k: v for k, v in WORKLOADS["vit_base"][1].items() if not k.startswith("explainer_")},
                                              explainer_normalize=True, **_LTT), 32)
WORKLOADS["ltt_bert_base"] = ("ltt_bert", dict({k: v for k, v in WORKLOADS["bert_base"][1].items() if not k.startswith("explainer_")},
                                                explainer_normalize=True, **_LTT), 32)

EPI_NAMES = {0: "gemm<bias>", 1: "gemm<bias+gelu>", 2: "gemm<bias+residual>", 3: "gemm<bias,f32out>", 4: "gemm<bias+tanh>",
             5: "gemm<bias+gelu+add>", 8: "masked_attention", 9: "layernorm"}


def flops_per_forward(kind, p, T):
    """F_ref: GEMM flops (2/MAC) of one masked forward as the reference executes it (SURVEY.md §8a footer):
    L*(8TH^2 + 4T^2H + 4THI) + embed + head."""
    H, I, Lr, C_ = p["hidden_size"], p["intermediate_size"], p["num_hidden_layers"], p["num_labels"]
    f = Lr * (8 * T * H * H + 4 * T * T * H + 4 * T * H * I)
    if kind.endswith("vit"):
        f += 2 * (T - 1) * (p["img_channels"] * p["img_patch_size"] ** 2) * H + 2 * H * C_
    else:
        f += 2 * H * H + 2 * H * C_
    if kind.startswith("ltt_"):   # one ladder: per backbone layer a map H->h and an h-wide layer; side head
        h, i_s = p["s_attn_hidden_size"], p["s_attn_intermediate_size"]
        f += Lr * (2 * T * H * h + 8 * T * h * h + 4 * T * T * h + 4 * T * h * i_s) + 2 * h * C_ + (0 if kind.endswith("vit") else 2 * h * h)
    return float(f)


def flops_executed(kind, p, T, K, frac=1.0):
    """F_exec: F_ref minus work legitimately skipped per row: embed + layer-0 QKV shared across the K masks,
    last layer's Q-projection/attention/out-proj/MLP on the CLS token only; BERT token pruning (frac = visible
    tokens / all tokens, measured): layers 1.. run on the packed rows (GEMMs x frac, attention ~ x frac^2)."""
    H, I, Lr = p["hidden_size"], p["intermediate_size"], p["num_hidden_layers"]
    if kind == "vanilla_bert" and frac < 1.0 and Lr >= 2:
        layer = 8 * T * H * H + 4 * T * T * H + 4 * T * H * I
        f = layer - 6 * T * H * H * (K - 1) / K                                   # layer 0: every token, shared QKV
        f += (Lr - 2) * (frac * (8 * T * H * H + 4 * T * H * I) + frac * frac * 4 * T * T * H)
        f += frac * 6 * T * H * H + 4 * frac * T * H + 2 * H * H + 4 * H * I     # last: QKV packed, the rest CLS only
        f += 2 * H * H + 2 * H * p["num_labels"]
        return float(f)
    if kind == "ltt_bert" and frac < 1.0 and Lr >= 2:                             # backbone + ladder, all packed after layer 0
        h, i_s = p["s_attn_hidden_size"], p["s_attn_intermediate_size"]
        layer = 8 * T * H * H + 4 * T * T * H + 4 * T * H * I
        side = 2 * T * H * h + 8 * T * h * h + 4 * T * T * h + 4 * T * h * i_s
        lin = lambda x, quad: frac * (x - quad) + frac * frac * quad               # noqa: E731  (GEMMs ~ frac, attention ~ frac^2)
        f = layer - 6 * T * H * H * (K - 1) / K + side
        f += (Lr - 1) * (lin(layer, 4 * T * T * H) + lin(side, 4 * T * T * h))
        f += 2 * H * H + 2 * H * p["num_labels"] + 2 * h * h + 2 * h * p["num_labels"]
        return float(f)
    f = flops_per_forward(kind, p, T)
    shared = 6 * T * H * H + (2 * (T - 1) * (p["img_channels"] * p["img_patch_size"] ** 2) * H if kind.endswith("vit") else 0)
    f -= shared * (K - 1) / K
    if not kind.startswith("ltt_"):   # (the ladder taps every token of every layer: nothing to skip there)
        # last layer: attention for 1 query instead of T, out-proj + MLP for 1 token instead of T (QKV still full)
        f -= (4 * T * T * H + 2 * T * H * H + 4 * T * H * I) * (T - 1) / T
    return float(f)


def collect(cls):
    ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    L.check(L.lib().ag_profile_collect(cls, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)))
    return ms.value, fl.value, by.value, n.value


def cpu_baseline(kind, params, xs_np, masks_np, sd_np):
    """The torch-CPU port of the reference path (oracle/torch_port.py, fp32, all host cores) timed over a
    bounded sample of the same workload, the K masked copies materialised as the reference does
    (scripts/train_explainer.py:159-163)."""
    from oracle import torch_port as otp
    if kind.startswith("ltt_"):
        def fn(x, m, sd_, prm):
            with torch.no_grad():
                return otp.ltt_surrogate_probs(x, m, sd_, prm, "vit" if kind.endswith("vit") else "bert")
    else:
        fn = otp.vit_surrogate if kind == "vanilla_vit" else otp.bert_surrogate
    rows = masks_np.shape[0]
    xs_ext = torch.from_numpy(np.repeat(xs_np, rows // xs_np.shape[0], axis=0))
    masks = torch.from_numpy(masks_np)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    # a 100+-core host oversubscribes these medium-sized GEMMs: time a few thread counts, report the best
    best, best_threads, reps_total = float("inf"), torch.get_num_threads(), 0
    t0 = time.perf_counter()
    ncpu = os.cpu_count() or 8
    prev = torch.get_num_threads()
    for nt in sorted({min(ncpu, n) for n in (8, 16, 32, 64)}):
        if time.perf_counter() - t0 > 22.0:
            break
        torch.set_num_threads(nt)
        fn(xs_ext, masks, sd, params)  # warm-up (thread pool, page-in)
        for _ in range(2):
            t1 = time.perf_counter()
            fn(xs_ext, masks, sd, params)
            dt = time.perf_counter() - t1
            reps_total += 1
            if dt < best:
                best, best_threads = dt, nt
    torch.set_num_threads(prev)
    return rows / best, rows, reps_total, best_threads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    # 48 inputs x 32 masks x 197 tokens = 1182 M-tiles of 256 rows: 3546 / 10638 / 14184 tiles for the N = 768 / 2304 / 3072
    # GEMMs = 13.85 / 41.6 / 55.4 rounds of 256 CUs (<= 1.1 % idle in the last round; B=16 loses 7.7 % on the N = 768 ones)
    ap.add_argument("--batch", type=int, default=48, help="images (or sequences) per GPU per step")
    ap.add_argument("--workload", default="vit_base", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--masks", type=int, default=0, help="K masks per input (default: the BASELINE config's K)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-batch", type=int, default=8, help="images per GPU per explainer training step of the secondary block (0 = skip)")
    # 220 images x 197 tokens = 170 M-tiles: 510 / 1530 / 2040 tiles for N = 768 / 2304 / 3072 = 1.99 / 5.98 / 7.97 rounds of 256 CUs
    # (128 images leave 42 % of the second round of the N = 768 GEMMs idle: 6.3 k -> 7.6 k attributions/s)
    ap.add_argument("--attr-batch", type=int, default=220, help="images per GPU per fw_final pass of the secondary metric (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run --nproc-per-node N (one process per GPU)")
        args.gpus = world
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)  # "nccl" is RCCL on ROCm
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    kind, params, k_default = WORKLOADS[args.workload]
    K = args.masks or k_default
    B = args.batch
    recipe = get_recipe(kind)
    cfg = recipe.t_config(**params)
    P = recipe.n_players(cfg)
    T = P + 1
    engine.set_precision(args.precision)

    surrogate = recipe.t_surrogate(cfg)
    synth.load_synth_weights(surrogate, seed=0)   # random-init weights of the named architecture (no network)
    surrogate = surrogate.to(dev).eval()
    if kind.endswith("vit"):
        xs_np = synth.synth_images(B, params["img_px_size"], params["img_channels"], seed=rank)
    else:
        xs_np = synth.synth_token_ids(B, params["max_position_embeddings"], params["vocab_size"], seed=rank)
    xs = torch.from_numpy(xs_np).to(dev)
    rng = ops.DeviceMT19937(dev, 3407 + rank)
    R = B * K

    def step():
        _, bits = ops.mask_shapley_new(rng, R, P, want_i64=False, want_bits=True)
        with torch.no_grad():
            v_s, _ = recipe.fw_surrogate(surrogate, xs, bits)
        return v_s

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    L.check(L.lib().ag_profile_enable(1))
    for c in EPI_NAMES:
        collect(c)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    L.check(L.lib().ag_profile_enable(0))
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert bool(torch.isfinite(out).all())

    stats = {c: collect(c) for c in EPI_NAMES}
    packed_rows = engine.LAST_PACKED_ROWS   # (BERT token pruning) visible tokens of the last timed step

    # ---- secondary metric of BASELINE.json: Shapley attributions per second through fw_final (classifier +
    # surrogate + explainer forwards on all-ones masks -> phi [B, C, P]); untimed by the contract's K steps.
    attrs_per_s = None
    if args.attr_batch > 0:
        final = recipe.t_final(cfg)
        synth.load_synth_weights(final, seed=1)
        final = final.to(dev).eval()
        if kind.endswith("vit"):
            fx_np = synth.synth_images(args.attr_batch, params["img_px_size"], params["img_channels"], seed=100 + rank)
        else:
            fx_np = synth.synth_token_ids(args.attr_batch, params["max_position_embeddings"], params["vocab_size"], seed=100 + rank)
        fx = torch.from_numpy(fx_np).to(dev)
        with torch.no_grad():
            for _ in range(2):
                _, phi = recipe.fw_final(final, fx)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            ta = time.perf_counter()
            n_attr = 4
            for _ in range(n_attr):
                _, phi = recipe.fw_final(final, fx)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            el = time.perf_counter() - ta
        assert phi.shape == (args.attr_batch, cfg.num_labels, P) and bool(torch.isfinite(phi).all())
        if dist is not None:
            tm = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            el = float(tm.item())
        attrs_per_s = args.attr_batch * world * n_attr / el
        del final, fx, phi

    # ---- training-step rate (SURVEY §8d metric 2): one reference _explainer_epoch_train body per step = K-mask
    # surrogate targets (bf16 inference path) + explainer forward/backward (fp32 training kernels) + AdamW step
    train_imgs_per_s = None
    if args.train_batch > 0:
        from autognothi_amd import training as _tr
        from autognothi_amd.scripts import train_explainer as te
        _tr.MIXED_BF16 = args.precision == "bf16"   # throughput mode: bf16 GEMM operands (autocast semantics), fp32 everything else
        m_exp = recipe.t_explainer(cfg)
        synth.load_synth_weights(m_exp, seed=1)
        m_exp = m_exp.to(dev)
        m_exp.train()
        tb = args.train_batch
        if kind.endswith("vit"):
            tx = torch.from_numpy(synth.synth_images(tb, params["img_px_size"], params["img_channels"], seed=200 + rank)).to(dev)
        else:
            tx = torch.from_numpy(synth.synth_token_ids(tb, params["max_position_embeddings"], params["vocab_size"], seed=200 + rank)).to(dev)
        labels = torch.zeros(tb, dtype=torch.long, device=dev)
        opt = torch.optim.AdamW([q for q in m_exp.parameters() if q.requires_grad], lr=1e-5)
        v0 = torch.full((1, cfg.num_labels), 1.0 / cfg.num_labels, device=dev)
        gen = lambda a, b_: (tx, labels)  # noqa: E731
        te.explainer_epoch_train(None, dev, K, P, v0, [(None, None)], recipe, surrogate, m_exp, opt, 1, gen, seed=7)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        tt = time.perf_counter()
        n_train = 3
        te.explainer_epoch_train(None, dev, K, P, v0, [(None, None)] * n_train, recipe, surrogate, m_exp, opt, 2, gen, seed=7)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - tt
        if dist is not None:
            tm = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            el = float(tm.item())
        train_imgs_per_s = tb * world * n_train / el
        surrogate.eval()
        del m_exp, opt
    if rank == 0:
        total_rows = R * world * args.steps
        value = total_rows / elapsed
        frac = packed_rows / float(R * T) if (kind in ("vanilla_bert", "ltt_bert") and packed_rows) else 1.0
        f_ref, f_exec = flops_per_forward(kind, params, T), flops_executed(kind, params, T, K, frac)
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
        # dominant kernel = the instrumented class with the largest total time
        dom = max(stats, key=lambda c: stats[c][0])
        ms, fl, by, n = stats[dom]
        per_kernel = {EPI_NAMES[c]: {"launches": int(s[3]), "avg_us": round(1e3 * s[0] / max(1, s[3]), 2),
                                     "tflops": round(s[1] / max(s[0], 1e-9) / 1e9, 1),
                                     "algo_gb_s": round(s[2] / max(s[0], 1e-9) / 1e6, 1)}
                      for c, s in stats.items() if s[3]}
        traffic = None
        try:  # measured in a separate rocprofv3 --pmc run of this same command (tools/profile_round.sh), committed
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if (tj["workload"], tj["batch"], tj["masks"], tj["precision"]) == (args.workload, B, K, args.precision):
                traffic = tj["traffic_bytes_per_launch"].get(EPI_NAMES[dom])
        except (OSError, KeyError, ValueError):
            pass
        roofline = {
            "bound": "mfma", "kernel": EPI_NAMES[dom], "launches": int(n), "avg_launch_us": round(1e3 * ms / max(1, n), 2),
            "achieved": round(fl / max(ms, 1e-9) / 1e9, 1), "peak": peak, "unit": "TFLOP/s",
            "frac": round(fl / max(ms, 1e-9) / 1e9 / peak, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": round(by / max(1, n)),
            "whole_step": {"f_ref_gflop_per_fwd": round(f_ref / 1e9, 3), "f_exec_gflop_per_fwd": round(f_exec / 1e9, 3),
                           "ref_equiv_tflops": round(value / world * f_ref / 1e12, 1),
                           "exec_tflops": round(value / world * f_exec / 1e12, 1),
                           "exec_frac_of_peak": round(value / world * f_exec / 1e12 / peak, 4)},
            "kernels": per_kernel,
        }
        if args.precision == "bf16":
            # what the matrix cores of this board sustain on random operands with nothing else running (power cap):
            # context for `frac`, which stays priced against the 2.4 GHz datasheet peak
            tf, ghz = C.c_double(), C.c_double()
            L.check(L.lib().ag_probe_mfma(100000, 0, C.byref(tf), C.byref(ghz), None))
            roofline["power_capped_mfma_probe"] = {"tflops": round(tf.value, 1), "shader_ghz": round(ghz.value, 3),
                                                   "frac_of_probe": round(fl / max(ms, 1e-9) / 1e9 / max(tf.value, 1e-9), 4)}
        line = {
            "metric": "masked-forwards/sec (K=%d)" % K, "value": round(value, 2), "unit": "masked-forwards/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": {"vit_base": "vit_base_imagenette_vanilla", "vit_large": "vit_large_imagenette_vanilla",
                                    "vit_tiny": "vit_tiny_imagenette_vanilla", "bert_base": "bert_base_tayp_vanilla seq_len=128",
                                    "ltt_vit_base": "vit_base_imagenette + LTT ladder (h=96)", "ltt_bert_base": "bert_base_tayp_ltt seq_len=128"}[args.workload],
                       "masks_per_input": K, "inputs_per_gpu_per_step": B, "rows_per_step": R * world, "tokens": T,
                       "sharding": "rows by input, no data-path collective", "weights": "seeded random init",
                       "visible_token_fraction_after_layer0": round(frac, 4)},
            "roofline": roofline,
        }
        if attrs_per_s is not None:
            line["secondary"] = {"metric": "Shapley-attrs/sec/image", "value": round(attrs_per_s, 1), "unit": "images/s",
                                 "path": "fw_final (classifier + surrogate + explainer forward -> phi[B,C,P])",
                                 "images_per_gpu_per_pass": args.attr_batch, "passes": 4}
            if train_imgs_per_s is not None:
                line["secondary"]["train_explainer_step"] = {
                    "value": round(train_imgs_per_s, 1), "unit": "images/s", "masks_per_image": K, "images_per_gpu_per_step": args.train_batch,
                    "body": "K-mask surrogate targets (bf16) + explainer fwd/bwd (bf16 GEMM and attention operands on the matrix cores, fp32 accumulate / activations / optimizer state) + AdamW, as scripts/train_explainer.py:128-207"}
        if world == 1 and not args.no_cpu_baseline:
            sample_b = 1
            masks_np = ops.mask_shapley_new(ops.DeviceMT19937(dev, 3407), sample_b * K, P)[0].cpu().numpy()
            sd_np = {k: v.detach().cpu().numpy() for k, v in surrogate.state_dict().items()}
            cpu_v, cpu_rows, cpu_reps, cpu_threads = cpu_baseline(kind, params, xs_np[:sample_b], masks_np, sd_np)
            line["cpu_baseline"] = {"value": round(cpu_v, 2), "unit": "masked-forwards/s", "cores": cpu_threads,
                                    "kind": "port",
                                    "sample": f"torch-CPU fp32 port of the reference path, {cpu_rows} rows (1 input x K={K}) of the same workload, best of {cpu_reps} runs over 8/16/32/64 threads (host has {os.cpu_count()} logical CPUs)"}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
