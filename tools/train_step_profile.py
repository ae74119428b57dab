#!/usr/bin/env python3
"""Dev tool: run a few explainer training steps (WL = a bench.py workload, TB images per step, K masks) for
rocprofv3 --kernel-trace --stats; prints wall ms/step so that the kernel-time sum can be set against it."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autognothi_amd import engine, training as _tr
from autognothi_amd.recipes import get_recipe
from autognothi_amd.scripts import train_explainer as te
from autognothi_amd.utils import synth
dev = torch.device("cuda:0")
kind, params, K = bench.WORKLOADS[os.environ.get("WL", "vit_base")]
B = int(os.environ.get("TB", 8))
recipe = get_recipe(kind); cfg = recipe.t_config(**params); P = recipe.n_players(cfg)
engine.set_precision("bf16")
_tr.MIXED_BF16 = os.environ.get("MIXED", "1") != "0"
srg = recipe.t_surrogate(cfg); synth.load_synth_weights(srg, seed=0); srg = srg.to(dev).eval()
exp = recipe.t_explainer(cfg); synth.load_synth_weights(exp, seed=1); exp = exp.to(dev); exp.train()
if kind.endswith("vit"):
    xs = torch.from_numpy(synth.synth_images(B, params["img_px_size"], params["img_channels"], seed=3)).to(dev)
else:
    xs = torch.from_numpy(synth.synth_token_ids(B, params["max_position_embeddings"], params["vocab_size"], seed=3)).to(dev)
opt = torch.optim.AdamW([q for q in exp.parameters() if q.requires_grad], lr=1e-5, fused=True)
v0 = torch.full((1, cfg.num_labels), 0.1, device=dev)
gen = lambda a, b: (xs, torch.zeros(B, dtype=torch.long, device=dev))
te.explainer_epoch_train(None, dev, K, P, v0, [(None, None)], recipe, srg, exp, opt, 1, gen, seed=7)
torch.cuda.synchronize(); t = time.perf_counter()
te.explainer_epoch_train(None, dev, K, P, v0, [(None, None)] * 6, recipe, srg, exp, opt, 2, gen, seed=7)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t) / 6 * 1e3)
