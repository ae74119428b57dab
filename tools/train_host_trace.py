#!/usr/bin/env python3
"""Dev tool: which Python call sites launch the torch-side fill / copy kernels of a training step (torch.profiler with stacks)."""
import collections, os, sys, torch
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
engine.set_precision("bf16"); _tr.MIXED_BF16 = True
srg = recipe.t_surrogate(cfg); synth.load_synth_weights(srg, seed=0); srg = srg.to(dev).eval()
exp = recipe.t_explainer(cfg); synth.load_synth_weights(exp, seed=1); exp = exp.to(dev); exp.train()
if kind.endswith("vit"):
    xs = torch.from_numpy(synth.synth_images(B, params["img_px_size"], params["img_channels"], seed=3)).to(dev)
else:
    xs = torch.from_numpy(synth.synth_token_ids(B, params["max_position_embeddings"], params["vocab_size"], seed=3)).to(dev)
opt = torch.optim.AdamW([q for q in exp.parameters() if q.requires_grad], lr=1e-5, fused=True)
v0 = torch.full((1, cfg.num_labels), 0.1, device=dev)
gen = lambda a, b: (xs, torch.zeros(B, dtype=torch.long, device=dev))
te.explainer_epoch_train(None, dev, K, P, v0, [(None, None)] * 2, recipe, srg, exp, opt, 1, gen, seed=7)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    te.explainer_epoch_train(None, dev, K, P, v0, [(None, None)] * 2, recipe, srg, exp, opt, 2, gen, seed=7)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::clone", "aten::zeros", "aten::ones", "aten::cat", "aten::index", "aten::_to_copy"):
        st = [s for s in (ev.stack or []) if "autognothi_amd" in s or "bench" in s or "torch/optim" in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (name, site), n in cnt.most_common(40):
    print(f"{n:5d} {name:14s} {site}")
