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
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
cnt = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        site = "?"
        for fr in reversed(traceback.extract_stack()):
            if ("autognothi_amd" in fr.filename or "torch/optim" in fr.filename) and "train_host_trace" not in fr.filename:
                site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                break
        cnt[(name, site)] += 1
        return func(*args, **(kwargs or {}))


STEPS = 2
with Log():
    te.explainer_epoch_train(None, dev, K, P, v0, [(None, None)] * STEPS, recipe, srg, exp, opt, 2, gen, seed=7)
    torch.cuda.synchronize()
print("aten ops per step by call site (views / metadata ops excluded):")
skip = ("view", "detach", "reshape", "slice", "select", "t.default", "transpose", "unsqueeze", "expand", "alias", "_unsafe_view", "as_strided",
        "empty", "size", "stride", "is_", "permute", "squeeze", "_local_scalar", "split", "unbind", "record_stream", "lift_fresh", "numel")
for (name, site), n in cnt.most_common(200):
    if any(k in name for k in skip):
        continue
    print(f"{n / STEPS:7.1f} {name:40s} {site}")
