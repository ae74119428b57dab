#!/usr/bin/env python3
"""Dev tool: cProfile of the host side of the explainer training step (TB images, default 8): where the launch-bound part goes."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autognothi_amd import engine, training
from autognothi_amd.recipes import get_recipe
from autognothi_amd.scripts import train_explainer as te
from autognothi_amd.utils import synth
dev = torch.device("cuda:0")
kind, params, K = bench.WORKLOADS[os.environ.get("WL", "vit_base")]
B = int(os.environ.get("TB", 8))
training.MIXED_BF16 = os.environ.get("AG_TRAIN_BF16", "1") == "1"
recipe = get_recipe(kind); cfg = recipe.t_config(**params); P = recipe.n_players(cfg)
engine.set_precision("bf16")
srg = recipe.t_surrogate(cfg); synth.load_synth_weights(srg, seed=0); srg = srg.to(dev).eval()
exp = recipe.t_explainer(cfg); synth.load_synth_weights(exp, seed=1); exp = exp.to(dev); exp.train()
xs = torch.from_numpy(synth.synth_images(B, params["img_px_size"], params["img_channels"], seed=3)).to(dev)
opt = torch.optim.AdamW([q for q in exp.parameters() if q.requires_grad], lr=1e-5)
v0 = torch.full((1, cfg.num_labels), 0.1, device=dev)
gen = lambda a, b: (xs, torch.zeros(B, dtype=torch.long, device=dev))
te.explainer_epoch_train(None, dev, K, P, v0, [(None, None)] * 2, recipe, srg, exp, opt, 1, gen, seed=7)
torch.cuda.synchronize()
n = 5
pr = cProfile.Profile()
t = time.perf_counter(); pr.enable()
te.explainer_epoch_train(None, dev, K, P, v0, [(None, None)] * n, recipe, srg, exp, opt, 2, gen, seed=7)
pr.disable(); t_issue = time.perf_counter() - t
torch.cuda.synchronize(); t_all = time.perf_counter() - t
print(f"ms/step: host issue {t_issue / n * 1e3:.1f}  wall {t_all / n * 1e3:.1f}")
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
