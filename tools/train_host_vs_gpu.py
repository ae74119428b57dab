#!/usr/bin/env python3
"""Dev tool: is the explainer forward/backward host-bound?  Times trainer.loss_and_grads on fixed targets: host issue time (no
synchronisation until the end of N calls: what Python + launch overhead cost) against device time (events around the same calls),
eager vs replayed from a hipGraph when the trainer supports it."""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autognothi_amd import engine, ops, training as _tr, _lib as L
from autognothi_amd.recipes import get_recipe
from autognothi_amd.utils import synth
dev = torch.device("cuda:0")
WL = os.environ.get("WL", "vit_base")
kind, params, K = bench.WORKLOADS[WL]
B = int(os.environ.get("TB", 8))
recipe = get_recipe(kind); cfg = recipe.t_config(**params); P = recipe.n_players(cfg)
engine.set_precision("bf16"); _tr.MIXED_BF16 = True
exp = recipe.t_explainer(cfg); synth.load_synth_weights(exp, seed=1); exp = exp.to(dev); exp.train()
if kind.endswith("vit"):
    xs = torch.from_numpy(synth.synth_images(B, params["img_px_size"], params["img_channels"], seed=3)).to(dev)
else:
    xs = torch.from_numpy(synth.synth_token_ids(B, params["max_position_embeddings"], params["vocab_size"], seed=3)).to(dev)
C = cfg.num_labels
bits = ops.mask_shapley_new(ops.DeviceMT19937(dev, 1), B * K, P, want_i64=False, want_bits=True)[1]
v_s = torch.softmax(torch.randn(B * K, C, device=dev), -1); v_1 = torch.softmax(torch.randn(B, C, device=dev), -1)
v_0 = torch.full((1, C), 1.0 / C, device=dev)
labels = torch.zeros(B, dtype=torch.long, device=dev)
opt = torch.optim.AdamW([q for q in exp.parameters() if q.requires_grad], lr=1e-5, fused=True)
engine.watch_optimizer(opt)
tr = _tr.make_explainer_trainer(recipe, exp)
tr.use_graph = os.environ.get("AG_TRAIN_GRAPH", "0") == "1"


def step():
    opt.zero_grad()
    tr.loss_and_grads(xs, bits, v_0, v_s, v_1, K, labels=labels, train=True, seed=3)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
N = 10
l0 = int(L.lib().ag_launch_count())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(N):
    step()
e1.record(); t_issue = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(json.dumps({"workload": WL, "images": B, "explainer_step_ms": {"host_issue": round(1e3 * t_issue / N, 3), "wall": round(1e3 * t_all / N, 3),
                  "device_events": round(e0.elapsed_time(e1) / N, 3)}, "library_launches": (int(L.lib().ag_launch_count()) - l0) / N,
                  "side_stream": os.environ.get("AG_TRAIN_SIDE", "1"), "graph": os.environ.get("AG_TRAIN_GRAPH", "0")}))
