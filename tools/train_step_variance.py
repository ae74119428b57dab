#!/usr/bin/env python3
"""Dev tool: run-to-run spread of the training-step rate inside ONE process: EPOCHS timed epochs of 12 steps each (the bench's
train_step_rate body), ms/step per epoch, plus torch allocator statistics (hipMalloc calls inside the timed region)."""
import os, sys, time, json, gc, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autognothi_amd import training as _tr
from autognothi_amd.scripts import train_explainer as te
from autognothi_amd.utils import synth
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
wl = os.environ.get("WL", "vit_base"); tb = int(os.environ.get("TB", 8)); n_train = 12
job = bench.Job(wl, dev, 0, 1, tb, 0, "bf16")
_tr.MIXED_BF16 = True
recipe, cfg = job.recipe, job.cfg
m_exp = recipe.t_explainer(cfg); synth.load_synth_weights(m_exp, seed=1); m_exp = m_exp.to(dev).train()
tx = torch.from_numpy(job.inputs(tb, 200)).to(dev); labels = torch.zeros(tb, dtype=torch.long, device=dev)
opt = torch.optim.AdamW([q for q in m_exp.parameters() if q.requires_grad], lr=1e-5, fused=True)
v0 = torch.full((1, cfg.num_labels), 1.0 / cfg.num_labels, device=dev)
gen = lambda a, b_: (tx, labels)
te.explainer_epoch_train(None, dev, job.K, job.P, v0, [(None, None)] * 2, recipe, job.surrogate, m_exp, opt, 1, gen, seed=7)
torch.cuda.synchronize()
out = []
for ep in range(int(os.environ.get("EPOCHS", 8))):
    st0 = torch.cuda.memory_stats()
    g0 = [s["collections"] for s in gc.get_stats()]
    t = time.perf_counter()
    te.explainer_epoch_train(None, dev, job.K, job.P, v0, [(None, None)] * n_train, recipe, job.surrogate, m_exp, opt, 2 + ep, gen, seed=7)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    st1 = torch.cuda.memory_stats()
    g1 = [s["collections"] for s in gc.get_stats()]
    out.append({"epoch": ep, "ms_per_step": round(1e3 * el / n_train, 3), "device_mallocs": st1["num_device_alloc"] - st0["num_device_alloc"],
                "device_frees": st1["num_device_free"] - st0["num_device_free"], "alloc_retries": st1["num_alloc_retries"] - st0["num_alloc_retries"],
                "gc": [b - a for a, b in zip(g0, g1)], "reserved_gb": round(st1["reserved_bytes.all.current"] / 1e9, 2)})
    print(json.dumps(out[-1]), flush=True)
