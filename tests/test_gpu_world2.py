"""GPU, TWO real ranks on the box's one MI355X (collectives over gloo: a second RCCL rank cannot share a device): the REAL epoch body of
scripts/train_explainer.explainer_epoch_train with the REAL HIP kernels under row sharding, against its own single-process run (BASELINE
config 5: train_explainer on N GPUs; reference loop scripts/train_explainer.py:128-207).

tests/test_distributed_entrypoints_gloo.py checks the same control flow on the CPU with stub kernels; here the device sampler (this rank's
rows of the ONE global mask call, the other ranks' draws stepped over), the masked surrogate forward, the manual backward with the bucketed
gradient exchange from inside it, ragged shards (3 inputs on 2 ranks), a batch with FEWER inputs than ranks (K-within-image sharding: the
targets are gathered, every rank takes the same step) and the epoch-loss reduction all run for real.  Asserted: the epoch figure and the
parameters after four SGD steps equal the world-1 run to fp32 rounding (dropout off: with it on, ranks key their keep decisions on their
first input — a different, equally distributed draw)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BATCHES = [4, 4, 3, 1]
SEED, EPOCH = 4321, 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(rank, world, port, out_dir, which="explainer"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from util import golden_json
    from autognothi_amd import engine
    from autognothi_amd.recipes import get_recipe
    from autognothi_amd.scripts import train_explainer as te
    from autognothi_amd.utils import synth
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        duo = which == "duo"
        meta = golden_json("model_duo_vit_tiny_l3.json" if duo else "model_vit_tiny_c1.json")
        prm = dict(meta["params"], hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        recipe = get_recipe("duo_vanilla_vit" if duo else "vanilla_vit")
        cfg = recipe.t_config(**prm)
        srg, exp = recipe.t_surrogate(cfg), recipe.t_explainer(cfg)
        synth.load_synth_weights(srg, seed=0)
        synth.load_synth_weights(exp, seed=1)
        srg, exp = srg.to(dev).eval(), exp.to(dev)
        engine.set_precision("fp32")
        k, p = 4, recipe.n_players(cfg)
        base = torch.from_numpy(synth.synth_images(4, prm["img_px_size"], prm["img_channels"], seed=3)).to(dev)
        batches = [(base[:b] * (1.0 - 0.1 * i) + 0.05 * i, torch.zeros(b, dtype=torch.long, device=dev)) for i, b in enumerate(BATCHES)]
        with torch.no_grad():
            v_0, _ = recipe.fw_surrogate(srg, torch.zeros_like(base[:1]), torch.ones((1, p), dtype=torch.long, device=dev))
        lines = []

        class Env:
            def log(self, msg):
                lines.append(msg)

        items = [(i, None) for i in range(len(BATCHES))]
        if which == "faithfulness":    # three single-image samples on two ranks: one full group (by image) + a tail sharded inside the image
            import json
            from autognothi_amd.scripts import measure_faithfulness as mf
            fin = recipe.t_final(cfg)
            synth.load_synth_weights(fin, seed=2)
            fin = fin.to(dev).eval()
            samples = [(base[i:i + 1] * (1.0 + 0.1 * i), torch.tensor([i % 3], device=dev)) for i in range(3)]
            rep = mf.measure_faithfulness_loaded(Env(), dev, recipe, srg, fin, samples, lambda a, b_: (a, b_), 8)
            torch.cuda.synchronize()
            if rank == 0:
                with open(os.path.join(out_dir, f"{which}_world{world}.json"), "w") as f:
                    json.dump({"report": rep, "lines": lines}, f)
            return
        if which == "surrogate":       # the surrogate is the trained model, a classifier of the same architecture gives the targets
            from autognothi_amd.scripts import train_surrogate as ts
            cls = recipe.t_classifier(cfg)
            synth.load_synth_weights(cls, seed=2)
            cls = cls.to(dev).eval()
            srg.train()
            params = [q for q in srg.parameters() if q.requires_grad]
            before = [q.detach().clone() for q in params]
            opt = torch.optim.SGD(params, lr=2e-3)
            got = ts.surrogate_epoch_train(Env(), dev, p, items, recipe, cls, srg, opt, EPOCH, lambda a, b_: batches[a], seed=SEED)
        else:
            params = [q for q in exp.parameters() if q.requires_grad]
            before = [q.detach().clone() for q in params]
            opt = torch.optim.SGD(params, lr=2e-3)
            if duo:
                from autognothi_amd.scripts import train_duo_explainer as td
                got = td.duo_explainer_epoch_train(Env(), dev, k, p, v_0, items, recipe, srg, exp, opt, EPOCH, lambda a, b_: batches[a], seed=SEED)
                got = float(got[2])        # (cls loss, reg loss, total loss, accuracy)
            else:
                got = te.explainer_epoch_train(Env(), dev, k, p, v_0, items, recipe, srg, exp, opt, EPOCH, lambda a, b_: batches[a], seed=SEED)
        torch.cuda.synchronize()
        if rank == 0:
            moved = max(float((q.detach() - q0).abs().max()) for q, q0 in zip(params, before))
            np.savez(os.path.join(out_dir, f"{which}_world{world}.npz"), loss=np.asarray([got]), n_lines=np.asarray([len(lines)]), moved=np.asarray([moved]),
                     **{f"p{i}": q.detach().cpu().numpy() for i, q in enumerate(params)})
        else:
            assert lines == []              # (rank-0-only logging)
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("which", ["explainer", "duo", "surrogate"])
def test_sharded_epoch_on_two_ranks_equals_one_rank(tmp_path, which):
    """explainer: scripts/train_explainer.explainer_epoch_train; duo: train_duo_explainer.duo_explainer_epoch_train (the cross-entropy head rides
    along); surrogate: train_surrogate.surrogate_epoch_train (ONE uniform mask per input: the one-input batch leaves a rank without rows — an
    empty shard that still enters every collective)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for world in (1, 2):
        port = _free_port()
        procs = [ctx.Process(target=_run, args=(r, world, port, str(tmp_path), which)) for r in range(world)]
        for pr in procs:
            pr.start()
        for pr in procs:
            pr.join(600)
            assert pr.exitcode == 0, f"world {world}: a rank failed (exit code {pr.exitcode})"
    one, two = np.load(tmp_path / f"{which}_world1.npz"), np.load(tmp_path / f"{which}_world2.npz")
    assert abs(float(one["loss"][0]) - float(two["loss"][0])) <= 1e-5 * abs(float(one["loss"][0])) + 1e-9
    assert int(one["n_lines"][0]) == int(two["n_lines"][0]) > 0
    keys = [k_ for k_ in one.files if k_[0] == "p" and k_[1:].isdigit()]
    for k_ in keys:
        scale = float(np.abs(one[k_]).max()) + 1e-12
        np.testing.assert_allclose(two[k_], one[k_], rtol=1e-4, atol=1e-5 * scale, err_msg=k_)
    assert len(keys) > 50 and float(one["moved"][0]) > 1e-6          # (four SGD steps did move the parameters)


def test_faithfulness_loop_on_two_ranks_equals_one_rank(tmp_path):
    """scripts/measure_faithfulness.measure_faithfulness_loaded (reference :195-218) with the real fw_final / fw_surrogate: samples by rank, the
    short last group by perturbation row inside the image (rows all-gathered), curves gathered and re-ordered: the single-process report."""
    import json
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for world in (1, 2):
        port = _free_port()
        procs = [ctx.Process(target=_run, args=(r, world, port, str(tmp_path), "faithfulness")) for r in range(world)]
        for pr in procs:
            pr.start()
        for pr in procs:
            pr.join(600)
            assert pr.exitcode == 0, f"world {world}: a rank failed (exit code {pr.exitcode})"
    one, two = (json.load(open(tmp_path / f"faithfulness_world{w}.json")) for w in (1, 2))
    assert one["report"]["data_cls"] == two["report"]["data_cls"] and len(one["report"]["data_cls"]) == 3
    assert one["lines"] == two["lines"] and len(one["lines"]) == 3
    for key in ("insertion", "deletion", "insertion_non_ok", "deletion_non_ok"):
        assert abs(one["report"][key]["auc"] - two["report"][key]["auc"]) <= 1e-6, key
    for a, b in zip(one["report"]["data_ins"] + one["report"]["data_del"], two["report"]["data_ins"] + two["report"]["data_del"]):
        assert a.keys() == b.keys()
        for cl in a:
            np.testing.assert_allclose(list(b[cl].values()), list(a[cl].values()), rtol=1e-5, atol=1e-7)
