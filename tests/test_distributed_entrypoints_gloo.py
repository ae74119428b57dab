"""CPU, world_size 2, gloo: the REAL epoch bodies of the pipeline entry points (scripts/train_explainer.explainer_epoch_train,
scripts/train_surrogate.surrogate_epoch_train, scripts/measure_faithfulness.measure_faithfulness_loaded) under row sharding,
against their own single-process run (BASELINE config 5: train_explainer on N GPUs; reference loop
scripts/train_explainer.py:128-207, scripts/train_surrogate.py:112-160, scripts/measure_faithfulness.py:195-218).

The kernels are stubbed (plain torch on the CPU: a stub trainer behind ``module._ag_trainer``, a stub ``fw_surrogate`` /
``fw_classifier``, the numpy oracle's bit-exact sampler behind the MaskSource contract) — what runs for real is the control
flow under test: input slices per rank, this rank's rows of the ONE global mask call, weighted bucketed gradient exchange
from inside the backward, batches with FEWER inputs than ranks sharded by mask inside every input (no rank idles, the
targets are gathered, no gradient travels), empty shards where that is impossible (the surrogate's one mask per input), the
once-per-epoch loss reduction, rank-0-only logging.
Asserted: masks (bit-exact, union of the ranks == the single-process stream), per-batch losses, the epoch figure and the
post-step parameters equal the world-1 run."""
import os
import socket
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

K, P, C, DFEAT = 4, 6, 3, 8
BATCHES = [4, 4, 3, 1, 2, 1]       # 3 -> ragged shards (2 + 1); 1 -> fewer inputs than ranks: K-within-image sharding (explainer) /
                                   #      one rank without inputs (surrogate: a single mask per input cannot be split)
SEED = 1234


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleMaskSource:
    """scripts/common.MaskSource on the numpy oracle (oracle/shapley.py, pinned bit-exact to the reference's sampler): the
    state advances by the WHOLE global call, the caller gets rows of inputs [lo, hi).  'bits' are the int64 masks here."""

    def __init__(self, seed):
        from oracle.mt19937 import MT19937
        from util import golden
        self.gen = MT19937(seed)
        self.prefix = golden("prefix_tables.npz")
        self.log = []

    def _prefix(self, p):
        from oracle import shapley as osh
        key = f"prefix_{p}"
        return self.prefix[key] if key in self.prefix.files else osh.shapley_prefix_table(p)

    def shapley(self, n_tot, lo, hi, k, p):
        from oracle import shapley as osh
        rows = osh.mask_shapley_new(n_tot * k, p, self.gen, self._prefix(p))[lo * k:hi * k]
        self.log.append(("shapley", n_tot, rows.copy()))
        return torch.from_numpy(rows)

    def uniform(self, n_tot, lo, hi, p):
        from oracle import shapley as osh
        rows = osh.mask_purely_uniform(n_tot, p, self.gen)[lo:hi]
        self.log.append(("uniform", n_tot, rows.copy()))
        return torch.from_numpy(rows)


def _data():
    g = torch.Generator().manual_seed(5)
    return [(torch.randn(b, DFEAT, generator=g), torch.randint(0, C, (b,), generator=g)) for b in BATCHES]


class _Recipe:
    def __init__(self):
        g = torch.Generator().manual_seed(6)
        self.w_srg = torch.randn(DFEAT + P, C, generator=g)
        self.w_cls = torch.randn(DFEAT, C, generator=g)
        self.rows = []         # (inputs, mask rows) of every masked forward this rank ran

    def fw_surrogate(self, model, xs, masks):
        kk = masks.shape[0] // xs.shape[0]
        self.rows.append((int(xs.shape[0]), int(masks.shape[0])))
        return torch.softmax(torch.cat([xs.repeat_interleave(kk, 0), masks.float()], 1) @ self.w_srg, -1), None

    def fw_classifier(self, model, xs, masks):
        y = torch.softmax(xs @ self.w_cls, -1)
        return y, y


class _Frozen(torch.nn.Module):
    pass


class _ExplainerTrainer:
    """stands in for training.make_explainer_trainer(...): forward + Shapley loss (models/shapley.py:40-52) + backward by
    torch autograd, gradients written to .grad and reported to training.GRAD_SINK in backward order."""

    def __init__(self, model):
        self.model, self.losses = model, []

    def loss_and_grads(self, xs, bits, v_0, v_s, v_1, k, labels=None, train=True, seed=0):
        from autognothi_amd import training as tr
        nb = xs.shape[0]
        phi = self.model(xs).view(nb, C, P)
        approx = v_0.view(1, 1, C) + bits.view(nb, k, P).float() @ phi.permute(0, 2, 1)
        loss = P * torch.nn.functional.mse_loss(approx.reshape(-1, C), v_s, reduction="mean")
        grads = torch.autograd.grad(loss, list(self.model.parameters()))
        for q, g_ in reversed(list(zip(self.model.parameters(), grads))):
            q.grad = g_.clone()
            if tr.GRAD_SINK is not None:
                tr.GRAD_SINK(q)
        self.losses.append(float(loss.detach()))
        return loss.detach(), phi.detach()


class _SurrogateTrainer:
    def __init__(self, model, recipe):
        self.model, self.recipe, self.losses = model, recipe, []

    def loss_and_grads(self, xs, bits, orig, train=True, seed=0):
        from autognothi_amd import training as tr
        cur = torch.softmax(torch.cat([xs, bits.float()], 1) @ self.model.weight.t() + self.model.bias, -1)
        loss = torch.nn.functional.kl_div(torch.log_softmax(orig, -1), torch.softmax(cur, -1), reduction="batchmean")
        grads = torch.autograd.grad(loss, list(self.model.parameters()))
        for q, g_ in reversed(list(zip(self.model.parameters(), grads))):
            q.grad = g_.clone()
            if tr.GRAD_SINK is not None:
                tr.GRAD_SINK(q)
        self.losses.append(float(loss.detach()))
        return loss.detach(), cur.detach()


class _Env:
    def __init__(self):
        self.lines = []

    def log(self, msg):
        self.lines.append(msg)


def _run_explainer_epochs():
    """two epochs of the real explainer_epoch_train on the stubs -> dict of everything comparable."""
    from autognothi_amd.scripts import train_explainer as te
    torch.manual_seed(11)
    model = torch.nn.Sequential(torch.nn.Linear(DFEAT, 16), torch.nn.Tanh(), torch.nn.Linear(16, C * P))
    trainer = _ExplainerTrainer(model)
    model.__dict__["_ag_trainer"] = trainer
    recipe, srg, env = _Recipe(), _Frozen(), _Env()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    v_0 = torch.full((1, C), 1.0 / C)
    src = OracleMaskSource(SEED)
    epoch_loss = []
    for epoch in (1, 2):
        epoch_loss.append(te.explainer_epoch_train(env, torch.device("cpu"), K, P, v_0, _data(), recipe, srg, model, opt, epoch,
                                                   lambda a, b: (a, b), seed=None, target_rows=16, mask_source=src))
    return dict(params=[q.detach().clone() for q in model.parameters()], epoch_loss=epoch_loss, losses=trainer.losses,
                masks=[(kind, n, rows) for kind, n, rows in src.log], log=env.lines, fw_rows=list(recipe.rows))


def _run_surrogate_epoch():
    from autognothi_amd.scripts import train_surrogate as ts
    torch.manual_seed(12)
    model = torch.nn.Linear(DFEAT + P, C)
    recipe, env = _Recipe(), _Env()
    trainer = _SurrogateTrainer(model, recipe)
    model.__dict__["_ag_trainer"] = trainer
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    src = OracleMaskSource(SEED + 1)
    val = ts.surrogate_epoch_train(env, torch.device("cpu"), P, _data(), recipe, _Frozen(), model, opt, 1, lambda a, b: (a, b),
                                   seed=None, mask_source=src)
    return dict(params=[q.detach().clone() for q in model.parameters()], epoch_loss=val, losses=trainer.losses,
                masks=[(kind, n, rows) for kind, n, rows in src.log], log=env.lines)


class _FaithRecipe:
    """fw_final / fw_surrogate stand-ins for measure_faithfulness: deterministic functions of the sample and the mask row."""

    def __init__(self):
        g = torch.Generator().manual_seed(21)
        self.w = torch.randn(2 * P, C, generator=g)
        self.rows = []

    def fw_final(self, m_final, xs):
        return None, torch.stack([xs.view(-1) * (c + 1) + 0.01 * torch.arange(P) for c in range(C)]).view(1, C, P)

    def fw_surrogate(self, model, xs, masks):
        self.rows.append(int(masks.shape[0]))
        return torch.softmax(torch.cat([xs.expand(masks.shape[0], -1), masks.float()], 1) @ self.w, -1), None


def _perturbed_masks_oracle(attr, steps, mask_base):
    """ops.perturbed_masks on the numpy oracle (oracle/shapley.py: reference scripts/measure_faithfulness.py:225-251)."""
    from oracle import shapley as osh
    a = attr.numpy()
    outs = [osh.get_perturbed_samples(a[c], a.shape[1], steps, mask_base) for c in range(a.shape[0])]
    return torch.from_numpy(outs[0][0]), torch.from_numpy(np.stack([m for _, m in outs]))


def _run_faithfulness():
    """the REAL measure_faithfulness_loaded + infer_perturbed (row split of the tail samples, gather) on stub kernels."""
    from autognothi_amd.scripts import measure_faithfulness as mf
    keep = mf.ops.perturbed_masks
    mf.ops.perturbed_masks = _perturbed_masks_oracle
    try:
        g = torch.Generator().manual_seed(9)
        samples = [(torch.randn(1, P, generator=g), torch.tensor([i % C])) for i in range(7)]
        env, recipe = _Env(), _FaithRecipe()
        rep = mf.measure_faithfulness_loaded(env, torch.device("cpu"), recipe, _Frozen(), _Frozen(), samples, lambda a, b: (a, b), 4)
    finally:
        mf.ops.perturbed_masks = keep
    return dict(report=rep, fw_rows=recipe.rows, log=env.lines)


def _baseline(path):
    """the world-1 run: no process group (distributed.world() == (0, 1))."""
    assert not dist.is_initialized()
    torch.save(dict(exp=_run_explainer_epochs(), srg=_run_surrogate_epoch(), faith=_run_faithfulness()), path)


def _merge_masks(mine, world, rank, by_mask):
    """[(kind, n_total, local rows)] of every rank -> the global rows of every call (rank order = input order).  A call in mask
    mode was drawn WHOLE by every rank (identically): one copy of it."""
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    out = []
    for calls in zip(*parts):
        assert len({(c[0], c[1]) for c in calls}) == 1
        if by_mask(calls[0][0], calls[0][1]):
            for c in calls[1:]:
                assert np.array_equal(c[2], calls[0][2])
            out.append(calls[0])
        else:
            out.append((calls[0][0], calls[0][1], np.concatenate([c[2] for c in calls], axis=0)))
    return out


def _worker(rank, world, port, path, out):
    here = os.path.dirname(os.path.abspath(__file__))
    for p_ in (os.path.dirname(here), here):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autognothi_amd import distributed as D
        want = torch.load(path, weights_only=False)
        for key, run in (("exp", _run_explainer_epochs), ("srg", _run_surrogate_epoch)):
            got, ref = run(), want[key]
            # masks: every call is the same global call on every rank, and the union of the ranks' rows is the world-1 stream
            by_mask = lambda kind, n: kind == "shapley" and n < world    # noqa: E731  (mask mode: every rank draws the whole call)
            merged = _merge_masks(got["masks"], world, rank, by_mask)
            assert len(merged) == len(ref["masks"])
            for (kind, n, rows), (kind1, n1, rows1) in zip(merged, ref["masks"]):
                assert (kind, n) == (kind1, n1) and np.array_equal(rows, rows1)
            if key == "exp":
                # NO RANK IDLES on the one-input batches: each rank ran K / world of that input's masks through the surrogate
                # (plus the all-ones forward); two epochs
                assert got["fw_rows"].count((1, K // world)) == 2 * BATCHES.count(1), got["fw_rows"]
            # per-batch losses: sum_r (B_r / B) * loss_r == the world-1 batch loss; a batch with fewer inputs than ranks runs in
            # mask mode where that is possible (explainer: K >= ranks): EVERY rank then computes the whole batch's loss
            reps = 2 if key == "exp" else 1
            spans = [D.shard_range(b) for b in BATCHES] * reps
            sizes = BATCHES * reps
            it = iter(got["losses"])
            mine = []
            for (lo, hi), n_ in zip(spans, sizes):
                if key == "exp" and n_ < world:
                    mine.append(next(it) / world)
                else:
                    mine.append(((hi - lo) / n_) * next(it) if hi > lo else 0.0)
            assert next(it, None) is None
            tot = torch.tensor(mine, dtype=torch.float64)
            dist.all_reduce(tot)
            np.testing.assert_allclose(tot.numpy(), np.asarray(ref["losses"]), rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(got["epoch_loss"], ref["epoch_loss"], rtol=1e-5)
            # parameters after every optimiser step of the run: identical on both ranks and equal to the world-1 run
            for q, q1 in zip(got["params"], ref["params"]):
                torch.testing.assert_close(q, q1, rtol=2e-5, atol=2e-6)
                both = [torch.empty_like(q) for _ in range(world)]
                dist.all_gather(both, q)
                assert torch.equal(both[0], both[1])
            assert (len(got["log"]) > 0) == (rank == 0)            # rank 0 is the only writer
        got, ref = _run_faithfulness(), want["faith"]
        # samples 0..5 sharded by image (three whole forwards of 2*C*S rows per rank); sample 6 — fewer samples than ranks — is
        # sharded INSIDE the image: both ranks run half of its rows and gather them
        full = ref["fw_rows"][0]
        assert ref["fw_rows"] == [full] * 7 and full == 2 * C * 4
        assert got["fw_rows"] == [full] * 3 + [full // world], got["fw_rows"]
        assert got["report"]["data_cls"] == ref["report"]["data_cls"]
        assert got["report"]["data_ins"] == ref["report"]["data_ins"] and got["report"]["data_del"] == ref["report"]["data_del"]
        for k_ in ("insertion", "deletion", "insertion_non_ok", "deletion_non_ok"):
            assert got["report"][k_] == ref["report"][k_]
        assert (len(got["log"]) > 0) == (rank == 0)
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_entrypoint_epochs_world_2_equal_world_1():
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "world1.pt")
        _baseline(path)
        port = _free_port()
        ctx = mp.get_context("spawn")
        out = ctx.Array("i", [0] * world)
        procs = [ctx.Process(target=_worker, args=(r, world, port, path, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        assert list(out) == [1, 1]


def test_checkpoint_write_is_atomic_and_main_only(tmp_path):
    """save_epoch_ckpt writes aside + renames (no partial file is ever visible) and prunes with unlink(missing_ok)."""
    from autognothi_amd.scripts import resources as R
    sd = {"w": torch.arange(4.0)}
    assert R.save_epoch_ckpt(tmp_path, "explainer", "_:%10==0", 3, 1, sd)
    assert R.save_epoch_ckpt(tmp_path, "explainer", "_:%10==0", 3, 2, sd)       # epoch 1 is neither scheduled nor final: pruned
    names = sorted(p.name for p in tmp_path.iterdir())
    assert names == ["explainer-epoch-2.ckpt"], names
    assert R.save_epoch_ckpt(tmp_path, "explainer", "_:%10==0", 3, 2, sd)       # overwrite in place
    assert torch.equal(torch.load(tmp_path / "explainer-epoch-2.ckpt")["w"], sd["w"])

    class Env:
        flushed = 0

        def flush_cfg(self):
            Env.flushed += 1

    class Cfg:
        ckpt_when, epochs = "_:%10==0", 3
    assert R.save_epoch_ckpt_main(tmp_path, "explainer", Cfg, 3, sd, Env())
    assert Env.flushed == 1 and (tmp_path / "explainer-epoch-3.ckpt").exists()
