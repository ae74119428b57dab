"""RCCL on the box's one GPU (world size 1): the collectives of the N > 1 path issued for real — bucketed asynchronous gradient
all-reduce overlapped with the manual backward (distributed.GradBucketReducer through training.GRAD_SINK), scalar reductions,
and bench.py launched by torch.distributed.run.  Values cannot differ across ranks here; what is checked is that the exchange
runs on RCCL streams next to the HIP kernels and leaves exactly the gradients of the plain step (scaled by the averaging)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from util import build_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def rccl_group(cuda_device):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda_device)
    yield dist
    dist.destroy_process_group()


def test_reducer_overlapped_with_backward_on_rccl(cuda_device, rccl_group, monkeypatch):
    """one explainer training step (ViT-tiny fixture) with every gradient passing through the bucket reducer — pretending
    two ranks, so the all-reduce (a sum over the ONE real rank) is followed by the division by two — against the plain step."""
    from autognothi_amd import distributed as D, engine, ops, training as T
    c = build_case("vit_tiny_c1")
    dev, recipe = cuda_device, c["recipe"]
    engine.set_precision("fp32")
    srg = c["surrogate"].to(dev).eval()
    exp = c["explainer"].to(dev)
    exp.train()
    xs = torch.from_numpy(c["xs"]).to(dev)
    b, k, p = c["B"], c["K"], c["P"]
    _, bits = ops.mask_shapley_new(ops.DeviceMT19937(dev, 3407), b * k, p, want_i64=False, want_bits=True)
    with torch.no_grad():
        v_s, _ = recipe.fw_surrogate(srg, xs, bits)
        v_1, _ = recipe.fw_surrogate(srg, xs, torch.ones((b, p), dtype=torch.int64, device=dev))
    v_0 = torch.full((1, v_s.shape[1]), 1.0 / v_s.shape[1], device=dev)
    trainer = T.make_explainer_trainer(recipe, exp)
    params = [q for q in exp.parameters() if q.requires_grad]

    def step(sink):
        for q in params:
            q.grad = None
        T.GRAD_SINK = sink
        try:
            loss, _ = trainer.loss_and_grads(xs, bits, v_0, v_s, v_1, k, train=False, seed=1)
        finally:
            T.GRAD_SINK = None
        return loss

    step(None)
    want = [q.grad.clone() for q in params]
    monkeypatch.setattr(D, "world", lambda: (0, 2))
    red = D.GradBucketReducer(params, bucket_bytes=1 << 20)      # ViT-tiny: 22 MB of gradients = many buckets in flight
    step(red.ready)
    n_coll = red.finish()
    torch.cuda.synchronize()
    assert n_coll >= 4, n_coll
    gscale = max(float(w.abs().max()) for w in want)
    for q, w in zip(params, want):     # (two runs of the backward agree to fp32 rounding, not bit for bit: dX of LayerNorm / attention
        torch.testing.assert_close(q.grad, w / 2, rtol=1e-4, atol=1e-6 * gscale)   # accumulate in launch-dependent order)
    assert D.reduce_scalars([3.0, 4.0], dev) == [3.0, 4.0]


def test_sharded_epoch_body_on_rccl(cuda_device, rccl_group, monkeypatch):
    """scripts/train_explainer.explainer_epoch_train on its N > 1 path with the collectives on RCCL: the process pretends to be
    rank 0 of 2, so of every global batch of 4 inputs it takes inputs [0, 2) and rows [0, 2K) of the global mask call, weights
    its gradients by 1/2 and sums them over the (one real) rank.  Expected, from the plain pieces: masks = the first 2K rows of
    mask_shapley_new(4K, P) on the epoch's seed, parameter step = lr * (1/2) * gradient of the local batch-mean loss, epoch
    loss = (1/2) * loss / 2 inputs."""
    from autognothi_amd import distributed as D, engine, ops, training as T
    from autognothi_amd.scripts import train_explainer as te
    c = build_case("vit_tiny_c1")
    dev, recipe = cuda_device, c["recipe"]
    engine.set_precision("fp32")
    srg = c["surrogate"].to(dev).eval()
    exp = c["explainer"].to(dev)
    k, p = c["K"], c["P"]
    xs4 = torch.cat([torch.from_numpy(c["xs"]), torch.from_numpy(c["xs"]).flip(0) * 0.5], 0).to(dev)     # a global batch of 4 inputs
    zs4 = torch.zeros(4, dtype=torch.long, device=dev)
    v_0 = torch.full((1, c["g"]["v_0"].shape[1]), 1.0 / c["g"]["v_0"].shape[1], device=dev)
    seed, epoch = 99, 1
    # expectation from the plain pieces
    bits_full = ops.mask_shapley_new(ops.DeviceMT19937(dev, seed), 4 * k, p, want_i64=False, want_bits=True)[1]
    bits = bits_full[:2 * k].contiguous()
    with torch.no_grad():
        v_s, _ = recipe.fw_surrogate(srg, xs4[:2], bits)
        v_1, _ = recipe.fw_surrogate(srg, xs4[:2], torch.ones((2, p), dtype=torch.int64, device=dev))
    trainer = T.make_explainer_trainer(recipe, exp)
    exp.__dict__["_ag_trainer"] = trainer
    params = [q for q in exp.parameters() if q.requires_grad]
    exp.train()
    for q in params:
        q.grad = None
    step0 = trainer.step                   # (the dropout streams are keyed on (seed, trainer.step): replay the same step below)
    loss, _ = trainer.loss_and_grads(xs4[:2], bits, v_0, v_s, v_1, k, labels=zs4[:2], train=True, seed=seed + epoch)
    trainer.step = step0
    want_grad = [q.grad.clone() for q in params]
    before = [q.detach().clone() for q in params]
    # the epoch body as rank 0 of 2
    monkeypatch.setattr(D, "world", lambda: (0, 2))
    opt = torch.optim.SGD(params, lr=0.5)
    got = te.explainer_epoch_train(None, dev, k, p, v_0, [(None, None)], recipe, srg, exp, opt, epoch, lambda a, b_: (xs4, zs4),
                                   seed=seed)
    torch.cuda.synchronize()
    assert abs(got - 0.5 * float(loss) / 2) <= 1e-5 * abs(float(loss))
    gscale = max(float(w.abs().max()) for w in want_grad)
    for q, q0, w in zip(params, before, want_grad):
        torch.testing.assert_close(q0 - q.detach(), 0.5 * 0.5 * w, rtol=1e-4, atol=1e-6 * gscale)


def test_three_steps_of_gradient_views_through_fused_adamw_on_rccl(cuda_device, rccl_group, monkeypatch):
    """the reducer makes every ``.grad`` a VIEW of a persistent flat bucket buffer that the next step overwrites, and the
    layouts are cached step after step — here three consecutive steps of the epoch body (rank 0 of a pretended 2, collectives on RCCL) with
    torch's fused AdamW reading those views, against the same three steps taken by hand: plain backward, gradients halved (rank 0's share of
    the batch, summed over the one real rank), the same optimiser.  Masks: the epoch's one generator continues from step to step; the
    dropout streams are keyed on (seed, trainer.step)."""
    import copy
    from autognothi_amd import distributed as D, engine, ops, training as T
    from autognothi_amd.scripts import train_explainer as te
    c = build_case("vit_tiny_c1")
    dev, recipe = cuda_device, c["recipe"]
    engine.set_precision("fp32")
    srg = c["surrogate"].to(dev).eval()
    exp = c["explainer"].to(dev)
    k, p = c["K"], c["P"]
    xs4 = torch.cat([torch.from_numpy(c["xs"]), torch.from_numpy(c["xs"]).flip(0) * 0.5], 0).to(dev)
    zs4 = torch.zeros(4, dtype=torch.long, device=dev)
    v_0 = torch.full((1, c["g"]["v_0"].shape[1]), 1.0 / c["g"]["v_0"].shape[1], device=dev)
    seed, epoch, n_steps = 77, 1, 3
    trainer = T.make_explainer_trainer(recipe, exp)
    exp.__dict__["_ag_trainer"] = trainer
    params = [q for q in exp.parameters() if q.requires_grad]
    state0 = copy.deepcopy(exp.state_dict())
    step0 = trainer.step
    exp.train()
    # ---- by hand
    opt = torch.optim.AdamW(params, lr=1e-3, eps=1e-4, fused=True)      # (eps well above the fp32 noise of two runs of the backward: no sign flips of ~0 gradients)
    engine.watch_optimizer(opt)
    rng = ops.DeviceMT19937(dev, seed)
    for _ in range(n_steps):
        bits = ops.mask_shapley_new(rng, 4 * k, p, want_i64=False, want_bits=True)[1][:2 * k].contiguous()
        with torch.no_grad():
            v_s, _ = recipe.fw_surrogate(srg, xs4[:2], bits)
            v_1, _ = recipe.fw_surrogate(srg, xs4[:2], torch.ones((2, p), dtype=torch.int64, device=dev))
        for q in params:
            q.grad = None
        trainer.loss_and_grads(xs4[:2], bits, v_0, v_s, v_1, k, labels=zs4[:2], train=True, seed=seed + epoch)
        for q in params:
            q.grad.mul_(0.5)
        gmax = [float(q.grad.abs().max()) for q in params]
        opt.step()
    want = [q.detach().clone() for q in params]
    # ---- the epoch body as rank 0 of 2, from the same start
    exp.load_state_dict(state0)
    trainer.step = step0
    monkeypatch.setattr(D, "world", lambda: (0, 2))
    opt2 = torch.optim.AdamW(params, lr=1e-3, eps=1e-4, fused=True)
    te.explainer_epoch_train(None, dev, k, p, v_0, [(None, None)] * n_steps, recipe, srg, exp, opt2, epoch, lambda a, b_: (xs4, zs4), seed=seed)
    torch.cuda.synchronize()
    moved = max(float((q.detach() - w0.to(dev)).abs().max()) for q, w0 in zip(params, [state0[n_] for n_, q_ in exp.named_parameters() if q_.requires_grad]))
    assert moved > 1e-4                                  # (three AdamW steps did move the parameters)
    # (a parameter whose gradient is analytically zero — the key biases: a soft-max ignores a constant added to every logit of a row; the last
    # bias of the explainer head: the Shapley normalisation removes any constant — receives rounding noise, which AdamW turns into steps of either sign: compared are the parameters with a gradient above the noise)
    live = [g >= 1e-4 * max(gmax) for g in gmax]
    assert sum(live) >= 0.9 * len(params)
    for q, w, on in zip(params, want, live):
        if on:
            torch.testing.assert_close(q.detach(), w, rtol=2e-4, atol=2e-5)    # (atol = 0.7 % of one AdamW step of lr 1e-3: two runs of the backward agree to fp32 rounding only)
    # every gradient is (still) a view of one of the reducer's flat buffers
    red = exp.__dict__.get("_ag_reducer")
    if red is not None:
        bases = {flat.untyped_storage().data_ptr() for flat, _, _ in red._bufs.values()}
        assert all(q.grad is None or q.grad.untyped_storage().data_ptr() in bases for q in params)


def test_mask_mode_epoch_body_on_rccl(cuda_device, rccl_group, monkeypatch):
    """A global batch with FEWER inputs than ranks (one input, pretending rank 0 of 2): explainer_epoch_train shards the K masks
    inside the input (common.shard_auto) — this rank draws the WHOLE global mask call, runs masks [0, K/2) through the surrogate,
    the targets are exchanged through RCCL (gather_masks_within_inputs: here the absent peer's half stays zero) and the explainer
    takes a whole-batch step with no gradient exchange.  Expected, from the plain pieces: that step on (all K mask rows,
    v_s = [this rank's half | zeros])."""
    from autognothi_amd import distributed as D, engine, ops, training as T
    from autognothi_amd.scripts import train_explainer as te
    c = build_case("vit_tiny_c1")
    dev, recipe = cuda_device, c["recipe"]
    engine.set_precision("fp32")
    srg = c["surrogate"].to(dev).eval()
    exp = c["explainer"].to(dev)
    k, p = c["K"], c["P"]
    xs1 = torch.from_numpy(c["xs"][:1]).to(dev)
    zs1 = torch.zeros(1, dtype=torch.long, device=dev)
    v_0 = torch.full((1, c["g"]["v_0"].shape[1]), 1.0 / c["g"]["v_0"].shape[1], device=dev)
    seed, epoch = 41, 1
    bits = ops.mask_shapley_new(ops.DeviceMT19937(dev, seed), k, p, want_i64=False, want_bits=True)[1]
    with torch.no_grad():
        v_half, _ = recipe.fw_surrogate(srg, xs1, bits[:k // 2].contiguous())
        v_1, _ = recipe.fw_surrogate(srg, xs1, torch.ones((1, p), dtype=torch.int64, device=dev))
    v_s = torch.cat([v_half, torch.zeros_like(v_half)], 0)
    trainer = T.make_explainer_trainer(recipe, exp)
    exp.__dict__["_ag_trainer"] = trainer
    params = [q for q in exp.parameters() if q.requires_grad]
    exp.train()
    for q in params:
        q.grad = None
    step0 = trainer.step
    loss, _ = trainer.loss_and_grads(xs1, bits, v_0, v_s, v_1, k, labels=zs1, train=True, seed=seed + epoch)
    trainer.step = step0
    want_grad = [q.grad.clone() for q in params]
    before = [q.detach().clone() for q in params]
    monkeypatch.setattr(D, "world", lambda: (0, 2))
    opt = torch.optim.SGD(params, lr=0.5)
    got = te.explainer_epoch_train(None, dev, k, p, v_0, [(None, None)], recipe, srg, exp, opt, epoch, lambda a, b_: (xs1, zs1), seed=seed)
    torch.cuda.synchronize()
    gscale = max(float(w.abs().max()) for w in want_grad)
    for q, q0, w in zip(params, before, want_grad):     # the whole-batch step, un-weighted, nothing exchanged
        torch.testing.assert_close(q0 - q.detach(), 0.5 * w, rtol=1e-4, atol=1e-6 * gscale)
    # epoch figure: this rank reports 1/2 of the batch loss and 1/2 of the sample; the peer's half is absent here: 0.5 L / max(round(0.5), 1)
    assert got == pytest.approx(0.5 * float(loss), rel=1e-5)


def test_bench_under_launcher_runs_rccl_barriers():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py`: the driver's N > 1 launch line with one rank."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k_, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--no-cpu-baseline", "--train-batch", "2", "--attr-batch", "4"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["secondary"]["train_explainer_step"]["value"] > 0


def test_bench_with_a_real_world_size_of_two_over_gloo():
    """bench.py's N > 1 code path with TWO real ranks on the box's one GPU: a second RCCL rank cannot share a device, so the collectives go over
    gloo (AG_BENCH_BACKEND=gloo, AG_BENCH_DEVICE=0: development switches, never a measurement).  What is checked: the launch line of the driver
    (`python bench.py --gpus 2` spawns torch.distributed.run) runs to the end — lean mode, the sharded training epochs with their two-rank
    gradient exchange and the one-decision-of-all-ranks schedule, the attribution leg, the max-over-ranks timing — and rank 0 prints ONE line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", AG_BENCH_BACKEND="gloo", AG_BENCH_DEVICE="0")
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k_, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--train-batch", "2",
           "--attr-batch", "4"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["rows_per_step"] == 2 * line["config"]["rows_per_gpu_per_step"] and line["config"]["collective_backend"] == "gloo (dev)"
    assert line["secondary"]["train_explainer_step"]["value"] > 0
    assert line["secondary"]["config5_train_explainer_step"]["duo_bert_base"]["value"] > 0
    assert line["secondary"]["value"] > 0        # Shapley attributions / s over both ranks
