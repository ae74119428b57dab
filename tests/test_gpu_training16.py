"""GPU: the bf16-activation training step (autognothi_amd/training16.py on ag_gemm_ex + the fused row kernels) against torch
autograd on the CPU port of the reference (oracle/torch_port.py), dropout off: loss, phi and the gradient of every trainable
parameter of the vanilla / froyo / duo explainers (ViT and BERT) and of the surrogate — within bf16 operand-rounding distance
(the reference under torch.autocast(bf16) is the yardstick: tests/test_gpu_fulldepth.py) — plus what the structure promises:
bit-identical gradients from run to run and with / without the side stream, dropout consistency between forward and backward,
no stale bf16 weights after fused optimiser steps."""
import numpy as np
import pytest
import torch

from oracle import torch_port as otp
from util import build_case

pytestmark = pytest.mark.gpu


def _zero_dropout(meta):
    prm = dict(meta["params"])
    prm["attention_probs_dropout_prob"] = 0.0
    prm["hidden_dropout_prob"] = 0.0
    return prm


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.fixture
def mixed():
    from autognothi_amd import training
    training.MIXED_BF16 = True
    yield training
    training.MIXED_BF16 = False


def _explainer_case(tag, dev, layers=None):
    from autognothi_amd import ops
    from autognothi_amd.utils import synth
    c = build_case(tag)
    g, recipe = c["g"], c["recipe"]
    prm = _zero_dropout(c["meta"])
    if layers is not None:
        prm["num_hidden_layers"] = layers
    cfg = recipe.t_config(**prm)
    exp = recipe.t_explainer(cfg)
    synth.load_synth_weights(exp, seed=1)
    exp = exp.to(dev)
    exp.train()
    xs = torch.from_numpy(c["xs"]).to(dev)
    masks = torch.from_numpy(c["masks"])
    bits = ops.pack_mask(masks.to(dev))
    v0, vs, v1 = [torch.from_numpy(g[k]) for k in ("v_0", "v_s", "v_1")]
    labels = torch.tensor([1, 0][:c["B"]], dtype=torch.long)
    return c, prm, exp, xs, masks, bits, v0, vs, v1, labels


def _reference_grads(c, prm, exp, masks, v0, vs, v1, labels):
    kind, duo = c["meta"]["kind"], c["meta"]["duo"]
    sd = {k: v.detach().cpu().clone().requires_grad_(exp.state_dict(keep_vars=True)[k].requires_grad)
          for k, v in exp.state_dict(keep_vars=True).items()}
    ones = torch.ones((c["B"], c["P"]), dtype=torch.long)
    phi_ref, z_ref = otp.explainer_phi(torch.from_numpy(c["xs"]), ones, v1, v0, sd, prm, kind)
    loss_ref = otp.shapley_loss(masks.reshape(c["B"], c["K"], c["P"]), v0, vs, phi_ref, c["P"])
    if duo:
        lin = torch.nn.functional.linear
        if kind == "vit":
            base = torch.softmax(lin(z_ref[:, 0], sd["classifier.weight"], sd["classifier.bias"]), -1)
        else:
            base = lin(torch.tanh(lin(z_ref[:, 0], sd["bert_pooler.dense.weight"], sd["bert_pooler.dense.bias"])),
                       sd["classifier.weight"], sd["classifier.bias"])
        loss_ref = loss_ref + torch.nn.functional.cross_entropy(base, labels)
    loss_ref.backward()
    return sd, phi_ref, loss_ref


@pytest.mark.parametrize("tag", ["froyo_vit_tiny_l3", "vit_tiny_c1", "bert_base_l2", "duo_vit_tiny_l3", "duo_bert_base_l2"])
def test_bf16_explainer_step_matches_autograd(cuda_device, mixed, tag):
    from autognothi_amd import training16
    dev = cuda_device
    c, prm, exp, xs, masks, bits, v0, vs, v1, labels = _explainer_case(tag, dev, layers=2 if tag == "vit_tiny_c1" else None)
    tr = mixed.ExplainerTrainer(c["recipe"], exp)
    assert isinstance(tr, training16.ExplainerTrainer16), "the bf16 step was not selected"
    loss, phi = tr.loss_and_grads(xs, bits, v0.to(dev), vs.to(dev), v1.to(dev), c["K"], labels=labels.to(dev), train=True)
    sd, phi_ref, loss_ref = _reference_grads(c, prm, exp, masks, v0, vs, v1, labels)
    np.testing.assert_allclose(loss.cpu().numpy()[0], loss_ref.item(), rtol=3e-2)
    pr = phi_ref.detach().numpy()
    assert np.abs(phi.cpu().numpy() - pr).max() <= 4e-2 * np.abs(pr).max() + 1e-4
    checked = 0
    gscale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    worst = (0.0, None)
    for name, p in exp.named_parameters():
        ref = sd[name].grad
        if not p.requires_grad:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert ref is not None and p.grad is not None, name
        got = p.grad.cpu().numpy()
        assert np.isfinite(got).all(), name
        if float(ref.abs().max()) < 1e-3 * gscale:   # structurally zero / tiny gradients: bounded relative to the largest one
            assert float(np.abs(got).max()) < 1e-2 * gscale, name
            continue
        r = _rel(got, ref.numpy())
        worst = max(worst, (r, name))
        assert r < 8e-2, (name, r)
        checked += 1
    assert checked >= 10
    print(f"{tag}: worst relative gradient deviation {worst[0]:.3e} ({worst[1]}), {checked} tensors")


def test_bf16_step_is_bit_reproducible_and_stream_independent(cuda_device, mixed):
    """same inputs -> the same bits in every gradient, run to run, and with the dW products on the side stream or inline
    (slabs are added in slab order, partials in block order: no atomics anywhere)."""
    from autognothi_amd import training16
    dev = cuda_device
    c, prm, exp, xs, masks, bits, v0, vs, v1, labels = _explainer_case("bert_base_l2", dev)

    def grads(side_on):
        keep = training16.SIDE_STREAM
        training16.SIDE_STREAM = side_on
        training16._Side._per_device.clear()
        try:
            for p in exp.parameters():
                p.grad = None
            tr = mixed.ExplainerTrainer(c["recipe"], exp)
            loss, _ = tr.loss_and_grads(xs, bits, v0.to(dev), vs.to(dev), v1.to(dev), c["K"], labels=labels.to(dev), train=True, seed=3)
            torch.cuda.synchronize()
            return float(loss), {n: p.grad.clone() for n, p in exp.named_parameters() if p.grad is not None}
        finally:
            training16.SIDE_STREAM = keep
            training16._Side._per_device.clear()

    l0, g0 = grads(True)
    l1, g1 = grads(True)
    l2, g2 = grads(False)
    assert l0 == l1 == l2
    assert set(g0) == set(g1) == set(g2) and len(g0) >= 10
    for n in g0:
        if n.endswith("word_embeddings.weight"):
            continue      # torch's index_add_ (float atomics on duplicate token ids): index plumbing outside this library
        assert torch.equal(g0[n], g1[n]), f"{n}: differs between two runs"
        assert torch.equal(g0[n], g2[n]), f"{n}: differs with the side stream off"


def test_an_aborted_backward_leaves_nothing_for_the_next_step(cuda_device, mixed):
    """training16._Side.begin: a backward that raised half way leaves deferred dW products (and kept tensors, unreported gradients)
    behind; the next step must not send them off with its first group — its gradients equal a clean step's bit for bit."""
    from autognothi_amd import training16
    dev = cuda_device
    c, prm, exp, xs, masks, bits, v0, vs, v1, labels = _explainer_case("bert_base_l2", dev)
    tr = mixed.ExplainerTrainer(c["recipe"], exp)

    def grads():
        for p in exp.parameters():
            p.grad = None
        loss, _ = tr.loss_and_grads(xs, bits, v0.to(dev), vs.to(dev), v1.to(dev), c["K"], labels=labels.to(dev), train=True, seed=3)
        torch.cuda.synchronize()
        return float(loss), {n: p.grad.clone() for n, p in exp.named_parameters() if p.grad is not None}

    l0, g0 = grads()
    side = training16._Side.of(dev)
    assert side.pending == [] and side.keep == [] and side.finals == []
    if training16.DW_GROUP > 1:
        # what an exception inside the backward leaves: some Linear's (dY, X) pair waiting for its group, a gradient not yet reported
        lin = next(m for m in exp.modules() if isinstance(m, torch.nn.Linear) and m.weight.requires_grad)
        lw = training16.LinW([lin], 1, dev)
        n, k = lin.weight.shape
        side.pending.append((lw, torch.ones((256, n), dtype=torch.bfloat16, device=dev), torch.ones((256, k), dtype=torch.bfloat16, device=dev), None))
    side.finals.append(next(p for p in exp.parameters() if p.requires_grad))
    side.keep.append(torch.ones(8, device=dev))
    l1, g1 = grads()
    assert l0 == l1 and set(g0) == set(g1)
    for n_ in g0:
        if n_.endswith("word_embeddings.weight"):
            continue      # (torch index_add_ atomics, see above)
        assert torch.equal(g0[n_], g1[n_]), f"{n_}: the aborted step's leftovers reached this step's gradient"
    assert side.pending == [] and side.keep == [] and side.finals == []


def _dropout_case(tag, dev):
    from autognothi_amd import ops
    from autognothi_amd.utils import synth
    c = build_case(tag)
    recipe = c["recipe"]
    prm = dict(c["meta"]["params"])
    prm["hidden_dropout_prob"], prm["attention_probs_dropout_prob"] = 0.1, 0.1
    if tag == "vit_tiny_c1":
        prm["num_hidden_layers"] = 2
    exp = recipe.t_explainer(recipe.t_config(**prm))
    synth.load_synth_weights(exp, seed=1)
    exp = exp.to(dev).train()
    xs = torch.from_numpy(c["xs"]).to(dev)
    bits = ops.pack_mask(torch.from_numpy(c["masks"]).to(dev))
    v0, vs, v1 = [torch.from_numpy(c["g"][k]).to(dev) for k in ("v_0", "v_s", "v_1")]
    labels = torch.tensor([1, 0][:c["B"]], dtype=torch.long, device=dev)
    return c, recipe, exp, xs, bits, v0, vs, v1, labels


@pytest.mark.parametrize("tag", ["vit_tiny_c1", "bert_base_l2", "duo_bert_base_l2"])
def test_bf16_step_with_dropout_matches_the_fp32_step(cuda_device, tag):
    """dropout ON: the bf16 step and the exact-fp32 step of training.py draw the same keep decisions (same counter hash, same
    element indices, same seed sequence), so their losses and gradients agree to bf16 rounding — which pins every dropout site
    of the forward AND its mirror in the backward (embeddings, attention probabilities, both residual branches, the BERT
    explainer dropout, the duo pooler)."""
    from autognothi_amd import training, training16
    dev = cuda_device
    c, recipe, exp, xs, bits, v0, vs, v1, labels = _dropout_case(tag, dev)

    def run(mixed_on):
        training.MIXED_BF16 = mixed_on
        try:
            for p in exp.parameters():
                p.grad = None
            tr = training.ExplainerTrainer(recipe, exp)
            assert isinstance(tr, training16.ExplainerTrainer16) == mixed_on
            loss, phi = tr.loss_and_grads(xs, bits, v0, vs, v1, c["K"], labels=labels, train=True, seed=11)
            return float(loss), phi.clone(), {n: p.grad.clone() for n, p in exp.named_parameters() if p.grad is not None}
        finally:
            training.MIXED_BF16 = False

    l32, phi32, g32 = run(False)
    l16, phi16, g16 = run(True)
    assert abs(l16 - l32) <= 3e-2 * abs(l32)
    assert float((phi16 - phi32).abs().max()) <= 4e-2 * float(phi32.abs().max()) + 1e-4
    assert set(g16) == set(g32)
    gscale = max(float(v.abs().max()) for v in g32.values())
    checked = 0
    for n in g32:
        ref = g32[n]
        if float(ref.abs().max()) < 1e-3 * gscale:
            assert float(g16[n].abs().max()) < 1e-2 * gscale, n
            continue
        r = float((g16[n] - ref).abs().max() / ref.abs().max())
        assert r < 8e-2, (n, r)
        checked += 1
    assert checked >= 10


@pytest.mark.parametrize("tag", ["vit_tiny_c1", "bert_base_l2"])
def test_bf16_step_with_dropout_trains(cuda_device, mixed, tag):
    """a few fused-AdamW steps with dropout on lower the dropout-free loss; every trainable parameter gets a finite gradient."""
    from autognothi_amd import engine, ops
    dev = cuda_device
    c, recipe, exp, xs, bits, v0, vs, v1, labels = _dropout_case(tag, dev)
    tr = mixed.ExplainerTrainer(recipe, exp)

    def eval_loss():
        phi, _ = tr.forward_phi(xs, v0, v1, False, 0)
        tr.saved = None
        loss, _ = ops.shapley_loss(bits, v0, vs, phi, c["B"], c["K"], want_grad=False)
        return float(loss)

    opt = torch.optim.AdamW([p for p in exp.parameters() if p.requires_grad], lr=1e-4, fused=True)
    engine.watch_optimizer(opt)
    first = eval_loss()
    for step in range(8):
        opt.zero_grad()
        loss, _ = tr.loss_and_grads(xs, bits, v0, vs, v1, c["K"], train=True, seed=5)
        for n, p in exp.named_parameters():
            if p.requires_grad:
                assert p.grad is not None and torch.isfinite(p.grad).all(), n
        opt.step()
    last = eval_loss()
    assert np.isfinite(last) and last < first, (first, last)


def test_bf16_surrogate_step_matches_autograd(cuda_device, mixed):
    from autognothi_amd import ops, training16
    from autognothi_amd.utils import synth
    c = build_case("vit_tiny_c1")
    dev, recipe = cuda_device, c["recipe"]
    prm = _zero_dropout(c["meta"])
    prm["num_hidden_layers"] = 2
    srg = recipe.t_surrogate(recipe.t_config(**prm))
    synth.load_synth_weights(srg, seed=0)
    srg = srg.to(dev).train()
    xs = torch.from_numpy(c["xs"]).to(dev)
    masks = torch.from_numpy(c["masks"][:c["B"]])
    bits = ops.pack_mask(masks.to(dev))
    orig = torch.softmax(torch.from_numpy(np.random.default_rng(2).standard_normal((c["B"], 10)).astype(np.float32)), -1)
    tr = mixed.SurrogateTrainer(recipe, srg)
    assert isinstance(tr, training16.SurrogateTrainer16)
    loss, probs = tr.loss_and_grads(xs, bits, orig.to(dev), train=True)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in srg.state_dict().items()}
    mask_t = torch.cat([torch.ones((c["B"], 1), dtype=torch.long), masks], 1)
    z = otp.vit_backbone(torch.from_numpy(c["xs"]), mask_t, sd, prm)
    p_ref = torch.softmax(torch.nn.functional.linear(z[:, 0], sd["classifier.weight"], sd["classifier.bias"]), -1)
    l_ref = torch.nn.functional.kl_div(torch.log_softmax(orig, -1), torch.softmax(p_ref, -1), reduction="batchmean")
    l_ref.backward()
    np.testing.assert_allclose(loss.cpu().numpy()[0], l_ref.item(), rtol=3e-2, atol=1e-6)
    gscale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    checked = 0
    for name, p in srg.named_parameters():
        r = sd[name].grad
        if float(r.abs().max()) < 1e-3 * gscale:
            assert float(p.grad.abs().max()) < 1e-2 * gscale, name
            continue
        assert _rel(p.grad.cpu().numpy(), r.numpy()) < 8e-2, (name, _rel(p.grad.cpu().numpy(), r.numpy()))
        checked += 1
    assert checked >= 10


def test_bf16_weight_bank_follows_fused_optimizer(cuda_device, mixed):
    """torch.optim.AdamW(fused=True) updates parameters without advancing Tensor._version: after three steps the trainer's bf16
    weights must be those of the CURRENT parameters (forward == a fresh trainer on a copy of the state dict)."""
    from autognothi_amd import engine, ops
    dev = cuda_device
    c, prm, exp, xs, masks, bits, v0, vs, v1, labels = _explainer_case("froyo_vit_tiny_l3", dev)
    tr = mixed.ExplainerTrainer(c["recipe"], exp)
    opt = torch.optim.AdamW([p for p in exp.parameters() if p.requires_grad], lr=1e-3, fused=True)
    engine.watch_optimizer(opt)
    for step in range(3):
        opt.zero_grad()
        tr.loss_and_grads(xs, bits, v0.to(dev), vs.to(dev), v1.to(dev), c["K"], train=True, seed=step)
        opt.step()
    phi_a, _ = tr.forward_phi(xs, v0.to(dev), v1.to(dev), False, 0)
    tr.saved = None
    from autognothi_amd.utils import synth
    fresh = c["recipe"].t_explainer(c["recipe"].t_config(**prm))
    fresh.load_state_dict({k: v.detach().clone() for k, v in exp.state_dict().items()})
    fresh = fresh.to(dev).train()
    phi_b, _ = mixed.ExplainerTrainer(c["recipe"], fresh).forward_phi(xs, v0.to(dev), v1.to(dev), False, 0)
    assert torch.equal(phi_a, phi_b)


def test_rows_kernels_against_numpy(cuda_device):
    """ag_rows_finish / ag_rows_ln_bwd / ag_slab_reduce / ag_colsum_bf16 / ag_cast_f32_many / ag_pad_cols_f32 one by one."""
    from autognothi_amd import ops
    dev = cuda_device
    g = np.random.default_rng(0)
    for m, h, s in [(197, 768, 3), (1030, 192, 1), (64, 1024, 4), (5, 96, 2)]:
        slabs = g.standard_normal((s, m, h)).astype(np.float32)
        bias = g.standard_normal(h).astype(np.float32)
        resid = g.standard_normal((m, h)).astype(np.float32)
        gam, bet = (1.0 + 0.1 * g.standard_normal(h)).astype(np.float32), (0.1 * g.standard_normal(h)).astype(np.float32)
        d = lambda a: torch.from_numpy(a).to(dev)   # noqa: E731
        t, zf, zb = ops.rows_finish(d(slabs), bias=d(bias), resid=d(resid), ln=(d(gam), d(bet), 1e-6), want_t=True, want_f32=True)
        x = slabs.astype(np.float64).sum(0) + bias + resid
        mu, var = x.mean(-1, keepdims=True), x.var(-1, keepdims=True)
        z = (x - mu) / np.sqrt(var + 1e-6) * gam + bet
        np.testing.assert_allclose(t.cpu().numpy(), x, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(zf.cpu().numpy(), z, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(zb.float().cpu().numpy(), z, rtol=1e-2, atol=1e-2)
        # LayerNorm backward against autograd
        xt = torch.from_numpy(x.astype(np.float32)).requires_grad_(True)
        gt, bt = torch.from_numpy(gam).requires_grad_(True), torch.from_numpy(bet).requires_grad_(True)
        dy_add = g.standard_normal((m, h)).astype(np.float32)
        add = g.standard_normal((m, h)).astype(np.float32)
        dy = torch.from_numpy(slabs).sum(0) + torch.from_numpy(dy_add)
        y = torch.nn.functional.layer_norm(xt, (h,), gt, bt, 1e-6)
        y.backward(dy)
        dg, db_, dbias = [torch.empty(h, device=dev) for _ in range(3)]
        dx, dxb = ops.rows_ln_bwd(d(slabs), x=d(x.astype(np.float32)), gamma=d(gam), eps=1e-6, dy_add=d(dy_add), add=d(add), want_bf16=True,
                                  dgamma=dg, dbeta=db_, dbias=dbias)
        want_dx = xt.grad.numpy() + add
        np.testing.assert_allclose(dx.cpu().numpy(), want_dx, rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(dxb.float().cpu().numpy(), want_dx, rtol=1e-2, atol=2e-2)
        np.testing.assert_allclose(dg.cpu().numpy(), gt.grad.numpy(), rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(db_.cpu().numpy(), bt.grad.numpy(), rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(dbias.cpu().numpy(), want_dx.astype(np.float64).sum(0), rtol=2e-4, atol=2e-3)
        # no LayerNorm: a plain sum
        dx2, _ = ops.rows_ln_bwd(d(slabs), dy_add=d(dy_add))
        np.testing.assert_allclose(dx2.cpu().numpy(), dy.numpy(), rtol=1e-5, atol=1e-5)
        # dropout: forward keep pattern == ag_dropout_f32 with the same seed on the same element indices
        t2, _, _ = ops.rows_finish(d(slabs[0]), p_drop=0.3, seed=77, want_t=True, want_bf16=False)
        np.testing.assert_array_equal(t2.cpu().numpy(), ops.dropout(d(slabs[0]), 0.3, 77).cpu().numpy())
        _, dxb3 = ops.rows_ln_bwd(d(slabs[0]), want_dx=False, want_bf16=True, p_drop=0.3, seed=77)
        np.testing.assert_array_equal(dxb3.float().cpu().numpy(), ops.dropout(d(slabs[0]), 0.3, 77).to(torch.bfloat16).float().cpu().numpy())
        np.testing.assert_allclose(ops.slab_reduce(d(slabs)).cpu().numpy(), slabs.astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    for m, n in [(1576, 2304), (8, 768), (1000, 3072), (129, 8)]:
        xb = torch.from_numpy(g.standard_normal((m, n)).astype(np.float32)).to(dev).to(torch.bfloat16)
        np.testing.assert_allclose(ops.colsum_bf16(xb).cpu().numpy(), xb.double().sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3)
    srcs = [torch.from_numpy(g.standard_normal(sz).astype(np.float32)).to(dev) for sz in [(768, 768), (10,), (3, 5000), (4096,), (17,)]]
    dsts = [torch.empty(s_.shape, dtype=torch.bfloat16 if i % 2 == 0 else torch.float32, device=dev) for i, s_ in enumerate(srcs)]
    ops.cast_many(list(zip(srcs, dsts)))
    for s_, d_ in zip(srcs, dsts):
        assert torch.equal(d_, s_.to(d_.dtype))
    src = torch.from_numpy(g.standard_normal((50, 10)).astype(np.float32)).to(dev)
    padded = ops.pad_cols(src, 16, ops.BF16)
    assert torch.equal(padded[:, :10], src.to(torch.bfloat16)) and float(padded[:, 10:].abs().max()) == 0.0
    assert torch.equal(ops.pad_cols(ops.pad_cols(src, 16, ops.F32), 10, ops.F32), src)


@pytest.mark.parametrize("mode", [0, 1])
def test_attention_bf16_io_matches_the_fp32_io_kernel(cuda_device, mode):
    """same MFMA kernel, bf16 I/O: forward equals the fp32-I/O form on bf16-representable inputs up to the output rounding; the
    backward from split-K slabs of dctx equals the backward from their sum."""
    from autognothi_amd import ops
    dev = cuda_device
    rows, t, heads = 3, 197, 3
    h = heads * 64
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn((rows * t, 3 * h), generator=g) * 0.7).to(torch.bfloat16).to(dev)
    masks = (torch.rand((rows, t - 1), generator=g) > 0.4).long()
    bits = ops.pack_mask(masks.to(dev))
    ctx16 = ops.masked_attention_train_bf16(qkv, bits, rows, t, h, heads, mode, 0.1, 9)
    ctx32 = ops.masked_attention_train(qkv.float(), bits, rows, t, h, heads, mode, 0.1, 9, mixed=True)
    np.testing.assert_allclose(ctx16.float().cpu().numpy(), ctx32.view(rows * t, h).cpu().numpy(), rtol=1e-2, atol=1e-2)
    dslabs = torch.randn((3, rows * t, h), generator=g).to(dev)
    dq16 = ops.masked_attention_bwd_bf16(qkv, bits, dslabs, rows, t, h, heads, mode, 0.1, 9)
    dq32 = ops.masked_attention_bwd(qkv.float().view(rows, t, 3 * h), bits, ctx32.view(rows, t, h), dslabs.sum(0).view(rows, t, h).contiguous(),
                                    rows, t, h, heads, mode, 0.1, 9, mixed=True)
    a, b = dq16.float().cpu().numpy(), dq32.view(rows * t, 3 * h).cpu().numpy()
    assert np.abs(a - b).max() <= 2e-2 * np.abs(b).max() + 1e-3


# 20-step trajectory of the bf16 step against the exact-fp32 step (same weights, inputs, masks and targets, dropout off, fused AdamW):
# the loss curves stay within LOSS_BAND of each other at every step, and the two parameter vectors end closer to each other than
# PARAM_DIST of the distance either has travelled.  Measured on MI355X (round 5) — see the printed line of the test.
TRAJ_STEPS, TRAJ_LOSS_BAND, TRAJ_PARAM_DIST = 20, 0.05, 0.5


@pytest.mark.parametrize("tag", ["duo_bert_base_l12", "froyo_vit_base_l12"])
def test_bf16_trajectory_tracks_fp32_at_full_depth(cuda_device, tag):
    """BASELINE config 5's explainers at the shipped depth (12 layers, K = 32): 20 optimiser steps in each training mode
    (reference loop: scripts/train_duo_explainer.py:180-198)."""
    from autognothi_amd import engine, ops, training
    from autognothi_amd.utils import synth
    dev = cuda_device
    c = build_case(tag)
    recipe, g = c["recipe"], c["g"]
    prm = _zero_dropout(c["meta"])
    engine.set_precision("fp32")
    xs = torch.from_numpy(c["xs"]).to(dev)
    v0, v1 = torch.from_numpy(g["v_0"]).to(dev), torch.from_numpy(g["v_1"]).to(dev)
    labels = torch.tensor([1][:c["B"]], dtype=torch.long, device=dev)
    c_ = v1.shape[-1]
    # per-step masks and targets, the same for both runs: the device sampler's stream, targets drawn around the fixture's grand value
    rng = ops.DeviceMT19937(dev, 3407)
    gen = torch.Generator(device="cpu").manual_seed(1)
    steps = []
    for _ in range(TRAJ_STEPS):
        _, bits = ops.mask_shapley_new(rng, c["B"] * c["K"], c["P"])
        vs = torch.softmax(torch.log(v1.cpu().repeat_interleave(c["K"], 0) + 1e-6) + 0.5 * torch.randn((c["B"] * c["K"], c_), generator=gen), -1)
        steps.append((bits, vs.to(dev)))

    def run(mixed):
        keep = training.MIXED_BF16
        training.MIXED_BF16 = mixed
        try:
            exp = recipe.t_explainer(recipe.t_config(**prm))
            synth.load_synth_weights(exp, seed=1)
            exp = exp.to(dev).train()
            start = {n: p.detach().clone() for n, p in exp.named_parameters() if p.requires_grad}
            tr = training.ExplainerTrainer(recipe, exp)
            opt = torch.optim.AdamW([p for p in exp.parameters() if p.requires_grad], lr=1e-5, fused=True)
            engine.watch_optimizer(opt)
            losses = []
            for bits, vs in steps:
                opt.zero_grad()
                loss, _ = tr.loss_and_grads(xs, bits, v0, vs, v1, c["K"], labels=labels, train=True, seed=0)
                losses.append(float(loss))
                opt.step()
            torch.cuda.synchronize()
            end = {n: p.detach().clone() for n, p in exp.named_parameters() if p.requires_grad}
            return np.array(losses), start, end
        finally:
            training.MIXED_BF16 = keep

    l32, s32, e32 = run(False)
    l16, s16, e16 = run(True)
    assert np.isfinite(l16).all() and np.isfinite(l32).all()
    band = float(np.abs(l16 - l32).max() / np.abs(l32).max())
    travelled = float(torch.sqrt(sum(((e32[n] - s32[n]).double() ** 2).sum() for n in e32)))
    apart = float(torch.sqrt(sum(((e32[n] - e16[n]).double() ** 2).sum() for n in e32)))
    print(f"{tag}: loss fp32 {l32[0]:.5f} -> {l32[-1]:.5f}, bf16 {l16[0]:.5f} -> {l16[-1]:.5f}; max |dloss| / max loss = {band:.4f}; "
          f"parameters {apart:.4e} apart after travelling {travelled:.4e} ({apart / travelled:.3f})")
    assert band <= TRAJ_LOSS_BAND, (band, l32, l16)
    assert apart <= TRAJ_PARAM_DIST * travelled, (apart, travelled)
