"""GPU: explainer / surrogate training step (HIP forward + backward) against torch autograd on the CPU port
(oracle/torch_port.py) — gradients of every trainable parameter, dropout off."""
import numpy as np
import pytest
import torch

from oracle import torch_port as otp
from util import LTT_TAGS, build_case

pytestmark = pytest.mark.gpu


def _zero_dropout(meta):
    prm = dict(meta["params"])
    prm["attention_probs_dropout_prob"] = 0.0
    prm["hidden_dropout_prob"] = 0.0
    return prm


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("tag", ["froyo_vit_tiny_l3", "vit_tiny_c1", "bert_base_l2", "duo_vit_tiny_l3", "duo_bert_base_l2"])
def test_explainer_step_gradients_match_autograd(cuda_device, tag):
    from autognothi_amd.training import ExplainerTrainer
    from autognothi_amd import ops
    c = build_case(tag)
    dev, g, recipe = cuda_device, c["g"], c["recipe"]
    prm = _zero_dropout(c["meta"])
    if tag == "vit_tiny_c1":
        prm["num_hidden_layers"] = 2  # keep the CPU autograd reference quick
    cfg = recipe.t_config(**prm)
    exp = recipe.t_explainer(cfg)
    from autognothi_amd.utils import synth
    synth.load_synth_weights(exp, seed=1)
    exp = exp.to(dev)
    exp.train()                      # froyo: freezes vit.* (models/froyo_vit.py:88-97)
    kind = c["meta"]["kind"]
    xs = torch.from_numpy(c["xs"]).to(dev)
    masks = torch.from_numpy(c["masks"])
    bits = ops.pack_mask(masks.to(dev))
    v0, vs, v1 = [torch.from_numpy(g[k]) for k in ("v_0", "v_s", "v_1")]
    tr = ExplainerTrainer(recipe, exp)
    duo = c["meta"]["duo"]
    labels = torch.tensor([1, 0][:c["B"]], dtype=torch.long)
    loss, phi = tr.loss_and_grads(xs, bits, v0.to(dev), vs.to(dev), v1.to(dev), c["K"], labels=labels.to(dev), train=True)
    # ---- CPU autograd reference on the same weights ----
    sd = {k: v.detach().cpu().clone().requires_grad_(exp.state_dict(keep_vars=True)[k].requires_grad)
          for k, v in exp.state_dict(keep_vars=True).items()}
    ones = torch.ones((c["B"], c["P"]), dtype=torch.long)
    phi_ref, z_ref = otp.explainer_phi(torch.from_numpy(c["xs"]), ones, v1, v0, sd, prm, kind)
    loss_ref = otp.shapley_loss(masks.reshape(c["B"], c["K"], c["P"]), v0, vs, phi_ref, c["P"])
    if duo:  # scripts/train_duo_explainer.py:184-195: + cross_entropy(base_Ys, Zs)
        lin = torch.nn.functional.linear
        if kind == "vit":   # CE applied on the soft-maxed head output (duo_vanilla_vit.py:121-122)
            base = torch.softmax(lin(z_ref[:, 0], sd["classifier.weight"], sd["classifier.bias"]), -1)
        else:               # raw logits through the pooler (duo_vanilla_bert.py:142-144)
            base = lin(torch.tanh(lin(z_ref[:, 0], sd["bert_pooler.dense.weight"], sd["bert_pooler.dense.bias"])),
                       sd["classifier.weight"], sd["classifier.bias"])
        loss_ref = loss_ref + torch.nn.functional.cross_entropy(base, labels)
    loss_ref.backward()
    np.testing.assert_allclose(loss.cpu().numpy()[0], loss_ref.item(), rtol=2e-4)
    np.testing.assert_allclose(phi.cpu().numpy(), phi_ref.detach().numpy(), rtol=1e-3, atol=2e-5)
    checked = 0
    gscale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for name, p in exp.named_parameters():
        ref = sd[name].grad
        if not p.requires_grad:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert ref is not None and p.grad is not None, name
        got = p.grad.cpu().numpy()
        if float(ref.abs().max()) < 1e-5 * gscale:   # structurally zero gradients (e.g. key.bias: soft-max shift invariance)
            assert float(np.abs(got).max()) < 1e-4 * gscale, name
            continue
        assert _rel(got, ref.numpy()) < 2e-3, (name, _rel(got, ref.numpy()))
        checked += 1
    assert checked >= 10


def test_surrogate_step_gradients_match_autograd(cuda_device):
    from autognothi_amd.training import SurrogateTrainer
    from autognothi_amd import ops
    from autognothi_amd.utils import synth
    c = build_case("vit_tiny_c1")
    dev, recipe = cuda_device, c["recipe"]
    prm = _zero_dropout(c["meta"]); prm["num_hidden_layers"] = 2
    cfg = recipe.t_config(**prm)
    srg = recipe.t_surrogate(cfg)
    synth.load_synth_weights(srg, seed=0)
    srg = srg.to(dev).train()
    xs = torch.from_numpy(c["xs"]).to(dev)
    masks = torch.from_numpy(c["masks"][:c["B"]])
    bits = ops.pack_mask(masks.to(dev))
    orig = torch.softmax(torch.from_numpy(np.random.default_rng(2).standard_normal((c["B"], 10)).astype(np.float32)), -1)
    tr = SurrogateTrainer(recipe, srg)
    loss, probs = tr.loss_and_grads(xs, bits, orig.to(dev), train=True)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in srg.state_dict().items()}
    mask_t = torch.cat([torch.ones((c["B"], 1), dtype=torch.long), masks], 1)
    z = otp.vit_backbone(torch.from_numpy(c["xs"]), mask_t, sd, prm)
    p_ref = torch.softmax(torch.nn.functional.linear(z[:, 0], sd["classifier.weight"], sd["classifier.bias"]), -1)
    l_ref = torch.nn.functional.kl_div(torch.log_softmax(orig, -1), torch.softmax(p_ref, -1), reduction="batchmean")
    l_ref.backward()
    np.testing.assert_allclose(loss.cpu().numpy()[0], l_ref.item(), rtol=1e-3, atol=1e-7)
    gscale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for name, p in srg.named_parameters():
        r = sd[name].grad
        if float(r.abs().max()) < 1e-5 * gscale:
            assert float(p.grad.abs().max()) < 1e-4 * gscale, name
            continue
        assert _rel(p.grad.cpu().numpy(), r.numpy()) < 3e-3, (name, _rel(p.grad.cpu().numpy(), r.numpy()))


def test_explainer_train_epoch_reduces_loss(cuda_device):
    """End-to-end: a few optimiser steps of explainer_epoch_train (dropout on, AdamW) lower the Shapley loss."""
    from autognothi_amd import engine
    from autognothi_amd.scripts import train_explainer as te
    c = build_case("froyo_vit_tiny_l3")
    dev, recipe = cuda_device, c["recipe"]
    srg, exp = c["surrogate"].to(dev), c["explainer"].to(dev)
    xs = torch.from_numpy(c["xs"]).to(dev)
    engine.set_precision("fp32")
    v0 = torch.from_numpy(c["g"]["v_0"]).to(dev)
    opt = torch.optim.AdamW([p for p in exp.parameters()], lr=1e-3)
    items = [(None, None)] * 3
    gen = lambda a, b: (xs, torch.zeros(c["B"], dtype=torch.long, device=dev))  # noqa: E731
    first = te.explainer_epoch_train(None, dev, 4, c["P"], v0, items, recipe, srg, exp, opt, 1, gen, seed=1)
    for e in range(2, 6):
        last = te.explainer_epoch_train(None, dev, 4, c["P"], v0, items, recipe, srg, exp, opt, e, gen, seed=1)
    assert last < first
    for n, p in exp.named_parameters():
        if n.startswith("vit."):
            assert not p.requires_grad


def _check_grads(module, sd, tol):
    gscale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    checked = 0
    for name, p in module.named_parameters():
        ref = sd[name].grad
        if not p.requires_grad:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert ref is not None and p.grad is not None, name
        got = p.grad.cpu().numpy()
        if float(ref.abs().max()) < 1e-5 * gscale:
            assert float(np.abs(got).max()) < 1e-4 * gscale, name
            continue
        assert _rel(got, ref.numpy()) < tol, (name, _rel(got, ref.numpy()))
        checked += 1
    return checked


@pytest.mark.parametrize("tag", LTT_TAGS)
def test_ltt_explainer_step_gradients_match_autograd(cuda_device, tag):
    """LTT: only the ladder (maps + narrow side layers + side LN) and the side explainer head train; the backbone is
    frozen (reference models/ltt_vit.py:132-138).  Gradients vs torch autograd on the CPU port, dropout off."""
    from autognothi_amd import ops
    from autognothi_amd.training import make_explainer_trainer
    from autognothi_amd.utils import synth
    c = build_case(tag)
    dev, g, recipe, kind = cuda_device, c["g"], c["recipe"], c["meta"]["kind"]
    prm = _zero_dropout(c["meta"])
    cfg = recipe.t_config(**prm)
    exp = recipe.t_explainer(cfg)
    synth.load_synth_weights(exp, seed=1)
    exp = exp.to(dev)
    exp.train()
    xs = torch.from_numpy(c["xs"]).to(dev)
    masks = torch.from_numpy(c["masks"])
    bits = ops.pack_mask(masks.to(dev))
    v0, vs, v1 = [torch.from_numpy(g[k]) for k in ("v_0", "v_s", "v_1")]
    tr = make_explainer_trainer(recipe, exp)
    loss, phi = tr.loss_and_grads(xs, bits, v0.to(dev), vs.to(dev), v1.to(dev), c["K"], train=True)
    sd = {k: v.detach().cpu().clone().requires_grad_(exp.state_dict(keep_vars=True)[k].requires_grad)
          for k, v in exp.state_dict(keep_vars=True).items()}
    ones = torch.ones((c["B"], c["P"]), dtype=torch.long)
    phi_ref = otp.ltt_explainer_phi(torch.from_numpy(c["xs"]), ones, v1, v0, sd, prm, kind)
    loss_ref = otp.shapley_loss(masks.reshape(c["B"], c["K"], c["P"]), v0, vs, phi_ref, c["P"])
    loss_ref.backward()
    np.testing.assert_allclose(loss.cpu().numpy()[0], loss_ref.item(), rtol=2e-4)
    np.testing.assert_allclose(phi.cpu().numpy(), phi_ref.detach().numpy(), rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(phi.cpu().numpy(), g["phi"], rtol=1e-3, atol=1e-4 * float(np.abs(g["phi"]).max()))
    assert _check_grads(exp, sd, 2e-3) >= 20


@pytest.mark.parametrize("tag", LTT_TAGS)
def test_ltt_surrogate_step_gradients_match_autograd(cuda_device, tag):
    from autognothi_amd import ops
    from autognothi_amd.training import make_surrogate_trainer
    from autognothi_amd.utils import synth
    c = build_case(tag)
    dev, recipe, kind = cuda_device, c["recipe"], c["meta"]["kind"]
    prm = _zero_dropout(c["meta"])
    cfg = recipe.t_config(**prm)
    srg = recipe.t_surrogate(cfg)
    synth.load_synth_weights(srg, seed=0)
    srg = srg.to(dev).train()
    xs = torch.from_numpy(c["xs"]).to(dev)
    masks = torch.from_numpy(c["masks"][:c["B"]])
    bits = ops.pack_mask(masks.to(dev))
    ncls = prm["num_labels"]
    orig = torch.softmax(torch.from_numpy(np.random.default_rng(2).standard_normal((c["B"], ncls)).astype(np.float32)), -1)
    tr = make_surrogate_trainer(recipe, srg)
    loss, probs = tr.loss_and_grads(xs, bits, orig.to(dev), train=True)
    sd = {k: v.detach().cpu().clone().requires_grad_(srg.state_dict(keep_vars=True)[k].requires_grad)
          for k, v in srg.state_dict(keep_vars=True).items()}
    p_ref = otp.ltt_surrogate_probs(torch.from_numpy(c["xs"]), masks, sd, prm, kind)
    l_ref = torch.nn.functional.kl_div(torch.log_softmax(orig, -1), torch.softmax(p_ref, -1), reduction="batchmean")
    l_ref.backward()
    np.testing.assert_allclose(probs.cpu().numpy(), p_ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(loss.cpu().numpy()[0], l_ref.item(), rtol=1e-3, atol=1e-7)
    assert _check_grads(srg, sd, 3e-3) >= 20


@pytest.mark.parametrize("tag", LTT_TAGS)
def test_ltt_train_epochs_run_and_reduce_loss(cuda_device, tag):
    """End-to-end on the LTT recipes: surrogate epochs (KL of the side head against the frozen backbone) and explainer
    epochs (Shapley loss), dropout on, AdamW over the trainable (ladder + side head) parameters only."""
    from autognothi_amd import engine
    from autognothi_amd.scripts import train_explainer as te
    from autognothi_amd.scripts import train_surrogate as ts
    c = build_case(tag)
    dev, recipe, cfg = cuda_device, c["recipe"], c["cfg"]
    engine.set_precision("fp32")
    cls = recipe.t_classifier(cfg)
    cls.load_state_dict(c["surrogate"].state_dict())
    cls = cls.to(dev).eval()
    srg, exp = c["surrogate"].to(dev), c["explainer"].to(dev)
    xs = torch.from_numpy(c["xs"]).to(dev)
    gen = lambda a, b: (xs, torch.zeros(c["B"], dtype=torch.long, device=dev))  # noqa: E731
    items = [(None, None)] * 3
    srg.train()
    trainable = [p for p in srg.parameters() if p.requires_grad]
    assert 0 < len(trainable) < len(list(srg.parameters()))
    opt_s = torch.optim.AdamW(trainable, lr=2e-3)
    first = ts.surrogate_epoch_train(None, dev, c["P"], items, recipe, cls, srg, opt_s, 1, gen, seed=1)
    for e in range(2, 6):
        last = ts.surrogate_epoch_train(None, dev, c["P"], items, recipe, cls, srg, opt_s, e, gen, seed=1)
    assert np.isfinite(first) and np.isfinite(last) and last < first
    before = {n: p.detach().clone() for n, p in srg.named_parameters() if not p.requires_grad}
    for n, p in srg.named_parameters():   # the frozen backbone did not move
        if n in before:
            assert torch.equal(p, before[n])
    srg.eval()
    v0 = torch.from_numpy(c["g"]["v_0"]).to(dev)
    exp.train()
    opt_e = torch.optim.AdamW([p for p in exp.parameters() if p.requires_grad], lr=1e-3)
    first = te.explainer_epoch_train(None, dev, 4, c["P"], v0, items, recipe, srg, exp, opt_e, 1, gen, seed=1)
    for e in range(2, 6):
        last = te.explainer_epoch_train(None, dev, 4, c["P"], v0, items, recipe, srg, exp, opt_e, e, gen, seed=1)
    assert np.isfinite(first) and last < first
    ev = te.explainer_epoch_eval(None, dev, 4, c["P"], v0, items[:1], recipe, srg, exp, 1, gen, seed=3407)
    assert np.isfinite(ev)


def test_mixed_precision_training_step(cuda_device):
    """Opt-in throughput mode (training.MIXED_BF16): bf16 GEMM operands, fp32 accumulate/activations.  Gradients stay within
    bf16 operand-rounding distance of fp32 autograd and a few optimiser steps still lower the loss."""
    from autognothi_amd import ops, training
    from autognothi_amd.utils import synth
    c = build_case("froyo_vit_tiny_l3")
    dev, g, recipe = cuda_device, c["g"], c["recipe"]
    prm = _zero_dropout(c["meta"])
    cfg = recipe.t_config(**prm)
    exp = recipe.t_explainer(cfg)
    synth.load_synth_weights(exp, seed=1)
    exp = exp.to(dev)
    exp.train()
    xs = torch.from_numpy(c["xs"]).to(dev)
    masks = torch.from_numpy(c["masks"])
    bits = ops.pack_mask(masks.to(dev))
    v0, vs, v1 = [torch.from_numpy(g[k]) for k in ("v_0", "v_s", "v_1")]
    training.MIXED_BF16 = True
    try:
        tr = training.ExplainerTrainer(recipe, exp)
        loss, phi = tr.loss_and_grads(xs, bits, v0.to(dev), vs.to(dev), v1.to(dev), c["K"], train=True)
        sd = {k: v.detach().cpu().clone().requires_grad_(exp.state_dict(keep_vars=True)[k].requires_grad)
              for k, v in exp.state_dict(keep_vars=True).items()}
        ones = torch.ones((c["B"], c["P"]), dtype=torch.long)
        phi_ref, _ = otp.explainer_phi(torch.from_numpy(c["xs"]), ones, v1, v0, sd, prm, "vit")
        loss_ref = otp.shapley_loss(masks.reshape(c["B"], c["K"], c["P"]), v0, vs, phi_ref, c["P"])
        loss_ref.backward()
        np.testing.assert_allclose(loss.cpu().numpy()[0], loss_ref.item(), rtol=3e-2)
        assert _check_grads(exp, sd, 6e-2) >= 10
        opt = torch.optim.AdamW([p for p in exp.parameters() if p.requires_grad], lr=1e-3)
        first = last = None
        for step in range(6):
            opt.zero_grad()
            loss, _ = tr.loss_and_grads(xs, bits, v0.to(dev), vs.to(dev), v1.to(dev), c["K"], train=True, seed=step)
            opt.step()
            first = float(loss) if first is None else first
            last = float(loss)
        assert last < first
    finally:
        training.MIXED_BF16 = False


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("t,heads,rows,p_drop", [(197, 3, 4, 0.0), (197, 2, 3, 0.1), (128, 2, 4, 0.1), (33, 1, 2, 0.0), (256, 1, 2, 0.0)])
def test_attention_backward_mixed_matches_fp32(cuda_device, mode, t, heads, rows, p_drop):
    """the matrix-core backward (bf16 operands) against the exact-fp32 kernels on the same inputs, masks and dropout
    decisions: dQ, dK, dV within bf16 operand rounding."""
    from autognothi_amd import ops
    h = heads * 64
    g = torch.Generator().manual_seed(t * 3 + heads + mode)
    qkv = torch.randn((rows, t, 3 * h), generator=g).to(cuda_device)
    dctx = torch.randn((rows, t, h), generator=g).to(cuda_device)
    mask = (torch.rand((rows, t - 1), generator=g) < 0.6).to(torch.int64)
    mask[0] = 0
    mask[-1] = 1
    bits = ops.pack_mask(mask.to(cuda_device))
    ctx = ops.masked_attention_train(qkv, bits, rows, t, h, heads, mode, p_drop, 77)
    ctx_mixed = ops.masked_attention_train(qkv, bits, rows, t, h, heads, mode, p_drop, 77, mixed=True)
    np.testing.assert_allclose(ctx_mixed.cpu().numpy(), ctx.cpu().numpy(), rtol=0, atol=3e-2)   # the forward of the same mode
    ref = ops.masked_attention_bwd(qkv, bits, ctx, dctx, rows, t, h, heads, mode, p_drop, 77).cpu().numpy()
    got = ops.masked_attention_bwd(qkv, bits, ctx, dctx, rows, t, h, heads, mode, p_drop, 77, mixed=True).cpu().numpy()
    assert np.isfinite(got).all()
    for name, sl in (("dQ", slice(0, h)), ("dK", slice(h, 2 * h)), ("dV", slice(2 * h, 3 * h))):
        r, o = ref[..., sl], got[..., sl]
        scale = float(np.abs(r).max())
        assert float(np.abs(o - r).max()) <= 3e-2 * scale, (name, float(np.abs(o - r).max()), scale)
        assert float(np.abs(o - r).mean()) <= 4e-3 * scale, name


@pytest.mark.parametrize("mixed", [False, True])
@pytest.mark.parametrize("tag", ["vit_tiny_c1", "duo_bert_base_l2"])
def test_fused_optimizer_leaves_no_stale_weight_cache(cuda_device, tag, mixed):
    """torch.optim.AdamW(fused=True) — what train_explainer(env, device) builds — updates parameters without advancing
    ``_version``.  After a few steps both families of weight caches must still follow the parameters: the trainer's operand
    forms (fused q|k|v, bf16 W / W^T pairs) and the inference packs (engine.Packed*): the same module must compute what a
    fresh copy of its state dict computes, in the training forward and in fw_explainer, and must differ from step 0."""
    from autognothi_amd import engine, ops, training as T
    from autognothi_amd.utils import synth
    c = build_case(tag)
    dev, g, recipe = cuda_device, c["g"], c["recipe"]
    engine.set_precision("fp32")
    prm = _zero_dropout(c["meta"])
    cfg = recipe.t_config(**prm)
    exp = recipe.t_explainer(cfg)
    synth.load_synth_weights(exp, seed=1)
    exp = exp.to(dev)
    exp.train()
    xs = torch.from_numpy(c["xs"]).to(dev)
    bits = ops.pack_mask(torch.from_numpy(c["masks"]).to(dev))
    v0, vs, v1 = [torch.from_numpy(g[k]).to(dev) for k in ("v_0", "v_s", "v_1")]
    labels = torch.tensor([1, 0][:c["B"]], dtype=torch.long, device=dev)
    ones = torch.ones((c["B"], c["P"]), dtype=torch.long, device=dev)
    T.MIXED_BF16 = mixed
    try:
        tr = T.make_explainer_trainer(recipe, exp)
        opt = torch.optim.AdamW([q for q in exp.parameters() if q.requires_grad], lr=1e-3, fused=True)
        with torch.no_grad():
            phi_start, _ = recipe.fw_explainer(exp, xs, ones, v1, v0)
        for _ in range(3):
            opt.zero_grad()
            tr.loss_and_grads(xs, bits, v0, vs, v1, c["K"], labels=labels, train=True, seed=3)
            opt.step()
        fresh = recipe.t_explainer(cfg)
        fresh.load_state_dict(exp.state_dict())
        fresh = fresh.to(dev)
        fresh.train()
        phi_a, base_a = tr.forward_phi(xs, v0, v1, False, 0)
        phi_b, base_b = T.make_explainer_trainer(recipe, fresh).forward_phi(xs, v0, v1, False, 0)
        assert torch.equal(phi_a, phi_b), float((phi_a - phi_b).abs().max())
        if base_a is not None:
            assert torch.equal(base_a, base_b)
        exp.eval(); fresh.eval()
        with torch.no_grad():
            inf_a, _ = recipe.fw_explainer(exp, xs, ones, v1, v0)
            inf_b, _ = recipe.fw_explainer(fresh, xs, ones, v1, v0)
        assert torch.equal(inf_a, inf_b), float((inf_a - inf_b).abs().max())
        assert float((inf_a - phi_start).abs().max()) > 1e-4          # (and the three steps did move the explainer)
    finally:
        T.MIXED_BF16 = False


@pytest.mark.parametrize("r,c", [(1576, 768), (1576, 3072), (1000, 100), (37, 8), (64, 64), (130, 2304), (5, 10)])
def test_operand_form_kernels(cuda_device, r, c):
    """the passes that make a mixed-precision Linear's bf16 operand forms ([R,C] and the zero-padded transpose [C,Rp]) — plain,
    with GELU applied on the way (fc2's operand) and with GELU' (fc1's incoming gradient, + its fp32 copy) — against torch;
    64x64 vector kernel for widths that are multiples of 4, the 32x32 scalar one otherwise (c = 10: num_labels)."""
    from autognothi_amd import ops
    g = torch.Generator().manual_seed(r * 7 + c)
    x = torch.randn((r, c), generator=g) * 1.5
    dy = torch.randn((r, c), generator=g)
    xd, dyd = x.to(cuda_device), dy.to(cuda_device)
    rp = (r + 31) // 32 * 32

    def check(plain, tr, want):
        wb = want.to(torch.bfloat16)
        assert plain.shape == (r, c) and tr.shape == (c, rp)
        assert torch.equal(plain.cpu(), wb)
        assert torch.equal(tr[:, :r].cpu(), wb.t())
        assert not bool(tr[:, r:].any())                      # the padding is written (zeros): the GEMM reads it

    check(*ops.cast_transpose_bf16(xd, pad_cols_to=32), x)
    assert torch.equal(ops.transpose_bf16(xd, pad_cols_to=32)[:, :r].cpu(), x.to(torch.bfloat16).t())
    if c % 4 == 0:
        gl = torch.nn.functional.gelu(x.double()).float()
        p_, t_ = ops.gelu_cast_transpose_bf16(xd, pad_cols_to=32)
        # erff / expf differ from torch's by an ulp or two of fp32: compare in fp32 with a bf16-ulp allowance at rounding ties
        np.testing.assert_allclose(p_.float().cpu().numpy(), gl.numpy(), rtol=8e-3, atol=1e-6)
        assert torch.equal(t_[:, :r], p_.t()) and not bool(t_[:, r:].any())
        xg = x.double().requires_grad_(True)
        torch.nn.functional.gelu(xg).backward(dy.double())
        du, p2, t2 = ops.gelu_bwd_cast_transpose_bf16(xd, dyd, pad_cols_to=32)
        np.testing.assert_allclose(du.cpu().numpy(), xg.grad.float().numpy(), rtol=1e-5, atol=1e-6)
        assert torch.equal(p2, du.to(torch.bfloat16)) and torch.equal(t2[:, :r], p2.t()) and not bool(t2[:, r:].any())
        assert torch.equal(du, ops.gelu_bwd(xd, dyd))          # the fused form is the stand-alone kernel's arithmetic


def test_fused_residual_kernels(cuda_device):
    """resid + dropout(x) in one pass == the two kernels it replaces (same keep decisions), p = 0 == plain add; LayerNorm backward
    with the residual-branch gradient added in the same pass == layernorm_bwd + add."""
    from autognothi_amd import ops
    g = torch.Generator().manual_seed(5)
    x, res, dy = [torch.randn((777, 768), generator=g).to(cuda_device) for _ in range(3)]
    for p in (0.0, 0.1, 0.5):
        want = ops.add(res, ops.dropout(x, p, 1234))
        assert torch.equal(ops.dropout_add(x, res, p, 1234), want)
    gamma = (1 + 0.1 * torch.randn(768, generator=g)).to(cuda_device)
    dg1, db1 = torch.zeros(768, device=cuda_device), torch.zeros(768, device=cuda_device)
    dg2, db2 = torch.zeros(768, device=cuda_device), torch.zeros(768, device=cuda_device)
    a = ops.add(ops.layernorm_bwd(x, gamma, dy, 1e-12, dg1, db1), res)
    b = ops.layernorm_bwd(x, gamma, dy, 1e-12, dg2, db2, add=res)
    torch.testing.assert_close(b, a, rtol=0, atol=0)
    torch.testing.assert_close(dg2, dg1, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(db2, db1, rtol=1e-5, atol=1e-5)
