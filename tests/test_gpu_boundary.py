"""The drop-in boundary under TRAINING callers: the reference's own epoch bodies — restated here call for call
(scripts/train_explainer.py:128-207, scripts/train_duo_explainer.py:121-213, scripts/train_surrogate.py:112-160) — run
against this package's recipes and its ``models.shapley`` drop-in with nothing but ``recipe.fw_*``, ``loss.backward()`` and
``optimizer.step()``; no import of autognothi_amd.training / autognothi_amd.scripts in this file.  Checked against
(1) fixtures of ONE reference _explainer_epoch_train step run by the reference itself (gradients seen by the optimiser,
post-step parameters, loss; tests/golden/train_step_*.npz) and (2) torch autograd on the pinned CPU port."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import torch_port as otp
from util import GOLDEN, build_case, golden, unpack

pytestmark = pytest.mark.gpu


class _Env:
    def __init__(self):
        self.lines = []

    def log(self, msg):
        self.lines.append(msg)


# ---- the reference loop bodies, restated against the drop-in surface ------------------------------------------------
def _explainer_epoch_train(env, device, n_mask_samples, n_players, surrogate_null, d_items, m_recipe, m_surrogate,
                           m_explainer, optimizer, epoch, gen_input):
    """scripts/train_explainer.py:128-207, same calls in the same order."""
    from autognothi_amd.models.shapley import loss_shapley_new, mask_shapley_new
    reg_loss, total = 0.0, 0
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        Xs, _Zs = gen_input(_inputs, _targets)
        batch_size = Xs.shape[0]
        Xs_mask_1 = torch.ones((batch_size, n_players), dtype=torch.long, device=device)
        Xs_mask_shap_ = mask_shapley_new(batch_size * n_mask_samples, n_players).to(device)
        Xs_mask_shap = Xs_mask_shap_.reshape((batch_size, n_mask_samples, n_players))
        Xs_EXT = torch.stack([Xs[b] for b in range(batch_size) for _ in range(n_mask_samples)], dim=0)
        optimizer.zero_grad()
        m_surrogate.eval()
        with torch.no_grad():
            surrogate_values, _ = m_recipe.fw_surrogate(m_surrogate, Xs_EXT, Xs_mask_shap_)
            surrogate_grand, _ = m_recipe.fw_surrogate(m_surrogate, Xs, Xs_mask_1)
        optimizer.zero_grad()
        m_explainer.train()
        explainer_shap, _ = m_recipe.fw_explainer(m_explainer, Xs, Xs_mask_1, surrogate_grand, surrogate_null)
        loss_shap = loss_shapley_new(batch_size=batch_size, n_mask_samples=n_mask_samples, n_players=n_players,
                                     mask=Xs_mask_shap, v_0=surrogate_null, v_s=surrogate_values, v_1=surrogate_grand,
                                     phi=explainer_shap)
        loss_shap.backward()
        optimizer.step()
        reg_loss += loss_shap.item()
        total += batch_size
        env.log(f"  > epoch {epoch} :{batch_idx}:train // loss: shap {loss_shap.item() / batch_size:.6f}, fin {total}")
    return reg_loss / total


def _duo_explainer_epoch_train(env, device, n_mask_samples, n_players, surrogate_null, d_items, m_recipe, m_surrogate,
                               m_explainer, optimizer, epoch, gen_input):
    """scripts/train_duo_explainer.py:121-213."""
    from autognothi_amd.models.shapley import loss_shapley_new, mask_shapley_new
    tot_loss, total = 0.0, 0
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        Xs, Zs = gen_input(_inputs, _targets)
        batch_size = Xs.shape[0]
        Xs_mask_1 = torch.ones((batch_size, n_players), dtype=torch.long, device=device)
        Xs_mask_shap_ = mask_shapley_new(batch_size * n_mask_samples, n_players).to(device)
        Xs_mask_shap = Xs_mask_shap_.reshape((batch_size, n_mask_samples, n_players))
        Xs_EXT = torch.stack([Xs[b] for b in range(batch_size) for _ in range(n_mask_samples)], dim=0)
        optimizer.zero_grad()
        m_surrogate.eval()
        with torch.no_grad():
            surrogate_values, _ = m_recipe.fw_surrogate(m_surrogate, Xs_EXT, Xs_mask_shap_)
            surrogate_grand, _ = m_recipe.fw_surrogate(m_surrogate, Xs, Xs_mask_1)
        optimizer.zero_grad()
        m_explainer.train()
        explainer_shap, base_Ys = m_recipe.fw_explainer(m_explainer, Xs, Xs_mask_1, surrogate_grand, surrogate_null)
        assert base_Ys is not None
        loss_cls = torch.nn.functional.cross_entropy(base_Ys, Zs)
        loss_shap = loss_shapley_new(batch_size=batch_size, n_mask_samples=n_mask_samples, n_players=n_players,
                                     mask=Xs_mask_shap, v_0=surrogate_null, v_s=surrogate_values, v_1=surrogate_grand,
                                     phi=explainer_shap)
        loss = loss_cls + loss_shap
        loss.backward()
        optimizer.step()
        tot_loss += loss.item()
        total += batch_size
    return tot_loss / total


def _surrogate_epoch_train(env, device, n_players, d_items, m_recipe, m_classifier, m_surrogate, optimizer, epoch, gen_input):
    """scripts/train_surrogate.py:112-160."""
    from autognothi_amd.models.shapley import loss_logits_kl_divergence, mask_purely_uniform
    kld, total = 0.0, 0
    for batch_idx, (_inputs, _targets) in enumerate(d_items):
        Xs, Zs = gen_input(_inputs, _targets)
        batch_size = Xs.shape[0]
        Xs_mask_1 = torch.ones((batch_size, n_players), dtype=torch.long, device=device)
        Xs_mask_rand = mask_purely_uniform(batch_size, n_players).to(device)
        optimizer.zero_grad()
        m_classifier.eval()
        with torch.no_grad():
            _, orig_Ys = m_recipe.fw_classifier(m_classifier, Xs, Xs_mask_1)
        optimizer.zero_grad()
        m_surrogate.train()
        adapt_Ys, _ = m_recipe.fw_surrogate(m_surrogate, Xs, Xs_mask_rand)
        loss_kld = loss_logits_kl_divergence(orig_Ys, adapt_Ys)
        loss_kld.backward()
        with torch.no_grad():
            torch.nn.functional.cross_entropy(adapt_Ys, Zs)
        optimizer.step()
        kld += loss_kld.item()
        total += batch_size
    return kld / total


# ---- helpers ---------------------------------------------------------------------------------------------------------
def _capture_step(optimizer, module):
    """wrap optimizer.step: record the gradients it sees, run the real step (as make_golden.py's gen_train_step does)."""
    seen = {}
    real = optimizer.step

    def step(*a, **k):
        for n, p in module.named_parameters():
            seen[n] = None if p.grad is None else p.grad.detach().clone()
        return real(*a, **k)
    optimizer.step = step
    return seen


def _sample_idx(n):
    return np.linspace(0, n - 1, min(n, 16)).astype(np.int64)


@pytest.mark.parametrize("tag", ["vit_tiny_l2", "froyo_vit_tiny_l2", "bert_base_l2"])
def test_reference_explainer_step_fixture(cuda_device, tag):
    """one reference _explainer_epoch_train step (dropout 0, AdamW 1e-3, torch.manual_seed(3407)) ->
    the loop above on the HIP path reproduces the loss (hence the masks), every gradient the optimiser saw and the
    post-step parameters."""
    from autognothi_amd import engine
    from autognothi_amd.recipes import get_recipe
    from autognothi_amd.utils import synth
    from util import recipe_kind
    with open(os.path.join(GOLDEN, f"train_step_{tag}.json")) as f:
        meta = json.load(f)
    g = golden(f"train_step_{tag}.npz")
    dev = cuda_device
    engine.set_precision("fp32")
    recipe = get_recipe(recipe_kind({"kind": meta["kind"], "duo": False, "froyo": meta["froyo"]}))
    cfg = recipe.t_config(**meta["params"])
    b, k, p = [int(x) for x in g["dims"]]
    srg, exp = recipe.t_surrogate(cfg), recipe.t_explainer(cfg)
    synth.load_synth_weights(srg, seed=0)
    synth.load_synth_weights(exp, seed=1)
    srg, exp = srg.to(dev), exp.to(dev)
    prm = meta["params"]
    if meta["kind"] == "vit":
        xs = torch.from_numpy(synth.synth_images(b, prm["img_px_size"], prm["img_channels"], seed=0)).to(dev)
        null = torch.zeros((1, prm["img_channels"], prm["img_px_size"], prm["img_px_size"]), device=dev)
    else:
        xs = torch.from_numpy(g["ids"]).to(dev)
        null = torch.from_numpy(synth.synth_null_ids(prm["max_position_embeddings"], prm["vocab_size"])).to(dev)
    srg.eval()
    with torch.no_grad():
        v_0, _ = recipe.fw_surrogate(srg, null, torch.ones((1, p), dtype=torch.long, device=dev))
    np.testing.assert_allclose(v_0.cpu().numpy(), g["v_0"], rtol=1e-4, atol=1e-5)
    lr = float(g["lr"][0])
    before = {n: q.detach().clone() for n, q in exp.named_parameters()}
    opt = torch.optim.AdamW(exp.parameters(), lr=lr)
    seen = _capture_step(opt, exp)
    torch.manual_seed(meta["torch_seed"])
    env = _Env()
    loss = _explainer_epoch_train(env, dev, k, p, v_0, [(None, None)], recipe, srg, exp, opt, 1,
                                  lambda a, b_: (xs, torch.zeros(b, dtype=torch.long, device=dev)))
    np.testing.assert_allclose(loss, float(g["loss_mean"][0]), rtol=2e-4)
    assert len(env.lines) == 1 and ":0:train" in env.lines[0]
    trained = set(meta["trained"])
    gscale = max(float(g["g/" + n][2]) for n in trained)        # largest gradient element of the step
    for n, q in exp.named_parameters():
        if n in meta["frozen"]:
            assert not q.requires_grad and seen[n] is None and torch.equal(q, before[n]), n
            continue
        assert n in trained and seen[n] is not None, n
        gw = g["g/" + n]
        gsum, gabs, gmax, gsamp = gw[0], gw[1], gw[2], gw[3:]
        got = seen[n].reshape(-1).double().cpu()
        idx = _sample_idx(got.numel())
        if gmax < 1e-5 * gscale:   # structurally zero in the reference (key biases: soft-max shift invariance): rounding noise
            assert float(got.abs().max()) < 1e-4 * gscale, n
            continue
        np.testing.assert_allclose(got[idx].numpy(), gsamp, rtol=2e-3, atol=2e-3 * gmax, err_msg=n)
        np.testing.assert_allclose(float(got.abs().sum()), gabs, rtol=5e-3, err_msg=n)
        # AdamW's first step moves a weight by lr * g / (|g| + 1e-8) (+ decay): compare where |g| is far above Adam's eps
        pw = g["p/" + n]
        after = q.detach().reshape(-1).double().cpu()
        firm = np.abs(gsamp) > 1e-5
        np.testing.assert_allclose(after[idx].numpy()[firm], pw[2:][firm], rtol=0, atol=1e-2 * lr, err_msg=n)
        np.testing.assert_allclose(float(after.abs().sum()), pw[1], rtol=1e-4, err_msg=n)
    if meta["kind"] == "bert":   # nn.Embedding(padding_idx): the [PAD] row took part in the forward and gets no gradient
        pad = prm["pad_token_id"]
        assert bool((xs == pad).any())
        assert float(seen["bert.embeddings.word_embeddings.weight"][pad].abs().max()) == 0.0


def _port_reference(c, prm, kind, duo, labels, masks, v0, vs, v1, exp, autocast_bf16=False):
    """loss and gradients of the CPU port under torch autograd; ``autocast_bf16``: the same under torch.autocast(bfloat16) — what the
    reference itself would compute in mixed precision: the yardstick of the bf16 training step."""
    import contextlib
    sd = {k_: v.detach().cpu().clone().requires_grad_(exp.state_dict(keep_vars=True)[k_].requires_grad)
          for k_, v in exp.state_dict(keep_vars=True).items()}
    ones = torch.ones((c["B"], c["P"]), dtype=torch.long)
    ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast_bf16 else contextlib.nullcontext()
    with ctx:
        phi_ref, z_ref = otp.explainer_phi(torch.from_numpy(c["xs"]), ones, v1, v0, sd, prm, kind)
        loss_ref = otp.shapley_loss(masks.reshape(c["B"], c["K"], c["P"]), v0, vs, phi_ref.float(), c["P"])
        if duo:
            lin = torch.nn.functional.linear
            if kind == "vit":
                base = torch.softmax(lin(z_ref[:, 0], sd["classifier.weight"], sd["classifier.bias"]).float(), -1)
            else:
                base = lin(torch.tanh(lin(z_ref[:, 0], sd["bert_pooler.dense.weight"], sd["bert_pooler.dense.bias"])),
                           sd["classifier.weight"], sd["classifier.bias"]).float()
            loss_ref = loss_ref + torch.nn.functional.cross_entropy(base, labels)
    loss_ref.backward()
    return sd, loss_ref.item()


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("tag", ["duo_vit_tiny_l3", "duo_bert_base_l2", "froyo_bert_base_l2"])
def test_reference_loops_gradients_match_autograd(cuda_device, tag):
    """the (duo) explainer loop through the boundary vs torch autograd on the CPU port, dropout off: loss and every
    trainable parameter's gradient.  The masks are the device sampler's; the port is fed the same ones."""
    from autognothi_amd import engine
    from autognothi_amd.models import shapley as amd_shapley
    from autognothi_amd.utils import synth
    c = build_case(tag)
    dev, recipe, kind, duo = cuda_device, c["recipe"], c["meta"]["kind"], c["meta"]["duo"]
    engine.set_precision("fp32")
    prm = dict(c["meta"]["params"], attention_probs_dropout_prob=0.0, hidden_dropout_prob=0.0)
    cfg = recipe.t_config(**prm)
    srg, exp = recipe.t_surrogate(cfg), recipe.t_explainer(cfg)
    synth.load_synth_weights(srg, seed=0)
    synth.load_synth_weights(exp, seed=1)
    srg, exp = srg.to(dev), exp.to(dev)
    xs = torch.from_numpy(c["xs"]).to(dev)
    labels = torch.tensor([1, 0][:c["B"]], dtype=torch.long)
    v0 = torch.from_numpy(c["g"]["v_0"])
    opt = torch.optim.SGD(exp.parameters(), lr=0.0)      # keep the weights: the gradients are what is compared
    seen = _capture_step(opt, exp)
    torch.manual_seed(11)
    body = _duo_explainer_epoch_train if duo else _explainer_epoch_train
    loss = body(_Env(), dev, c["K"], c["P"], v0.to(dev), [(None, None)], recipe, srg, exp, opt, 1,
                lambda a, b_: (xs, labels.to(dev)))
    torch.manual_seed(11)
    masks = amd_shapley.mask_shapley_new(c["B"] * c["K"], c["P"]).cpu()      # the same draw again
    with torch.no_grad():
        vs, _ = recipe.fw_surrogate(srg.eval(), xs, masks.to(dev))
        v1, _ = recipe.fw_surrogate(srg, xs, torch.ones((c["B"], c["P"]), dtype=torch.long, device=dev))
    sd, loss_ref = _port_reference(c, prm, kind, duo, labels, masks, v0, vs.cpu(), v1.cpu(), exp)
    np.testing.assert_allclose(loss * c["B"], loss_ref, rtol=3e-4)
    gscale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    checked = 0
    for name, p_ in exp.named_parameters():
        ref = sd[name].grad
        if not p_.requires_grad:
            assert seen[name] is None, name
            continue
        got = seen[name].cpu().numpy()
        if float(ref.abs().max()) < 1e-5 * gscale:
            assert float(np.abs(got).max()) < 1e-4 * gscale, name
            continue
        assert _rel(got, ref.numpy()) < 2e-3, (name, _rel(got, ref.numpy()))
        checked += 1
    assert checked >= 10


def test_reference_surrogate_loop(cuda_device):
    """scripts/train_surrogate.py's epoch body through the boundary: KL + gradients vs autograd on the CPU port."""
    from autognothi_amd import engine
    from autognothi_amd.models import shapley as amd_shapley
    from autognothi_amd.utils import synth
    c = build_case("vit_tiny_c1")
    dev, recipe = cuda_device, c["recipe"]
    engine.set_precision("fp32")
    prm = dict(c["meta"]["params"], attention_probs_dropout_prob=0.0, hidden_dropout_prob=0.0, num_hidden_layers=2)
    cfg = recipe.t_config(**prm)
    cls, srg = recipe.t_classifier(cfg), recipe.t_surrogate(cfg)
    synth.load_synth_weights(cls, seed=3)
    synth.load_synth_weights(srg, seed=0)
    cls, srg = cls.to(dev), srg.to(dev)
    xs = torch.from_numpy(c["xs"]).to(dev)
    opt = torch.optim.SGD(srg.parameters(), lr=0.0)
    seen = _capture_step(opt, srg)
    torch.manual_seed(5)
    kld = _surrogate_epoch_train(_Env(), dev, c["P"], [(None, None)], recipe, cls, srg, opt, 1,
                                 lambda a, b_: (xs, torch.zeros(c["B"], dtype=torch.long, device=dev)))
    torch.manual_seed(5)
    masks = amd_shapley.mask_purely_uniform(c["B"], c["P"]).cpu()
    with torch.no_grad():
        _, orig = recipe.fw_classifier(cls.eval(), xs, torch.ones((c["B"], c["P"]), dtype=torch.long, device=dev))
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in srg.state_dict().items()}
    mask_t = torch.cat([torch.ones((c["B"], 1), dtype=torch.long), masks], 1)
    z = otp.vit_backbone(torch.from_numpy(c["xs"]), mask_t, sd, prm)
    p_ref = torch.softmax(torch.nn.functional.linear(z[:, 0], sd["classifier.weight"], sd["classifier.bias"]), -1)
    l_ref = torch.nn.functional.kl_div(torch.log_softmax(orig.cpu(), -1), torch.softmax(p_ref, -1), reduction="batchmean")
    l_ref.backward()
    np.testing.assert_allclose(kld * c["B"], l_ref.item(), rtol=1e-3, atol=1e-7)
    gscale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for name, p_ in srg.named_parameters():
        r = sd[name].grad
        if float(r.abs().max()) < 1e-5 * gscale:
            assert float(seen[name].abs().max()) < 1e-4 * gscale, name
            continue
        assert _rel(seen[name].cpu().numpy(), r.numpy()) < 3e-3, name


def test_shapley_module_is_a_drop_in(cuda_device):
    """autognothi_amd.models.shapley vs reference models/shapley.py: masks bit-identical to the reference fixtures under the
    same torch.manual_seed, the global CPU generator left exactly where the reference leaves it, and the differentiable
    functions (loss, normalise, KL) agree with the reference formulas in value and gradient."""
    from autognothi_amd.models import shapley as S
    dev = cuda_device
    g = golden("masks_shapley.npz")
    for s, r, p in [(3407, 8, 196), (0, 32, 127), (3407, 4, 511)]:
        torch.manual_seed(s)
        m1 = S.mask_shapley_new(r, p)
        m2 = S.mask_shapley_new(r, p)
        after = torch.rand(700)                    # the host stream continues behind the device draws, across a twist
        assert m1.dtype == torch.int64 and m1.is_cuda and tuple(m1.shape) == (r, p)
        np.testing.assert_array_equal(m1.cpu().numpy(), unpack(g[f"s{s}_R{r}_P{p}_a"], p))
        np.testing.assert_array_equal(m2.cpu().numpy(), unpack(g[f"s{s}_R{r}_P{p}_b"], p))
        torch.manual_seed(s)
        torch.rand(2 * (r // 2 * p + r // 2))      # what two reference calls consume
        assert torch.equal(after, torch.rand(700))
    go = golden("masks_other.npz")
    torch.manual_seed(3407)
    np.testing.assert_array_equal(S.mask_purely_uniform(8, 196).cpu().numpy(), unpack(go["uniform_s3407_B8_P196"], 196))
    with pytest.raises(AssertionError):
        S.mask_shapley_new(3, 196)
    # differentiable functions vs the reference's formulas (fixtures of the reference's own outputs)
    f = golden("shapley_fns.npz")
    for tag in ("vit", "bert"):
        b, k, p, c = [int(x) for x in f[f"{tag}_dims"]]
        mask = torch.from_numpy(unpack(f[f"{tag}_loss_mask"], p)).to(dev)                  # [B, K, P]
        phi = torch.from_numpy(f[f"{tag}_loss_phi"]).to(dev).requires_grad_(True)
        loss = S.loss_shapley_new(b, k, p, mask, torch.from_numpy(f[f"{tag}_loss_v0"]).to(dev),
                                  torch.from_numpy(f[f"{tag}_loss_vs"]).to(dev), torch.from_numpy(f[f"{tag}_loss_v1"]).to(dev), phi)
        assert loss.dim() == 0
        (2.0 * loss).backward()
        np.testing.assert_allclose(loss.item(), f[f"{tag}_loss_out"][0], rtol=1e-5)
        np.testing.assert_allclose(phi.grad.cpu().numpy(), 2.0 * f[f"{tag}_loss_dphi"], rtol=1e-4, atol=6e-6)
        pred = torch.from_numpy(f[f"{tag}_norm_pred"]).to(dev).requires_grad_(True)
        out = S.normalize_shapley_explanation(pred, torch.from_numpy(f[f"{tag}_norm_grand"]).to(dev),
                                              torch.from_numpy(f[f"{tag}_norm_null"]).to(dev))
        np.testing.assert_allclose(out.detach().cpu().numpy(), f[f"{tag}_norm_out"], rtol=0, atol=1e-5)
        w = torch.from_numpy(np.random.default_rng(2).standard_normal(out.shape).astype(np.float32)).to(dev)
        (out * w).sum().backward()
        wn = w.cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(pred.grad.cpu().numpy(), wn - wn.sum(1, keepdims=True) / wn.shape[1], rtol=1e-5, atol=1e-6)
        ref, cur = torch.from_numpy(f[f"{tag}_kl_ref"]).to(dev), torch.from_numpy(f[f"{tag}_kl_cur"]).to(dev).requires_grad_(True)
        kl = S.loss_logits_kl_divergence(ref, cur)
        kl.backward()
        np.testing.assert_allclose(kl.item(), f[f"{tag}_kl_out"][0], rtol=1e-5, atol=1e-7)
        cur64 = torch.from_numpy(f[f"{tag}_kl_cur"]).double().requires_grad_(True)
        l64 = torch.nn.functional.kl_div(torch.log_softmax(torch.from_numpy(f[f"{tag}_kl_ref"]).double(), -1),
                                         torch.softmax(cur64, -1), reduction="batchmean")
        l64.backward()
        np.testing.assert_allclose(cur.grad.cpu().numpy(), cur64.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_gradient_accumulation_and_no_grad_paths(cuda_device):
    """autograd semantics of the bridge: two backward passes accumulate into .grad, zero_grad(set_to_none) resets, the
    same module under no_grad takes the inference path (no grad_fn), and eval()+grad still differentiates."""
    from autognothi_amd import engine
    c = build_case("froyo_vit_tiny_l3")
    dev, recipe = cuda_device, c["recipe"]
    engine.set_precision("fp32")
    exp = c["explainer"].to(dev)
    exp.train()
    xs = torch.from_numpy(c["xs"]).to(dev)
    ones = torch.ones((c["B"], c["P"]), dtype=torch.long, device=dev)
    v1, v0 = torch.from_numpy(c["g"]["v_1"]).to(dev), torch.from_numpy(c["g"]["v_0"]).to(dev)
    for q in exp.parameters():
        q.grad = None
    cfgd = exp.config
    assert cfgd.hidden_dropout_prob >= 0.0
    exp.eval()                                   # no dropout: two identical passes
    for q in exp.explainer_mlp.parameters():
        q.requires_grad_(True)
    phi, _ = recipe.fw_explainer(exp, xs, ones, v1, v0)
    assert phi.grad_fn is not None
    phi.sum().backward()
    g1 = exp.explainer_mlp[5].weight.grad.clone()
    phi2, _ = recipe.fw_explainer(exp, xs, ones, v1, v0)
    phi2.sum().backward()
    torch.testing.assert_close(exp.explainer_mlp[5].weight.grad, 2 * g1, rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        phi3, _ = recipe.fw_explainer(exp, xs, ones, v1, v0)
    assert phi3.grad_fn is None
    torch.testing.assert_close(phi3, phi.detach(), rtol=1e-4, atol=1e-5)     # training forward == inference forward (fp32)
