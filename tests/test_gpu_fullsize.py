"""BASELINE.json full-size configurations (ViT-base K=32 bf16; BERT-base L=128 K=32), checked through properties that do not
need an oracle run at that size: the hot path must be invariant to how the (input x mask) rows are sharded, batched and
ordered, an all-visible mask must reproduce the unmasked forward, and the Shapley normalisation must be efficient.

Run-to-run the outputs are bit-identical (the LayerNorm-fold row statistics are per-tile partial sums added in a fixed
order: no float atomics; test_bf16_forward_is_bit_reproducible).  Across different batchings the comparisons keep the bf16
model tolerance: a row can meet a differently shaped GEMM (ring kernel vs 128-tile kernel below 1024 rows), which may move a
bf16 activation by one rounding step; a row mix-up shows up at >= 5e-2 (asserted below)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

K = 32
TOL = 2e-2   # bf16 mode, 12 layers, random-init weights: the tolerance of the bf16-vs-reference model tests (test_gpu_models.py)


def _setup(workload, dev, batch, seed=0):
    import bench
    from autognothi_amd import engine
    from autognothi_amd.recipes import get_recipe
    from autognothi_amd.utils import synth
    kind, params, _ = bench.WORKLOADS[workload]
    recipe = get_recipe(kind)
    cfg = recipe.t_config(**params)
    engine.set_precision("bf16")
    m = recipe.t_surrogate(cfg)
    synth.load_synth_weights(m, seed=0)
    m = m.to(dev).eval()
    if kind.endswith("vit"):
        xs = synth.synth_images(batch, params["img_px_size"], params["img_channels"], seed=seed)
    else:
        xs = synth.synth_token_ids(batch, params["max_position_embeddings"], params["vocab_size"], seed=seed)
    return recipe, cfg, m, torch.from_numpy(xs).to(dev)


def _probs(recipe, model, xs, mask):
    with torch.no_grad():
        v, _ = recipe.fw_surrogate(model, xs, mask)
    return v.float().cpu().numpy()


@pytest.mark.parametrize("workload", ["vit_base", "bert_base"])
def test_sharding_and_order_invariance(cuda_device, workload):
    """rows shard by input across ranks: forwarding inputs [0:2] and [2:4] separately, or the inputs in reverse
    order, gives the joint result; so does a permutation of the K masks of an input (the outputs permute with them)."""
    from autognothi_amd import ops
    recipe, cfg, m, xs = _setup(workload, cuda_device, 4)
    P = recipe.n_players(cfg)
    rng = ops.DeviceMT19937(cuda_device, 11)
    masks, _ = ops.mask_shapley_new(rng, 4 * K, P, want_i64=True, want_bits=False)
    joint = _probs(recipe, m, xs, masks)
    assert joint.shape == (4 * K, cfg.num_labels) and np.isfinite(joint).all()
    np.testing.assert_allclose(joint.sum(-1), 1.0, atol=1e-3)
    halves = np.concatenate([_probs(recipe, m, xs[:2], masks[:2 * K]), _probs(recipe, m, xs[2:], masks[2 * K:])])
    np.testing.assert_allclose(halves, joint, rtol=0, atol=TOL)
    rev = _probs(recipe, m, xs.flip(0), masks.view(4, K, P).flip(0).reshape(4 * K, P))
    np.testing.assert_allclose(rev.reshape(4, K, -1)[::-1].reshape(4 * K, -1), joint, rtol=0, atol=TOL)
    perm = torch.randperm(K, generator=torch.Generator().manual_seed(5)).to(cuda_device)
    shuffled = _probs(recipe, m, xs, masks.view(4, K, P)[:, perm].reshape(4 * K, P))
    np.testing.assert_allclose(shuffled, joint.reshape(4, K, -1)[:, perm.cpu().numpy()].reshape(4 * K, -1), rtol=0, atol=TOL)
    # masks matter: the K rows of an input are not all alike
    assert float(np.abs(joint.reshape(4, K, -1) - joint.reshape(4, K, -1)[:, :1]).max()) > 5e-2
    assert float(np.abs(joint.reshape(4, K, -1)[0] - joint.reshape(4, K, -1)[1]).max()) > 5e-2   # and so do inputs


@pytest.mark.parametrize("workload,batch", [("vit_base", 48), ("bert_base", 48), ("vit_large", 24)])
def test_benchmarked_step_equals_its_small_batches(cuda_device, workload, batch):
    """the step bench.py times — 48 inputs x K = 32 (ViT-base: M = 302 592 token rows through the persistent 256^2 kernel, its tile groups,
    the residual-through-LDS epilogue, the stream attention, the K/V-free last layer; BERT-base with token pruning) and 24 inputs x K = 64
    of ViT-large (BASELINE config 4's K; QKV in two tile groups) — against the same rows forwarded four inputs at a time, where every
    Linear meets other routes (under-filled launches, the planner's 128^2 units): the rows must agree within the bf16 model tolerance,
    and the big step must replay bit for bit."""
    import bench
    from autognothi_amd import ops
    k = bench.WORKLOADS[workload][2]
    recipe, cfg, m, xs = _setup(workload, cuda_device, batch, seed=7)
    P = recipe.n_players(cfg)
    masks, _ = ops.mask_shapley_new(ops.DeviceMT19937(cuda_device, 29), batch * k, P, want_i64=True, want_bits=False)
    joint = _probs(recipe, m, xs, masks)
    assert joint.shape == (batch * k, cfg.num_labels) and np.isfinite(joint).all()
    np.testing.assert_allclose(joint.sum(-1), 1.0, atol=1e-3)
    np.testing.assert_array_equal(_probs(recipe, m, xs, masks), joint)
    parts = np.concatenate([_probs(recipe, m, xs[i:i + 4], masks[i * k:(i + 4) * k]) for i in range(0, batch, 4)])
    np.testing.assert_allclose(parts, joint, rtol=0, atol=TOL)
    per_input = joint.reshape(batch, k, -1)
    assert float(np.abs(per_input - per_input[:, :1]).max()) > 5e-2        # masks matter
    assert float(np.abs(per_input[0] - per_input[1]).max()) > 5e-2         # and so do inputs


@pytest.mark.parametrize("workload", ["vit_base", "bert_base"])
def test_all_visible_equals_single_unmasked_forward(cuda_device, workload):
    """K all-ones masks (layer-0 sharing, and for BERT the token-pruned path with nothing pruned) == the K=1 forward."""
    recipe, cfg, m, xs = _setup(workload, cuda_device, 3, seed=1)
    P = recipe.n_players(cfg)
    ones = torch.ones((3 * K, P), dtype=torch.int64, device=cuda_device)
    many = _probs(recipe, m, xs, ones).reshape(3, K, -1)
    single = _probs(recipe, m, xs, ones[:3])
    np.testing.assert_allclose(many, np.repeat(single[:, None], K, axis=1), rtol=0, atol=TOL)


def test_bert_pruned_equals_unpruned_full_size(cuda_device):
    """token pruning against the same kernels without it, at BERT-base L=128 K=32."""
    from autognothi_amd import engine, ops
    recipe, cfg, m, xs = _setup("bert_base", cuda_device, 4, seed=2)
    P = recipe.n_players(cfg)
    rng = ops.DeviceMT19937(cuda_device, 23)
    masks, _ = ops.mask_shapley_new(rng, 4 * K, P, want_i64=True, want_bits=False)
    keep = engine.PRUNE_BERT_TOKENS
    try:
        engine.PRUNE_BERT_TOKENS = True
        a = _probs(recipe, m, xs, masks)
        engine.PRUNE_BERT_TOKENS = False
        b = _probs(recipe, m, xs, masks)
    finally:
        engine.PRUNE_BERT_TOKENS = keep
    np.testing.assert_allclose(a, b, rtol=0, atol=TOL)


def test_shapley_normalize_full_size_and_efficiency(cuda_device):
    """ag_shapley_normalize at B=64 inputs x T=197 tokens x C=10 classes against the oracle (numpy, milliseconds at this
    size), and the reference's efficiency property WITH its quirk (models/shapley.py:82-93 divides by T = P+1 and the CLS
    row is dropped afterwards): sum_p phi + (adjusted CLS row) == grand - null."""
    from autognothi_amd import ops
    from oracle import shapley as osh
    g = np.random.default_rng(17)
    pred = g.standard_normal((64, 197, 10)).astype(np.float32) * 0.05
    grand = g.random((64, 10)).astype(np.float32)
    null = np.full((1, 10), 0.1, dtype=np.float32)
    phi = ops.shapley_normalize(torch.from_numpy(pred).to(cuda_device), torch.from_numpy(grand).to(cuda_device),
                                torch.from_numpy(null).to(cuda_device)).cpu().numpy()          # [B, C, P]
    ref = osh.normalize_shapley_explanation(pred, grand, null)                                 # [B, T, C]
    np.testing.assert_allclose(phi, ref[:, 1:].transpose(0, 2, 1), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(phi.sum(-1) + ref[:, 0], grand - null, rtol=0, atol=2e-5)


def test_chained_per_layer_forward_equals_one_call(cuda_device):
    """ag_encoder_forward_chained (one call per layer, LayerNorm-fold row statistics handed from call to call: the LTT
    backbone) against ag_encoder_forward over all layers, on the ViT-base encoder with layer-0 sharing."""
    from autognothi_amd import _lib as L, engine, ops
    recipe, cfg, m, xs = _setup("vit_base", cuda_device, 2, seed=5)
    P = recipe.n_players(cfg)
    T = P + 1
    rng = ops.DeviceMT19937(cuda_device, 31)
    _, bits = ops.mask_shapley_new(rng, 2 * K, P, want_i64=False, want_bits=True)
    dtype = engine.get_precision()
    vit = m.vit
    vit.packed()                      # (builds the patch-embedding pack that embed() uses)
    layers = list(vit.encoder.layers)
    c = cfg
    whole = engine.PackedEncoder(layers, L.AG_MASK_VIT_MUL, T, c.hidden_size, c.intermediate_size, c.num_attention_heads, c.layer_norm_eps)
    singles = [engine.PackedEncoder([ly], L.AG_MASK_VIT_MUL, T, c.hidden_size, c.intermediate_size, c.num_attention_heads, c.layer_norm_eps)
               for ly in layers]
    h0 = vit.embed(xs, dtype)
    rows = 2 * K
    ref = whole.forward(h0, rows, K, bits, False, dtype).float().cpu().numpy()
    chain = [ops.new_row_stats(rows * T, 768, cuda_device), False, True]   # [ceil(H/256), rows*T, 2] slab partials
    h, used = h0, []
    for i, enc in enumerate(singles):
        chain[2] = i + 1 < len(singles)
        used.append(chain[1])
        h = enc.forward(h, rows, K if i == 0 else 1, bits, False, dtype, chain=chain)
    if engine.FOLD_LAYERNORM:             # (AG_LN_FOLD=0 runs the LayerNorm kernels instead: nothing to hand on)
        assert used[0] is False and all(used[1:]), used      # every layer after the first consumed its predecessor's statistics
    got = h.float().cpu().numpy()
    np.testing.assert_array_equal(got, ref)       # same kernels, same statistics partials, same order: bit for bit


@pytest.mark.parametrize("workload", ["vit_base", "bert_base"])
def test_bf16_forward_is_bit_reproducible(cuda_device, workload):
    """the reference reseeds every epoch so that runs replay (utils/tools.py:46-54); the throughput mode must not give that
    up: the same inputs and masks give the same bits, launch after launch (LayerNorm folding without float atomics)."""
    from autognothi_amd import ops
    recipe, cfg, m, xs = _setup(workload, cuda_device, 8, seed=3)
    P = recipe.n_players(cfg)
    masks, _ = ops.mask_shapley_new(ops.DeviceMT19937(cuda_device, 5), 8 * K, P, want_i64=True, want_bits=False)
    first = _probs(recipe, m, xs, masks)
    for _ in range(3):
        np.testing.assert_array_equal(_probs(recipe, m, xs, masks), first)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_bert_full_sequence_length_512(cuda_device, precision):
    """the reference's BERT experiments run at max_position_embeddings = 512 (experiments/bert_base_tayp_*/.hparams.json; BASELINE
    config 3 shortens it to 128): the same path at T = 512, P = 511 — packed (token-pruned, LayerNorm-free in bf16) against
    un-pruned, masks from the device sampler at P = 511, an all-visible mask against the single unmasked forward."""
    import bench
    from autognothi_amd import engine, ops
    from autognothi_amd.recipes import get_recipe
    from autognothi_amd.utils import synth
    kind, params, _ = bench.WORKLOADS["bert_base"]
    params = dict(params, max_position_embeddings=512, num_hidden_layers=4)
    recipe = get_recipe(kind)
    cfg = recipe.t_config(**params)
    engine.set_precision(precision)
    try:
        m = recipe.t_surrogate(cfg)
        synth.load_synth_weights(m, seed=0)
        m = m.to(cuda_device).eval()
        b, k, p = 3, 8, recipe.n_players(cfg)
        assert p == 511
        xs = torch.from_numpy(synth.synth_token_ids(b, 512, params["vocab_size"], seed=4)).to(cuda_device)
        masks, _ = ops.mask_shapley_new(ops.DeviceMT19937(cuda_device, 77), b * k, p, want_i64=True, want_bits=False)
        keep = engine.PRUNE_BERT_TOKENS
        try:
            engine.PRUNE_BERT_TOKENS = True
            a = _probs(recipe, m, xs, masks)
            engine.PRUNE_BERT_TOKENS = False
            u = _probs(recipe, m, xs, masks)
        finally:
            engine.PRUNE_BERT_TOKENS = keep
        tol = 2e-5 if precision == "fp32" else TOL
        np.testing.assert_allclose(a, u, rtol=0, atol=tol)
        np.testing.assert_allclose(a.sum(-1), 1.0, atol=1e-3)
        assert float(np.abs(a.reshape(b, k, -1) - a.reshape(b, k, -1)[:, :1]).max()) > 1e-3      # masks matter
        ones = torch.ones((b * k, p), dtype=torch.int64, device=cuda_device)
        many = _probs(recipe, m, xs, ones).reshape(b, k, -1)
        single = _probs(recipe, m, xs, ones[:b])
        np.testing.assert_allclose(many, np.repeat(single[:, None], k, axis=1), rtol=0, atol=tol)
    finally:
        engine.set_precision("bf16")
