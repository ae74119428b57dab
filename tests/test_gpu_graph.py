"""hipGraph capture of the hot-path step (engine.GraphedStep): a replay is the eager step, bit for bit, including the
device-resident mask generator's progress (reference loop: scripts/train_explainer.py:149-171)."""
import numpy as np
import pytest
import torch

from util import build_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,precision", [("vit_tiny_c1", "fp32"), ("vit_tiny_c1", "bf16"), ("vit_base_l12", "bf16"), ("bert_base_l12", "bf16"),
                                           ("ltt_bert_base_l2", "bf16"), ("ltt_vit_tiny_l3", "bf16")])
def test_graphed_step_equals_eager(cuda_device, tag, precision):
    from autognothi_amd import engine, ops
    c = build_case(tag)
    dev, recipe = cuda_device, c["recipe"]
    engine.set_precision(precision)
    srg = c["surrogate"].to(dev).eval()
    xs = torch.from_numpy(c["xs"]).to(dev)
    rows, p = c["B"] * c["K"], c["P"]
    rng = ops.DeviceMT19937(dev, 3407)

    def step():
        _, bits = ops.mask_shapley_new(rng, rows, p, want_i64=False, want_bits=True)
        with torch.no_grad():
            v_s, _ = recipe.fw_surrogate(srg, xs, bits)
        return v_s, bits

    g = engine.GraphedStep(step)
    rng.seed(3407)                       # (the warm-up calls drew from the generator)
    got = []
    for _ in range(3):
        v, b = g()
        got.append((v.clone(), b.clone()))
    rng.seed(3407)
    for i in range(3):
        v, b = step()
        assert torch.equal(b, got[i][1]), f"replay {i}: masks differ from the eager stream"
        assert torch.equal(v, got[i][0]), f"replay {i}: outputs differ from the eager step"
    # and the first replay is the reference's first batch of masks for this seed
    np.testing.assert_array_equal(ops.pack_mask(torch.from_numpy(c["masks"]).to(dev)).cpu().numpy(), got[0][1].cpu().numpy())


def _train_case(tag, dev, p_drop):
    from autognothi_amd import ops
    from autognothi_amd.utils import synth
    c = build_case(tag)
    recipe = c["recipe"]
    prm = dict(c["meta"]["params"])
    prm["hidden_dropout_prob"], prm["attention_probs_dropout_prob"] = p_drop, p_drop
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


@pytest.mark.parametrize("tag", ["vit_tiny_c1", "duo_bert_base_l2", "froyo_vit_tiny_l3"])
def test_graphed_training_step_equals_eager(cuda_device, tag):
    """the explainer training step (weight refresh + forward + Shapley loss + backward on both streams) replayed from a hipGraph:
    loss and every gradient bit-identical to the eager step, across optimiser steps in between (the graph re-reads the updated
    fp32 parameters), dropout off (reference loop body: scripts/train_explainer.py:182-196)."""
    from autognothi_amd import engine, training, training16
    dev = cuda_device
    c, recipe, exp, xs, bits, v0, vs, v1, labels = _train_case(tag, dev, 0.0)
    training.MIXED_BF16 = True
    try:
        params = [p for p in exp.parameters() if p.requires_grad]
        start = [p.detach().clone() for p in params]

        def run(use_graph):
            with torch.no_grad():
                for p, s0 in zip(params, start):
                    p.copy_(s0)
            engine.invalidate_weight_caches()
            tr = training.ExplainerTrainer(recipe, exp)
            assert isinstance(tr, training16.ExplainerTrainer16)
            tr.use_graph = use_graph
            opt = torch.optim.SGD(params, lr=1e-7)      # (small: four steps on synthetic weights must stay finite)
            engine.watch_optimizer(opt)
            out = []
            for step in range(4):
                opt.zero_grad()
                loss, _ = tr.loss_and_grads(xs, bits, v0, vs, v1, c["K"], labels=labels, train=True, seed=2)
                out.append((float(loss), [p.grad.clone() for p in params]))
                assert all(bool(torch.isfinite(g_).all()) for g_ in out[-1][1])
                opt.step()
            if use_graph:
                assert len(tr._graphs) == 1, "the step was not captured"
            return out

        names = [n for n, p in exp.named_parameters() if p.requires_grad]
        eager, eager2, graphed = run(False), run(False), run(True)
        for other, what in ((eager2, "a second eager run"), (graphed, "the replayed step")):
            for i, ((le, ge), (lg, gg)) in enumerate(zip(eager, other)):
                assert le == lg, f"step {i}: loss {le} vs {lg} ({what})"
                bad = [(n, float((a - b).abs().max())) for n, a, b in zip(names, ge, gg) if not torch.equal(a, b)]
                assert not bad, f"step {i}: gradients differ between the eager step and {what}: {bad[:8]} ({len(bad)} of {len(names)})"
    finally:
        training.MIXED_BF16 = False


def test_graphed_training_step_with_dropout(cuda_device):
    """dropout on.  (1) A replay under salt 0 IS the captured step: loss and gradients bit-identical to the eager step with the
    same (seed, step) keys — forward and backward of the graph share their keep patterns.  (2) Replays under the per-step salt
    (ag_set_dropout_salt) draw different patterns from ONE captured graph.  (3) Gradients stay finite over optimiser steps."""
    from autognothi_amd import engine, training
    dev = cuda_device
    c, recipe, exp, xs, bits, v0, vs, v1, labels = _train_case("bert_base_l2", dev, 0.1)
    training.MIXED_BF16 = True
    try:
        tr = training.ExplainerTrainer(recipe, exp)
        params = [p for p in exp.parameters() if p.requires_grad]
        names = [n for n, p in exp.named_parameters() if p.requires_grad]

        def call(graph, step_before):
            for p in params:
                p.grad = None
            tr.use_graph, tr.step = graph, step_before
            loss, _ = tr.loss_and_grads(xs, bits, v0, vs, v1, c["K"], train=True, seed=9)
            return float(loss), [p.grad.clone() for p in params]

        call(True, 10)                       # first call with these shapes: eager
        tr.graph_salt = 0
        l_cap, g_cap = call(True, 20)        # captured at step 21 and replayed under salt 0
        assert len(tr._graphs) == 1
        l_rep, g_rep = call(True, 33)        # another replay, salt 0: the captured step again, whatever the step counter says
        l_eag, g_eag = call(False, 20)       # the eager step with the captured keys
        assert l_cap == l_rep == l_eag
        for n, a, b, e in zip(names, g_cap, g_rep, g_eag):
            if n.endswith("word_embeddings.weight"):
                continue                     # torch's index_add_ (float atomics)
            assert torch.equal(a, b) and torch.equal(a, e), n
        tr.graph_salt = None
        losses = [call(True, 40 + i)[0] for i in range(3)]
        assert len(set(losses + [l_cap])) == 4, losses       # fresh keep patterns at every replay
        opt = torch.optim.AdamW(params, lr=1e-5, fused=True)
        engine.watch_optimizer(opt)
        tr.use_graph = True
        for _ in range(4):
            opt.zero_grad()
            loss, _ = tr.loss_and_grads(xs, bits, v0, vs, v1, c["K"], train=True, seed=9)
            assert np.isfinite(float(loss)) and all(bool(torch.isfinite(p.grad).all()) for p in params)
            opt.step()
        assert len(tr._graphs) == 1
    finally:
        training.MIXED_BF16 = False
