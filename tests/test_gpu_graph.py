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
