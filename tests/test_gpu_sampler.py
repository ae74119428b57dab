"""GPU parity: device MT19937 + mask samplers vs the reference-generated fixtures and the oracle (bit-exact)."""
import zlib

import numpy as np
import pytest
import torch

from oracle import shapley as osh
from oracle.mt19937 import MT19937
from util import bits_to_mask, golden, unpack

pytestmark = pytest.mark.gpu


def test_mt19937_raw_stream(cuda_device):
    from autognothi_amd import ops
    for seed in (0, 5, 3407, 2 ** 32 - 1):
        rng = ops.DeviceMT19937(cuda_device, seed)
        want = MT19937(seed)
        for n in (1, 7, 623, 624, 625, 5000):  # crosses twists, continues the stream
            got = rng.raw(n).cpu().numpy().view(np.uint32)
            assert np.array_equal(got, want.raw(n)), (seed, n)


def test_mask_shapley_new_fixtures(cuda_device):
    from autognothi_amd import ops
    g, t = golden("masks_shapley.npz"), golden("prefix_tables.npz")
    for s, r, p in g["cases"]:
        rng = ops.DeviceMT19937(cuda_device, int(s))
        prefix = torch.from_numpy(t[f"prefix_{p}"]).to(cuda_device)
        for tag in ("a", "b"):
            mi, mb = ops.mask_shapley_new(rng, int(r), int(p), prefix=prefix)
            want = unpack(g[f"s{s}_R{r}_P{p}_{tag}"], int(p))
            assert np.array_equal(mi.cpu().numpy(), want), (s, r, p, tag)
            cls, from_bits = bits_to_mask(mb.cpu().numpy().view(np.uint32), int(p))
            assert np.array_equal(from_bits, want) and cls.min() == 1
            # bits beyond T are zero
            tw = mb.shape[1]
            full = mb.cpu().numpy().view(np.uint32)
            extra = (tw * 32) - (int(p) + 1)
            if extra:
                assert (full[:, -1] >> np.uint32(32 - extra)).max() == 0


def test_prefix_table_matches_fixture(cuda_device):
    from autognothi_amd import ops
    t = golden("prefix_tables.npz")
    for p in (196, 127, 511):
        assert np.array_equal(ops.shapley_prefix_table(p, cuda_device).cpu().numpy(), t[f"prefix_{p}"])


def test_mask_shapley_new_many_twists_and_full_size(cuda_device):
    from autognothi_amd import ops
    g = golden("masks_shapley.npz")
    mi, _ = ops.mask_shapley_new(ops.DeviceMT19937(cuda_device, 99), 1024, 196)
    m = mi.cpu().numpy()
    assert np.array_equal(m.sum(1), g["big_s99_R1024_P196_rowsum"])
    assert np.array_equal(m.sum(0), g["big_s99_R1024_P196_colsum"])
    assert zlib.crc32(np.packbits(m.astype(np.uint8), axis=1).tobytes()) == int(g["big_s99_R1024_P196_crc"][0])
    # BASELINE-size property check (B=64 x K=32): every pair is complementary
    mi, mb = ops.mask_shapley_new(ops.DeviceMT19937(cuda_device, 1), 2048, 196)
    assert bool(((mi[0::2] + mi[1::2]) == 1).all())
    t = golden("prefix_tables.npz")
    assert np.array_equal(mi.cpu().numpy(), osh.mask_shapley_new(2048, 196, MT19937(1), t["prefix_196"]))


def test_odd_sample_count_rejected(cuda_device):
    from autognothi_amd import ops
    with pytest.raises(AssertionError):
        ops.mask_shapley_new(ops.DeviceMT19937(cuda_device, 0), 3, 196)


def test_mask_purely_uniform_fixtures(cuda_device):
    from autognothi_amd import ops
    g = golden("masks_other.npz")
    for s, b, p in g["cases"]:
        mi, mb = ops.mask_purely_uniform(ops.DeviceMT19937(cuda_device, int(s)), int(b), int(p))
        want = unpack(g[f"uniform_s{s}_B{b}_P{p}"], int(p))
        assert np.array_equal(mi.cpu().numpy(), want)
        assert np.array_equal(bits_to_mask(mb.cpu().numpy().view(np.uint32), int(p))[1], want)


def test_torch_cpu_generator_roundtrip(cuda_device):
    """The device stream continues torch's CPU generator and can hand it back (drop-in semantics of the
    reference, which draws masks from the global CPU generator)."""
    from autognothi_amd import ops
    torch.manual_seed(3407)
    _ = torch.rand(10)  # advance the host generator a bit
    rng = ops.DeviceMT19937(cuda_device).import_torch_cpu_state()
    want1 = (torch.rand(8, 196) > 0.5)  # what the host would have drawn next
    got = rng.raw(8 * 196).cpu().numpy().view(np.uint32)
    got1 = ((got & 0xFFFFFF).astype(np.float32) * np.float32(2 ** -24)).reshape(8, 196) > 0.5
    assert np.array_equal(got1, want1.numpy())
    torch.manual_seed(3407)
    _ = torch.rand(10)
    rng = ops.DeviceMT19937(cuda_device).import_torch_cpu_state()
    rng.raw(1000)
    rng.export_to_torch_cpu()
    after_device = torch.rand(1500)      # crosses two 624-word block boundaries of the handed-back generator
    torch.manual_seed(3407)
    _ = torch.rand(10 + 1000)
    assert torch.equal(after_device, torch.rand(1500))
    # hand-back exactly at a block boundary (pos == 624: the host twists on its next draw)
    torch.manual_seed(7)
    rng = ops.DeviceMT19937(cuda_device).import_torch_cpu_state()
    rng.raw(624)
    rng.export_to_torch_cpu()
    after_device = torch.rand(700)
    torch.manual_seed(7)
    _ = torch.rand(624)
    assert torch.equal(after_device, torch.rand(700))


def test_pack_mask(cuda_device):
    from autognothi_amd import ops
    g = np.random.default_rng(0)
    for r, p in [(1, 196), (7, 127), (33, 511), (5, 31), (5, 32)]:
        m = g.integers(0, 2, size=(r, p), dtype=np.int64)
        bits = ops.pack_mask(torch.from_numpy(m).to(cuda_device)).cpu().numpy().view(np.uint32)
        cls, back = bits_to_mask(bits, p)
        assert np.array_equal(back, m) and cls.min() == 1


def test_perturbed_masks(cuda_device):
    from autognothi_amd import ops
    g = golden("perturbed.npz")
    for i, (p, steps) in enumerate(g["cases"]):
        for base in (0, 1):
            attr = torch.from_numpy(g[f"c{i}_attr"]).to(cuda_device)[None]
            stops, masks = ops.perturbed_masks(attr, int(steps), base)
            assert np.array_equal(stops.cpu().numpy(), g[f"c{i}_b{base}_stops"])
            assert np.array_equal(masks[0].cpu().numpy(), unpack(g[f"c{i}_b{base}_masks"], int(p)))
    # ties: the flipped set must still be a valid top-i set
    attr = np.round(np.random.default_rng(3).standard_normal((4, 196)), 1).astype(np.float32)
    stops, masks = ops.perturbed_masks(torch.from_numpy(attr).to(cuda_device), 16, 0)
    masks, stops = masks.cpu().numpy(), stops.cpu().numpy()
    for a in range(4):
        for s, st in enumerate(stops):
            on = masks[a, s] == 1
            assert on.sum() == st
            if 0 < st < 196:
                assert attr[a][on].min() >= attr[a][~on].max()


def test_perturbed_masks_with_ties(cuda_device):
    """attributions WITH ties through the reference's _get_perturbed_samples (fixture made by the reference
    on the build host).  np.argsort's default kind is not stable and its tie order depends on the host's numpy build (the
    fixture records that it differs from the stable order), so bit-exactness is only defined where the top-k set is unique:
    there the device masks equal the reference's; inside a tie group the device takes the higher index first and the
    test checks that both choices are valid top-k sets."""
    from autognothi_amd import ops
    from util import check_perturbed_against_reference
    g = golden("perturbed_ties.npz")
    split = 0
    for i, (p, steps) in enumerate(g["cases"]):
        attr = g[f"c{i}_attr"]
        for base in (0, 1):
            stops, masks = ops.perturbed_masks(torch.from_numpy(attr).to(cuda_device)[None], int(steps), base)
            split += check_perturbed_against_reference(attr, base, stops.cpu().numpy(), masks[0].cpu().numpy(),
                                                       g[f"c{i}_b{base}_stops"], unpack(g[f"c{i}_b{base}_masks"], int(p)))
    assert split > 20          # the fixture does exercise tie-splitting cuts
    # the documented device rule: inside a tie group the higher index ranks first (stable ascending argsort, reversed)
    attr = g["c0_attr"]
    stops, masks = ops.perturbed_masks(torch.from_numpy(attr).to(cuda_device)[None], 196, 0)
    rank_dev = np.argsort(attr, kind="stable")[::-1]
    for s, st in enumerate(stops.cpu().numpy()):
        want = np.zeros(196, dtype=np.int64)
        want[rank_dev[:st]] = 1
        assert np.array_equal(masks[0, s].cpu().numpy(), want)


def test_sharded_mask_stream_is_the_unsharded_stream(cuda_device):
    """row-sharded ranks (distributed.ShardedMaskStream / ag_mask_shapley_new_rows): each rank's rows of the global call are
    bit-identical to the single call's rows, for every rank count, and every rank's generator ends where the single
    generator ends (the next step stays in lockstep).  Reference: one mask_shapley_new(B*K, P) per step, models/shapley.py:56-79."""
    from autognothi_amd import ops
    from autognothi_amd.distributed import ShardedMaskStream
    dev = cuda_device
    n_inputs, k, p = 8, 6, 196
    rng = ops.DeviceMT19937(dev, 3407)
    want1, wbits1 = ops.mask_shapley_new(rng, n_inputs * k, p)
    want2, _ = ops.mask_shapley_new(rng, n_inputs * k, p)
    tail = rng.raw(5).cpu().numpy()
    for world in (1, 2, 4, 8):
        per = n_inputs // world
        got1, got2, bits1, tails = [], [], [], []
        for r in range(world):
            st = ShardedMaskStream(dev, 3407)
            m1, b1 = st.sample(n_inputs, r * per, (r + 1) * per, k, p, want_i64=True)
            m2, _ = st.sample(n_inputs, r * per, (r + 1) * per, k, p, want_i64=True)
            got1.append(m1); got2.append(m2); bits1.append(b1)
            tails.append(st.rng.raw(5).cpu().numpy())
        assert torch.equal(torch.cat(got1), want1) and torch.equal(torch.cat(got2), want2), world
        assert torch.equal(torch.cat(bits1), wbits1)
        assert all(np.array_equal(t, tail) for t in tails)
    # skip == drawing and discarding, across twists
    a, b = ops.DeviceMT19937(dev, 11), ops.DeviceMT19937(dev, 11)
    for n in (1, 623, 624, 625, 5000):
        a.raw(n)
        b.skip(n)
        assert np.array_equal(a.raw(3).cpu().numpy(), b.raw(3).cpu().numpy()), n
