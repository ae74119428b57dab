"""Pin the oracle's samplers / seeding / faithfulness masks against the reference-generated fixtures (CPU)."""
import zlib

import numpy as np

from oracle import shapley as osh
from oracle.mt19937 import MT19937
from util import golden, golden_json, unpack


def test_iterative_seeds():
    g = golden_json("iterative_seeds.json")
    for key, val in g["derived"].items():
        assert osh.iterative_seed(g["master"], key) == val


def test_prefix_table_restatement_close():
    t = golden("prefix_tables.npz")
    for p in (196, 127, 511):
        np.testing.assert_allclose(osh.shapley_prefix_table(p), t[f"prefix_{p}"], rtol=0, atol=3e-7)


def test_mask_shapley_new_bit_exact():
    g, t = golden("masks_shapley.npz"), golden("prefix_tables.npz")
    for s, r, p in g["cases"]:
        gen = MT19937(int(s))
        for tag in ("a", "b"):  # second call continues the stream
            got = osh.mask_shapley_new(int(r), int(p), gen, t[f"prefix_{p}"])
            want = unpack(g[f"s{s}_R{r}_P{p}_{tag}"], int(p))
            assert np.array_equal(got, want), (s, r, p, tag)


def test_mask_shapley_new_many_twists():
    g, t = golden("masks_shapley.npz"), golden("prefix_tables.npz")
    m = osh.mask_shapley_new(1024, 196, MT19937(99), t["prefix_196"])
    assert np.array_equal(m.sum(1), g["big_s99_R1024_P196_rowsum"])
    assert np.array_equal(m.sum(0), g["big_s99_R1024_P196_colsum"])
    assert zlib.crc32(np.packbits(m.astype(np.uint8), axis=1).tobytes()) == int(g["big_s99_R1024_P196_crc"][0])


def test_pairs_are_complements_and_edge_rows():
    t = golden("prefix_tables.npz")
    m = osh.mask_shapley_new(64, 196, MT19937(3407), t["prefix_196"])
    assert np.array_equal(m[0::2] + m[1::2], np.ones((32, 196), dtype=np.int64))
    try:
        osh.mask_shapley_new(3, 196, MT19937(0), t["prefix_196"])
        raise SystemExit("odd n must assert")
    except AssertionError:
        pass


def test_other_samplers():
    g = golden("masks_other.npz")
    for s, b, p in g["cases"]:
        got = osh.mask_purely_uniform(int(b), int(p), MT19937(int(s)))
        assert np.array_equal(got, unpack(g[f"uniform_s{s}_B{b}_P{p}"], int(p)))
        got = osh.mask_uniform_selective(int(b), int(p), int(p) // 3, seed=int(s))
        assert np.array_equal(got, unpack(g[f"selective_s{s}_B{b}_P{p}"], int(p)))
        assert (got == 0).sum(1).tolist() == [int(p) // 3] * int(b)


def test_perturbed_samples():
    g = golden("perturbed.npz")
    for i, (p, steps) in enumerate(g["cases"]):
        for base in (0, 1):
            stops, masks = osh.get_perturbed_samples(g[f"c{i}_attr"], int(p), int(steps), base)
            assert np.array_equal(stops, g[f"c{i}_b{base}_stops"])
            assert np.array_equal(masks, unpack(g[f"c{i}_b{base}_masks"], int(p)))
    assert abs(osh.auc(g["auc_vals"]) - float(g["auc_out"][0])) < 1e-12
