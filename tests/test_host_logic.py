"""CPU: host-side logic that needs no GPU — state-dict converters, seeding, recipe registry, the state-dict
layout contract and FLOP accounting."""
import os

import pytest
import torch

from util import golden_json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_iterative_seed_matches_reference_fixture():
    from autognothi_amd.utils.tools import derive_seed, set_iterative_seed
    g = golden_json("iterative_seeds.json")
    for key, val in g["derived"].items():
        assert derive_seed(g["master"], key) == val
    s = set_iterative_seed(3407, "train_explainer[epoch=1]")
    assert s == torch.initial_seed()


def test_merge_rules():
    from autognothi_amd.utils.nnmodel import New, merge_items
    src = {"a.0.w": 1, "a.1.w": 2, "b.x": 3, "c": 4}
    out = merge_items([({"a.{i}.w": "z.{i}.weight", "b.{_}": None, "c": [..., "c2"]}, src)],
                      {"z.0.weight": 0}, duplicate_action=lambda v: v * 10)
    assert out == {"z.0.weight": 1, "z.1.weight": 2, "c": 4, "c2": 40}
    with pytest.raises(ValueError):  # destination key neither produced nor declared New
        merge_items([({"a.{i}.w": ..., "b.{_}": None, "c": ...}, src)], {"fresh.k": 9}, lambda v: v)
    out = merge_items([({"a.{i}.w": ..., "b.{_}": None, "c": ..., New(): "fresh.{_}"}, src)], {"fresh.k": 9}, lambda v: v)
    assert out["fresh.k"] == 9 and out["a.1.w"] == 2 and "b.x" not in out
    with pytest.raises(ValueError):  # unmatched source key
        merge_items([({"a.{i}.w": ...}, src)], {}, lambda v: v)


def test_state_dict_layout_contract():
    """Key names/shapes recorded from the reference classes (tests/golden/state_keys.json)."""
    from autognothi_amd.recipes import get_recipe
    g = dict(golden_json("state_keys.json"))
    g.update(golden_json("state_keys_ltt.json"))
    for kind, entry in g.items():
        recipe = get_recipe(kind)
        cfg = recipe.t_config(**entry["params"])
        for role, want in entry["roles"].items():
            mod = getattr(recipe, "t_" + role)(cfg)
            got = {k: list(v.shape) for k, v in mod.state_dict().items()}
            assert got == want, (kind, role)


def test_converters_round_trip_on_cpu_weights():
    """conv_classifier_surrogate / conv_surrogate_explainer keep the backbone and add fresh heads."""
    from autognothi_amd.recipes import get_recipe
    g = golden_json("state_keys.json")
    for kind in ("vanilla_vit", "froyo_vit", "duo_vanilla_vit"):
        recipe = get_recipe(kind)
        cfg = recipe.t_config(**g[kind]["params"])
        cls = recipe.t_classifier(cfg)
        srg = recipe.conv_classifier_surrogate(cfg, None, cls)
        exp = recipe.conv_surrogate_explainer(cfg, None, srg)
        a, b, c = cls.state_dict(), srg.state_dict(), exp.state_dict()
        for k in a:
            assert torch.equal(a[k], b[k])
            if k.startswith("vit."):
                assert torch.equal(a[k], c[k])
        with pytest.raises(RuntimeError):  # forward has no CPU path
            with torch.no_grad():
                recipe.fw_surrogate(srg, torch.zeros(1, 3, cfg.img_px_size, cfg.img_px_size), torch.ones(1, recipe.n_players(cfg), dtype=torch.long))


def test_recipe_registry():
    from autognothi_amd.recipes import get_recipe
    assert get_recipe("vanilla_vit").id == "vanilla_bert"  # the reference's own copy-paste id (recipes/vanilla_vit.py:37)
    assert get_recipe("ltt_vit").id == "ltt_vit" and get_recipe("ltt_bert").id == "ltt_bert"
    with pytest.raises(ValueError):
        get_recipe("kernel_shap_bert")


def test_ltt_converters_move_the_ladders():
    """LTT stage machine (reference recipes/ltt_vit.py:84-224): classifier -> surrogate -> explainer keep the frozen
    backbone; the surrogate's ladder stays branch 0 of Final and the explainer's becomes branch 1."""
    from autognothi_amd.recipes import get_recipe
    from autognothi_amd.utils import synth
    g = golden_json("state_keys_ltt.json")
    recipe = get_recipe("ltt_vit")
    cfg = recipe.t_config(**g["ltt_vit"]["params"])
    cls = recipe.t_classifier(cfg)
    synth.load_synth_weights(cls, seed=3)
    srg = recipe.conv_classifier_surrogate(cfg, None, cls)
    exp = recipe.conv_surrogate_explainer(cfg, None, srg)
    synth_exp = {k: v.clone() for k, v in exp.state_dict().items()}
    a, b, c = cls.state_dict(), srg.state_dict(), exp.state_dict()
    for k in a:
        assert torch.equal(a[k], b[k])
        if k.startswith("vit.") or k.startswith("classifier."):
            assert torch.equal(a[k], c[k])
    assert not any(k.startswith("s_attn_classifier") for k in c)
    # Final: rules only (the null replay needs the GPU) — apply the same rules with a given surrogate_null
    from autognothi_amd.recipes import ltt_vit as r
    from autognothi_amd.utils.nnmodel import merge_state_dicts
    final = recipe.t_final(cfg)
    backbone = {"vit.embeddings.{_}": ..., "vit.encoder.layers.{_}": ..., "vit.layernorm.{wb}": ..., "classifier.{wb}": ...}
    drop = {"vit.embeddings.{_}": None, "vit.encoder.layers.{_}": None, "vit.layernorm.{wb}": None, "classifier.{_}": None}
    rc = dict(backbone); rc.update(r._side_rules(0, None, None)); rc["s_attn_classifier.{wb}"] = None
    rs = dict(drop); rs.update(r._side_rules(0, None, ...)); rs["s_attn_classifier.{wb}"] = ...
    re_ = dict(drop); re_.update(r._side_rules(0, 1, "move")); re_["s_explainer_attn.{_}"] = ...; re_["s_explainer_mlp.{_}"] = ...
    merge_state_dicts((rc, cls), (rs, srg), (re_, exp), ({"surrogate_null": ...}, {"surrogate_null": torch.zeros(1, cfg.num_labels)}), into=final)
    f = final.state_dict()
    assert torch.equal(f["vit.encoder.s_attn_maps.0_1.weight"], b["vit.encoder.s_attn_maps.0_1.weight"])
    assert torch.equal(f["vit.encoder.s_attn_maps.1_1.weight"], synth_exp["vit.encoder.s_attn_maps.0_1.weight"])
    assert torch.equal(f["vit.s_attn_layernorm.1.weight"], synth_exp["vit.s_attn_layernorm.0.weight"])
    assert torch.equal(f["s_explainer_mlp.5.weight"], synth_exp["s_explainer_mlp.5.weight"])


def test_flop_accounting_matches_survey():
    import bench
    for name, want in (("vit_base", 35.13e9), ("vit_tiny", 2.507e9), ("vit_large", 123.1e9), ("bert_base", 22.35e9)):
        kind, params, _ = bench.WORKLOADS[name]
        t = 197 if "vit" in name else 128
        assert abs(bench.flops_per_forward(kind, params, t) - want) / want < 2e-3


def test_checkpoint_wire_format_round_trip(tmp_path):
    """reference scripts/resources.py:150-271: `{section}-epoch-{n}.ckpt` plain state dicts; retention by `ckpt_when`."""
    from autognothi_amd.recipes import get_recipe
    from autognothi_amd.scripts import resources as rs
    from autognothi_amd.utils import synth
    assert [e for e in range(0, 31) if rs.ranged_modulo_test("<=10:%2==1; _:%10==0")(e)] == [1, 3, 5, 7, 9, 20, 30]
    # the reference's own known-answer vectors for this helper (utils/strings.py:176-184)
    for patt, expected in [("<=10:%2==0; <=5:%3==1; <= 20 : %5 == 0", ".*..*.*.*.*....*....*"), (" <=6:%4==2 ;", "..*...*......."),
                           ("<=5:%2==1; _:%3==0", ".*.*.**..*..*..*..")]:
        fn = rs.ranged_modulo_test(patt)
        assert "".join("*" if fn(i) else "." for i in range(len(expected))) == expected
    g = golden_json("state_keys.json")
    recipe = get_recipe("vanilla_vit")
    cfg = recipe.t_config(**g["vanilla_vit"]["params"])
    srg = recipe.t_surrogate(cfg)
    synth.load_synth_weights(srg, seed=4)
    for ep in range(0, 5):
        rs.save_epoch_ckpt(tmp_path, "surrogate", "_:%2==0", 4, ep, srg)
    assert rs.get_epoch_ckpts(tmp_path, "surrogate", 10) == [0, 2, 4]          # 1 and 3 were dropped when 2 and 4 arrived
    epoch, model = rs.load_epoch_model(tmp_path, recipe, cfg, "surrogate", 10)
    assert epoch == 4 and not model.training
    for k, v in srg.state_dict().items():
        assert torch.equal(v, model.state_dict()[k])
    raw = torch.load(rs.ckpt_path(tmp_path, "surrogate", 4), weights_only=False)
    assert list(raw.keys()) == list(g["vanilla_vit"]["roles"]["surrogate"].keys())   # the reference's own key order / names
    with pytest.raises(FileNotFoundError):
        rs.load_epoch_ckpt(tmp_path, "explainer", 3, required=True)


def test_merge_items_reference_known_answers():
    """The reference's own unit-test vectors for the state-dict merge engine (utils/nnmodel.py:242-307), as data."""
    from autognothi_amd.utils.nnmodel import New, merge_items
    src_1 = {"alpha.default.0": 0, "alpha.default.1": 0, "alpha.0": 0, "alpha.1": 0, "beta.2": 0, "gamma.3": 0}
    src_2 = {"iota.0": 1, "kappa.1": 1}
    dest = {"gamma.3": 2, "theta.4": 2}
    rules_1 = {"alpha.default.{_}": ..., "alpha.{_}": [..., "epsilon.{_}", "zeta.{_}"], "beta.{_}": None, "gamma.{_}": None,
               New(): "gamma.{_}", New(): "theta.{_}"}
    rules_2 = {"iota.{_}": ..., "kappa.{_}": None}
    actual = merge_items([(rules_1, src_1), (rules_2, src_2)], dest, duplicate_action=lambda x: x + 5)
    assert actual == {"alpha.default.0": 0, "alpha.default.1": 0, "alpha.0": 0, "alpha.1": 0, "epsilon.0": 5, "epsilon.1": 5,
                      "gamma.3": 2, "iota.0": 1, "theta.4": 2, "zeta.0": 5, "zeta.1": 5}
    with pytest.raises(ValueError):
        merge_items([({"alpha.{_}": "beta.{_}"}, {"alpha.0": 0, "alpha.1": 0})], {"beta.0": 1, "beta.1": 1, "gamma.0": 1}, lambda x: x)


def test_set_iterative_seed_reference_property():
    """utils/tools.py:57-70: the same (master, key) reseeds python's generator to the same stream."""
    import random
    from autognothi_amd.utils.tools import set_iterative_seed

    def get(key):
        set_iterative_seed(3407, key)
        return random.randint(0, 1000)
    a, b, c = get("stage-a"), get("stage-b"), get("stage-c")
    assert (get("stage-c"), get("stage-a"), get("stage-b")) == (c, a, b)


def test_torch_cpu_generator_state_fields_round_trip():
    """(mt, pos) <-> torch's CPU generator state (at::mt19937: left == 625 - next inside a block, left == 1 when the next
    draw twists): the state handed back by DeviceMT19937.export_to_torch_cpu must continue the host stream across block
    boundaries.  Host-only: the field arithmetic, no device."""
    from autognothi_amd import ops
    for n in (0, 1, 10, 386, 623, 624, 625, 1000, 1248):
        g = torch.Generator()
        g.manual_seed(3407)
        if n:
            torch.rand(n, generator=g)
        mt, pos = ops.parse_torch_cpu_state(g.get_state())
        assert pos == (624 if n % 624 == 0 else n % 624)
        h = torch.Generator()
        h.manual_seed(1)
        h.set_state(ops.fill_torch_cpu_state(h.get_state(), mt, pos))
        assert torch.equal(torch.rand(1500, generator=h), torch.rand(1500, generator=g)), n


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts `python -m torch.distributed.run --nproc-per-node N ... bench.py` as a
    child (before any GPU call) and leaves with its exit code; under a launcher (WORLD_SIZE set) or at N = 1 it does nothing."""
    import argparse
    import sys
    import bench
    calls = []
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 7)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.maybe_spawn(argparse.Namespace(gpus=4))
    assert e.value.code == 7 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert os.path.basename(cmd[-5]) == "bench.py" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    bench.maybe_spawn(argparse.Namespace(gpus=1))                 # single GPU: in-process
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.maybe_spawn(argparse.Namespace(gpus=4))                 # already a rank of a launched job
    assert len(calls) == 1


def test_epoch_wrapper_binds_arguments_by_name():
    """the reference calls its epoch bodies by keyword too (_explainer_epoch_train(env=..., device=...)); the
    epoch-stream wrapper must find `device` wherever it is passed, and on a CPU device it is the plain call."""
    import torch
    from autognothi_amd.scripts import common
    seen = []

    @common.on_epoch_stream
    def body(env, device, n, flag=False):
        seen.append((env, device, n, flag))
        return n + 1

    cpu = torch.device("cpu")
    assert body("e", cpu, 1) == 2
    assert body(env="e", device=cpu, n=2, flag=True) == 3
    assert body("e", n=3, device=cpu) == 4
    assert seen == [("e", cpu, 1, False), ("e", cpu, 2, True), ("e", cpu, 3, False)]


def test_schedule_is_named_on_request(caplog):
    """the epoch names the schedule it took: always on the autognothi_amd.schedule logger, in the epoch log only when env.log_schedule is
    set (the reference's log has no such line)."""
    import logging
    from autognothi_amd.scripts import common

    class Env:
        def __init__(self, on):
            self.lines, self.log_schedule = [], on

        def log(self, msg):
            self.lines.append(msg)

    quiet, loud = Env(False), Env(True)
    with caplog.at_level(logging.INFO, logger="autognothi_amd.schedule"):
        common.log_schedule(quiet, None)
        common.log_schedule(loud, None)
    assert quiet.lines == [] and loud.lines == ["  > schedule: one stream"]
    assert sum("one stream" in r.getMessage() for r in caplog.records) == 2
