"""CPU: the experiment environment in the reference's on-disk format (autognothi_amd/scripts/env.py vs reference scripts/env.py,
scripts/types.py ExpConfig): the reference's own experiment configs (tests/golden/hparams/*.hparams.json, copied and digested by
make_golden.py hparams through the reference's pydantic model and get_recipe) load into the view the entry points use, pick the
same recipe and model config, survive flush_cfg byte for byte, and the log / metrics lines have the reference's shape."""
import json
import os
import shutil

import pytest

HP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hparams")
EXPECTED = json.load(open(os.path.join(HP, "expected.json")))


def _model_dir(tmp_path, exp):
    shutil.copy(os.path.join(HP, f"{exp}.hparams.json"), tmp_path / ".hparams.json")
    return tmp_path


@pytest.mark.parametrize("exp", sorted(EXPECTED))
def test_reference_experiment_configs_load(tmp_path, exp):
    from autognothi_amd.scripts.env import ExpEnv
    from autognothi_amd.scripts.resources import get_recipe
    want = EXPECTED[exp]
    env = ExpEnv(_model_dir(tmp_path, exp), echo=False)
    cfg = env.config
    assert cfg.net.kind == want["kind"] and cfg.seed == 3407
    recipe, m_cfg = get_recipe(cfg)
    assert recipe.n_players(m_cfg) == want["n_players"]
    assert type(m_cfg).__name__ == want["config_class"]
    te = cfg.train_explainer
    assert {"epochs": te.epochs, "n_mask_samples": te.n_mask_samples, "batch_size": te.batch_size} == want["train_explainer"]
    assert cfg.train_surrogate.EXPERIMENTAL_progressive_training == want["progressive"]["surrogate"]
    assert te.EXPERIMENTAL_progressive_training == want["progressive"]["explainer"]
    assert cfg.eval_faithfulness.resolution >= 1 and cfg.eval_cls_acc.on_exp_epochs is None
    # flush_cfg writes the file back unchanged (json, indent 2, trailing newline: reference env.py:117-124)
    before = open(tmp_path / ".hparams.json").read()
    env.flush_cfg()
    assert json.loads(open(tmp_path / ".hparams.json").read()) == json.loads(before)
    # ... and carries what a run changes
    cfg.train_explainer.epochs = 7
    env.flush_cfg()
    assert json.load(open(tmp_path / ".hparams.json"))["train_explainer"]["epochs"] == 7
    assert "EXPERIMENTAL_progressive_training" not in json.load(open(tmp_path / ".hparams.json"))["train_explainer"]


def test_env_log_metrics_fork_and_validation(tmp_path):
    from autognothi_amd.scripts.env import ExpEnv, parse_config
    d = _model_dir(tmp_path, "vit_base_imagenette_vanilla")
    with ExpEnv(d, echo=False) as env:
        env.metrics({"epoch": 1, "train_reg_loss": 0.25, "test_plots": []})
        sub = env.fork(lambda c: c.logger_explainer)
        sub.log("  > epoch 1 done")
        assert sub.config is env.config and sub.model_path == env.model_path
    lines = open(d / ".log.txt").read().splitlines()
    assert "NEW RUN: load config from" in lines[0] and lines[0].startswith("[20")
    assert any("METRICS: {'epoch': 1, 'train_reg_loss': 0.25, 'test_plots': '<list>'}" in ln for ln in lines)
    assert lines[-1].endswith("  > epoch 1 done")
    raw = json.load(open(d / ".hparams.json"))
    bad = json.loads(json.dumps(raw)); bad["net"]["version"] = "beta.0.99"
    with pytest.raises(ValueError, match="version mismatch"):
        parse_config(bad)
    bad = json.loads(json.dumps(raw)); del bad["train_surrogate"]
    with pytest.raises(ValueError, match="train_surrogate"):
        parse_config(bad)
