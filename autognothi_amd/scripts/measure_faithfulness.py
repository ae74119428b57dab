"""Insertion / deletion faithfulness curves (reference scripts/measure_faithfulness.py:41-251) on the HIP path.

Per test sample the reference runs 2*C masked-forward batches of `steps` rows each through a Python loop
(:195-218).  Here all 2*C*steps perturbation masks of a sample are built by one ranking kernel and
evaluated in ONE masked forward whose rows all share the sample's embeddings / layer-0 projections.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from .. import distributed, ops
from ..recipes.types import ModelRecipe
from .common import Log

CurvePoint = Dict[int, Dict[int, float]]  # class -> stop -> metric


def auc(curve: Dict[int, float]) -> float:
    """reference :143-146."""
    vals = np.array(list(curve.values()))
    return float(((vals[1:] + vals[:-1]) / 2).mean())


def explain(recipe: ModelRecipe, m_final, xs: Tensor) -> Tensor:
    """reference _explain (:174-180) -> attributions [1, C, P]."""
    m_final.eval()
    with torch.no_grad():
        _logits, attr = recipe.fw_final(m_final, xs)
    return attr


def infer_perturbed(recipe: ModelRecipe, m_surrogate, xs: Tensor, explanation: Tensor, steps: int,
                    shard_rows: bool = False) -> Tuple[CurvePoint, CurvePoint]:
    """reference _infer (:183-221) for mask_base 0 (insertion) and 1 (deletion) at once.
    xs [1, ...]; explanation [1, C, P] -> (insertion curves, deletion curves).
    ``shard_rows`` (N > 1 ranks, every rank calling with the SAME sample: fewer samples than ranks): this rank evaluates its
    contiguous slice of the 2*C*S perturbation rows of the one input — the K-within-image split of SURVEY §8e — and the rows are
    all-gathered (C floats each), so every rank returns the full curves."""
    _, n_classes, n_players = explanation.shape
    attr = explanation[0].contiguous().float()                       # [C, P]
    stops, m_ins = ops.perturbed_masks(attr, steps, 0)               # [S], [C, S, P]
    _, m_del = ops.perturbed_masks(attr, steps, 1)
    s = stops.shape[0]
    masks = torch.cat([m_ins.reshape(n_classes * s, n_players), m_del.reshape(n_classes * s, n_players)], dim=0)
    m_surrogate.eval()
    _, n_ranks = distributed.world()
    with torch.no_grad():
        if shard_rows and n_ranks > 1:
            spans = [distributed.shard_range(masks.shape[0], q, n_ranks) for q in range(n_ranks)]
            lo, hi = spans[distributed.world()[0]]
            ys_loc, _ = recipe.fw_surrogate(m_surrogate, xs, masks[lo:hi].contiguous())
            ys = distributed.gather_rows(ys_loc.contiguous(), [h_ - l_ for l_, h_ in spans])
        else:
            ys, _ = recipe.fw_surrogate(m_surrogate, xs, masks)      # one input, 2*C*S mask rows
    ys = ys.reshape(2, n_classes, s, -1).cpu().numpy()
    stops_l = stops.cpu().numpy().tolist()
    out: List[CurvePoint] = []
    for base in (0, 1):
        res: CurvePoint = {}
        for c in range(n_classes):
            # duplicate stops collapse because the reference keys results by stop (:205-218)
            res[c] = {int(st): float(ys[base, c, i, c]) for i, st in enumerate(stops_l)}
        out.append(res)
    return out[0], out[1]


def measure_faithfulness(env: Any, device: torch.device, d_loader: Optional[Any], resolution: Optional[int]) -> Dict[str, Any]:
    """reference measure_faithfulness(env, device, d_loader, resolution) (:41-140): load the surrogate + final checkpoints
    of ``env.model_path``, walk ``d_loader.test(1)`` and return the report fields (as a dict: the reference's
    MeasureFaithfulnessReport / FaithfulnessCurve are pydantic containers of the same keys).  ``env`` is duck-typed:
    ``.config`` (``net``, ``train_*.epochs``, ``eval_faithfulness.resolution``), ``.model_path``, ``.log``; ``d_loader`` None
    falls back to ``env.d_loader`` (scripts/resources.load_cfg_dataset)."""
    from .resources import get_recipe, load_cfg_dataset, load_epoch_model_env
    env = distributed.main_only(env)
    env.log("loading final model...")
    config = env.config
    m_recipe, m_config = get_recipe(config)
    if not m_recipe.measurements.allow_faithfulness:
        raise ValueError("unsupported recipe action")
    _, m_surrogate = load_epoch_model_env(env, m_recipe, "surrogate", device=device)
    _, m_final = load_epoch_model_env(env, m_recipe, "final", device=device)
    m_misc = m_recipe.load_misc(env.model_path, m_config)
    gen_input = m_recipe.gen_input(m_config, m_misc, device)
    if d_loader is None:
        env.log("loading dataset...")
        d_loader = load_cfg_dataset(env, getattr(getattr(config, "eval_faithfulness", None), "dataset", None))
    if resolution is None:
        resolution = config.eval_faithfulness.resolution
    env.log("[[[ running measurement... ]]]")
    report = measure_faithfulness_loaded(env, device, m_recipe, m_surrogate, m_final, d_loader.test(1), gen_input, resolution)
    env.log("FINAL RESULTS:\n"
            f"  > insertion: target {report['insertion']['auc']:.6f}, non-target {report['insertion_non_ok']['auc']:.6f}\n"
            f"  > deletion: target {report['deletion']['auc']:.6f}, non-target {report['deletion_non_ok']['auc']:.6f}")
    return report


def measure_faithfulness_loaded(env: Any, device: torch.device, recipe: ModelRecipe, m_surrogate, m_final,
                                samples: Iterable[Tuple[Any, Any]], gen_input: Callable[[Any, Any], Tuple[Tensor, Tensor]],
                                resolution: int) -> Dict[str, Any]:
    """reference measure_faithfulness (:41-140) given loaded models and a test iterator of single samples.
    Returns the report fields (insertion / deletion AUC for target and non-target classes + raw curves).
    N > 1 ranks (the per-image loop :195-218): every rank walks the same iterator in groups of `ranks` samples and
    evaluates sample `rank` of each full group (sharding by image: no collective per sample); a last group with FEWER samples
    than ranks is sharded inside the image instead — every rank runs its slice of the 2*C*S perturbation rows of every remaining
    sample and the rows are all-gathered — so no rank idles on the tail.  The per-sample curves are gathered (host objects, a
    few KB each) and re-ordered by sample index: every rank returns the single-process report."""
    env = distributed.main_only(env) or Log()
    rank, n_ranks = distributed.world()
    mine: List[Tuple[int, int, CurvePoint, CurvePoint]] = []

    def groups(it):
        """the sample stream in groups of `ranks` samples (one per rank); the last group may be short"""
        buf = []
        for item in enumerate(it):
            buf.append(item)
            if len(buf) == n_ranks:
                yield buf
                buf = []
        if buf:
            yield buf

    for group in groups(samples):
        if len(group) == n_ranks:       # one sample per rank (by image: no collective per sample)
            todo, split = [group[rank]], False
        else:                           # fewer samples than ranks (the tail): every rank works on every sample, rows split
            todo, split = group, True
        for i, (_inputs, _targets) in todo:
            xs, zs = gen_input(_inputs, _targets)
            ok_cls = int(zs.item())
            explanation = explain(recipe, m_final, xs)
            ins_curve, del_curve = infer_perturbed(recipe, m_surrogate, xs, explanation, resolution, shard_rows=split)
            if not split or rank == 0:
                mine.append((i, ok_cls, ins_curve, del_curve))
            if n_ranks == 1:
                env.log(f"> sample {i}: ok_cls {ok_cls}, ins^ {auc(ins_curve[ok_cls]):.6f}, del^ {auc(del_curve[ok_cls]):.6f}")
    everything = sorted((item for part in distributed.gather_objects(mine) for item in part), key=lambda it: it[0])
    ok_cls_l: List[int] = [it[1] for it in everything]
    ins_curves: List[CurvePoint] = [it[2] for it in everything]
    del_curves: List[CurvePoint] = [it[3] for it in everything]
    if n_ranks > 1:
        for i, ok_cls, ins_curve, del_curve in everything:
            env.log(f"> sample {i}: ok_cls {ok_cls}, ins^ {auc(ins_curve[ok_cls]):.6f}, del^ {auc(del_curve[ok_cls]):.6f}")

    def paint(curves: List[Dict[int, float]]) -> Dict[str, Any]:
        items: Dict[int, List[float]] = {}
        for curve in curves:
            for st, point in curve.items():
                items.setdefault(st, []).append(point)
        avg = {st: float(np.mean(v)) for st, v in items.items()}
        std = {st: float(np.std(v)) for st, v in items.items()}
        a = np.array(list(avg.values()))
        return {"auc": float(((a[1:] + a[:-1]) / 2).mean()) if len(a) > 1 else float("nan"), "avg": avg, "std": std}

    sel = lambda curves, ok: [c[cl] for c, k in zip(curves, ok_cls_l) for cl in c if (cl == k) == ok]  # noqa: E731
    return {"insertion": paint(sel(ins_curves, True)), "deletion": paint(sel(del_curves, True)),
            "insertion_non_ok": paint(sel(ins_curves, False)), "deletion_non_ok": paint(sel(del_curves, False)),
            "data_cls": ok_cls_l, "data_ins": ins_curves, "data_del": del_curves}
