"""All evaluation reports of a model directory (reference scripts/measure_all.py): each report is computed once and kept as
``<model_path>/.reports/<name>.json`` (json, indent 2: what the reference's playground/grab_results.py scrapes); an existing file is
loaded instead of re-measured (:112-135).  The CKA and dual-task-similarity reports are outside this build's scope (VERDICT
r1) and are skipped with a log line when asked for."""
from __future__ import annotations

import json
from typing import Any, Callable, Dict, Optional

import torch

from .measure_accuracy import measure_accuracy, measure_cls_acc
from .measure_faithfulness import measure_faithfulness
from .measure_performance import measure_performance
from .measure_train_resources import measure_train_resources
from .resources import get_recipe


def _to_jsonable(report: Any) -> Any:
    if hasattr(report, "model_dump_json"):
        return json.loads(report.model_dump_json(by_alias=True))
    return json.loads(json.dumps(report))          # (plain dict reports: int keys become strings, as pydantic writes them)


def load_or_run_report(env: Any, filename: str, run: Callable[[], Any]) -> Any:
    """reference load_or_run_report (:112-135) -> the report as plain JSON data."""
    f_path = env.model_path / ".reports" / filename
    if f_path.exists():
        with open(f_path, "r", encoding="utf-8") as f:
            return json.load(f)
    data = _to_jsonable(run())
    f_path.parent.mkdir(parents=True, exist_ok=True)
    with open(f_path, "w", encoding="utf-8") as f:
        f.write(json.dumps(data, indent=2) + "\n")
    return data


def measure_all(env: Any, device: torch.device, run_accuracy: bool = True, run_faithfulness: bool = True, run_cls_acc: bool = True,
                run_performance: bool = True, run_train_resources: bool = True, run_branches_cka: bool = False,
                run_dual_task_similarity: bool = False) -> Dict[str, Any]:
    """reference measure_all(env, device, run_*) (:24-106) -> {report name: data} of what ran or was loaded."""
    m_recipe, _ = get_recipe(env.config)
    meas = m_recipe.measurements
    done: Dict[str, Any] = {}

    def run_report(filename: str, run: Callable[[], Any], recipe_allow: bool, cli_allow: bool) -> None:
        name = filename.split(".")[0]
        if not recipe_allow:
            return
        if not cli_allow:
            env.log(f"[[[ skip: {name} ]]]")
            return
        env.log(f"[[[ Measuring: {name} ]]]")
        done[name] = load_or_run_report(env, filename, run)

    run_report("accuracy.json", lambda: measure_accuracy(env, device, None), meas.allow_accuracy, run_accuracy)
    run_report("faithfulness.json", lambda: measure_faithfulness(env, device, None, None), meas.allow_faithfulness, run_faithfulness)
    run_report("cls_acc.json", lambda: measure_cls_acc(env, device, None), meas.allow_cls_acc, run_cls_acc)
    run_report("performance.json", lambda: measure_performance(env, device, None),
               meas.allow_performance_cls or meas.allow_performance_srg_exp or meas.allow_performance_fin, run_performance)
    run_report("train_resources.json", lambda: measure_train_resources(env, device, None), meas.allow_train_resources, run_train_resources)
    for name, asked in (("branches_cka", run_branches_cka), ("dual_task_similarity", run_dual_task_similarity)):
        if asked:
            env.log(f"[[[ skip: {name} (outside this build's scope) ]]]")
    env.log("[[[ done all measurements ]]]")
    return done
