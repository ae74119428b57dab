"""Host glue mirrored from reference utils/nnmodel.py: parameter freezing (:48-60), the
state-dict merge rules used by the recipes' ``conv_*`` converters (:63-191) and the observation
mixin the ``Final`` modules call (:194-239).  Own implementation; same behaviour at the call sites."""
from __future__ import annotations

import re
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import torch
from torch import nn


def freeze_model_parameters(on: nn.Module, *item_names: Any, requires_grad: bool = False) -> None:
    """freeze(model, "vit", "classifier") freezes parameters under those prefixes; freeze(model, ...)
    freezes everything."""
    if len(item_names) == 1 and item_names[0] is ...:
        for p in on.parameters():
            p.requires_grad = requires_grad
        return
    for name, p in on.named_parameters():
        if any(name.startswith(f"{n}.") for n in item_names):
            p.requires_grad = requires_grad


class New:
    """Rule key meaning "this destination key is freshly initialised" (each instance is a distinct dict key)."""
    _count = 0

    def __init__(self):
        New._count += 1
        self._id = New._count

    def __hash__(self):
        return hash(("New", self._id))

    def __eq__(self, other):
        return isinstance(other, New) and other._id == self._id

    def __repr__(self):
        return "new()"


MergeStateDictRules = Dict[Union[str, New], Union[str, type(Ellipsis), List[Union[str, type(Ellipsis)]], None]]


def _compile(pattern: str) -> Tuple[re.Pattern, List[str]]:
    """'a.{i}.b.{wb}' -> regex with one lazy group per placeholder, and the placeholder names."""
    names: List[str] = []
    out = ""
    for part in re.split(r"(\{[^}]*\})", pattern):
        if part.startswith("{") and part.endswith("}"):
            names.append(part[1:-1])
            out += "(.*?)"
        else:
            out += re.escape(part)
    return re.compile(out), names


def _rewrite(src_pattern: str, dst_pattern: str) -> Callable[[str], Optional[str]]:
    rx, names = _compile(src_pattern)

    def fn(key: str) -> Optional[str]:
        m = rx.fullmatch(key)
        if m is None:
            return None
        binds = dict(zip(names, m.groups()))
        return re.sub(r"\{([^}]*)\}", lambda g: binds[g.group(1)], dst_pattern)

    return fn


def merge_items(rules_src: List[Tuple[MergeStateDictRules, Dict[str, Any]]], dest: Dict[str, Any],
                duplicate_action: Callable[[Any], Any] = lambda v: v) -> Dict[str, Any]:
    """Apply rules:  'pat' -> 'pat2' rename | 'pat' -> ... keep | 'pat' -> [..] fan-out | 'pat' -> None drop |
    New() -> 'pat' keep the destination's own (fresh) value.  Every source key must match a rule and every
    destination key must be produced or declared New — otherwise ValueError("merge failed")."""
    ok = True
    result: Dict[str, Any] = {}
    new_matchers = []
    for rules, src in rules_src:
        edits: List[List[Callable[[str], Optional[str]]]] = []
        drops: List[re.Pattern] = []
        for k, v in rules.items():
            if isinstance(k, New):
                if not isinstance(v, str):
                    raise ValueError(f"invalid rule: {k} -> {v}")
                new_matchers.append(_compile(v)[0])
            elif v is None:
                drops.append(_compile(k)[0])
            elif v is Ellipsis:
                edits.append([_rewrite(k, k)])
            elif isinstance(v, str):
                edits.append([_rewrite(k, v)])
            elif isinstance(v, list):
                if any(not (isinstance(x, str) or x is Ellipsis) for x in v):
                    raise ValueError(f"invalid rule: {k} -> {v}")
                if v:
                    edits.append([_rewrite(k, k if x is Ellipsis else x) for x in v])
                else:
                    drops.append(_compile(k)[0])
            else:
                raise ValueError(f"invalid rule: {k} -> {v}")
        for key, val in src.items():
            targets = None
            for fns in edits:
                outs = [f(key) for f in fns]
                if all(o is not None for o in outs):
                    targets = outs
                    break
            if targets is not None:
                for i, nk in enumerate(targets):
                    if nk in result:
                        print(f" [!] duplicate key: {nk}")
                        ok = False
                    result[nk] = val if i == 0 else duplicate_action(val)
                continue
            if any(rx.fullmatch(key) for rx in drops):
                continue
            print(f" [!] no rule matches key from `from_model`: {key}")
            ok = False
    for key, val in dest.items():
        if key in result:
            continue
        if any(rx.fullmatch(key) for rx in new_matchers):
            result[key] = val
            continue
        print(f" [!] ignored key from `into_model`: {key}")
        ok = False
    if not ok:
        raise ValueError("merge failed")
    return result


def merge_state_dicts(*rules_src: Tuple[MergeStateDictRules, Union[nn.Module, Any]], into: nn.Module) -> None:
    srcs = [(rules, m.state_dict() if isinstance(m, nn.Module) else m) for rules, m in rules_src]
    merged = merge_items(srcs, into.state_dict(),
                         duplicate_action=lambda v: v.clone() if isinstance(v, torch.Tensor) else v)
    into.load_state_dict(merged)


class ObservableModuleMixin:
    """Lets a caller retain the backbone representation of the last forward (used by the Final modules
    and the CKA measurement; reference utils/nnmodel.py:194-239)."""

    def __init__(self):
        self._om_observing = False
        self._om_features: Optional[Dict[str, torch.Tensor]] = None

    @classmethod
    def using(cls, m: nn.Module) -> "ObservableModuleMixin":
        if not isinstance(m, ObservableModuleMixin):
            raise ValueError("not an ObservableModuleMixin")
        return m

    def om_is_observing(self) -> bool:
        return self._om_observing

    def om_retain_observations(self, flag: bool = True) -> None:
        self._om_observing = flag
        if not flag:
            self._om_features = None

    def om_record_features(self, repr_cls=None, repr_srg=None, repr_exp=None, extra=None) -> None:
        if not self._om_observing:
            return
        feats = {"repr_cls": repr_cls, "repr_srg": repr_srg, "repr_exp": repr_exp}
        feats.update(extra or {})
        self._om_features = {k: v for k, v in feats.items() if v is not None}

    def om_observe(self) -> Dict[str, torch.Tensor]:
        if self._om_features is None:
            raise ValueError("no features to observe. use `forward()` first")
        return self._om_features

    def om_take_observations(self) -> Dict[str, torch.Tensor]:
        feats, self._om_features = self._om_features, None
        return feats or {}
