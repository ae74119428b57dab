"""The plugin interface the pipelines program against — mirror of reference recipes/types.py:96-162
(same field names and callable signatures), so a recipe from this package drops into
``scripts/resources.get_recipe`` unchanged."""
from __future__ import annotations

import dataclasses
import enum
import pathlib
from typing import Any, Callable, Generic, Literal, Optional, Tuple, Type, TypeVar, Union

import pydantic
import torch
from torch import Tensor, nn

TConfig = TypeVar("TConfig", bound=pydantic.BaseModel)
TMisc = TypeVar("TMisc")
TClassifier = TypeVar("TClassifier", bound=nn.Module)
TSurrogate = TypeVar("TSurrogate", bound=nn.Module)
TExplainer = TypeVar("TExplainer", bound=nn.Module)
TFinal = TypeVar("TFinal", bound=nn.Module)

RECIPE_VERSION = "beta.1.01"  # checked by scripts/resources.py:79-82


class ModelMode(enum.Enum):
    classifier_eval = "classifier_eval"
    surrogate_eval = "surrogate_eval"
    surrogate_train = "surrogate_train"
    explainer_eval = "explainer_eval"
    explainer_train = "explainer_train"


@dataclasses.dataclass
class ModelRecipe_Training:
    support_classifier: bool
    support_surrogate: bool
    support_explainer: bool
    exp_variant_duo: bool
    exp_variant_kernel_shap: bool


@dataclasses.dataclass
class ModelRecipe_Measurements(Generic[TConfig, TClassifier, TExplainer]):
    verify_final_coherency: bool
    allow_accuracy: bool
    allow_faithfulness: bool
    allow_cls_acc: bool
    allow_performance_cls: bool
    allow_performance_srg_exp: bool
    allow_performance_fin: bool
    allow_train_resources: bool
    allow_dual_task_similarity: Union[Literal[False], Any]
    allow_branches_cka: bool


@dataclasses.dataclass
class ModelRecipe(Generic[TConfig, TMisc, TClassifier, TSurrogate, TExplainer, TFinal]):
    id: str
    version: str
    t_config: Type[TConfig]
    t_classifier: Type[TClassifier]
    t_surrogate: Type[TSurrogate]
    t_explainer: Type[TExplainer]
    t_final: Type[TFinal]

    load_misc: Callable[[pathlib.Path, TConfig], TMisc]
    conv_pretrained_classifier: Callable[[TConfig, Union[nn.Module, Any]], TClassifier]
    conv_classifier_surrogate: Callable[[TConfig, TMisc, TClassifier], TSurrogate]
    conv_surrogate_explainer: Callable[[TConfig, TMisc, TSurrogate], TExplainer]
    conv_explainer_final: Callable[[TConfig, TMisc, TClassifier, TSurrogate, TExplainer], TFinal]

    n_players: Callable[[TConfig], int]
    gen_input: Callable[[TConfig, TMisc, torch.device], Callable[[Any, Any], Tuple[Tensor, Tensor]]]
    gen_null: Callable[[TConfig, TMisc, torch.device], Tensor]

    training: ModelRecipe_Training

    # :: (F, Xs, mask[R,P]) -> Ys[R,C], Ys verbatim.   Xs may hold R rows (reference behaviour) or
    #    B = R/K rows (this package's extension: the K masked copies share the input).
    fw_classifier: Callable[[TClassifier, Tensor, Tensor], Tuple[Tensor, Tensor]]
    fw_surrogate: Callable[[TSurrogate, Tensor, Tensor], Tuple[Tensor, Optional[Tensor]]]
    # :: (F, Xs, mask, surrogate_grand[B,C], surrogate_null[1,C]) -> shap[B,C,P], logits?
    fw_explainer: Callable[[TExplainer, Tensor, Tensor, Tensor, Tensor], Tuple[Tensor, Optional[Tensor]]]
    # :: (F, Xs) -> Ys, shap
    fw_final: Callable[[TFinal, Tensor], Tuple[Tensor, Tensor]]

    measurements: ModelRecipe_Measurements
