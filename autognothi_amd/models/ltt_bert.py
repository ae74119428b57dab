"""MI355X-native mirror of reference ``models/ltt_bert.py`` (LTT = ladder / side-network tuning on a frozen
BERT): same class names, constructor arguments, ``forward`` signatures and ``state_dict`` keys; ``forward``
drives the HIP kernels through the C ABI.  See ``ltt_vit.py`` for the ladder; differences here (reference
:404-500): post-LN BERT layers (backbone and h-wide side layers), additive key mask, no final LayerNorms on the
backbone or the side outputs, pooler heads (tanh) on both.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import pydantic
import torch
from torch import Tensor, nn
from typing_extensions import Self

from .. import _lib as L
from .. import engine, ops
from ..utils.nnmodel import ObservableModuleMixin, freeze_model_parameters
from .. import autograd as _ag
from .vanilla_bert import (VanillaBertConfig, VanillaBertEmbeddings, VanillaBertLayer, VanillaBertModel, VanillaBertPooler,
                           _BertHead, _no_autograd)


class LttBertConfig(pydantic.BaseModel):
    """reference models/ltt_bert.py:20-66"""

    attention_probs_dropout_prob: float
    explainer_s_attn_num_layers: int  # side head
    explainer_s_head_hidden_size: int  # side head
    explainer_normalize: bool  # side head
    hidden_dropout_prob: float
    hidden_size: int
    intermediate_size: int
    layer_norm_eps: float
    max_position_embeddings: int
    num_attention_heads: int
    num_hidden_layers: int
    num_labels: int
    pad_token_id: int
    s_attn_hidden_size: int  # side attention
    s_attn_intermediate_size: int  # side attention
    type_vocab_size: int
    vocab_size: int

    @property
    def is_decoder(self) -> bool:
        return False

    def into(self) -> VanillaBertConfig:
        return VanillaBertConfig(
            attention_probs_dropout_prob=self.attention_probs_dropout_prob,
            explainer_attn_num_layers=self.explainer_s_attn_num_layers,
            explainer_head_hidden_size=self.explainer_s_head_hidden_size,
            explainer_normalize=self.explainer_normalize,
            hidden_dropout_prob=self.hidden_dropout_prob,
            hidden_size=self.hidden_size,
            intermediate_size=self.intermediate_size,
            layer_norm_eps=self.layer_norm_eps,
            max_position_embeddings=self.max_position_embeddings,
            num_attention_heads=self.num_attention_heads,
            num_hidden_layers=self.num_hidden_layers,
            num_labels=self.num_labels,
            pad_token_id=self.pad_token_id,
            type_vocab_size=self.type_vocab_size,
            vocab_size=self.vocab_size,
        )


class LttBertMultiEncoder(nn.Module):
    """reference :404-500"""

    def __init__(self, attention_probs_dropout_prob: float, hidden_dropout_prob: float, hidden_size: int,
                 intermediate_size: int, layer_norm_eps: float, num_attention_heads: int, num_hidden_layers: int,
                 num_side_branches: int, s_attn_hidden_size: int, s_attn_intermediate_size: int):
        super().__init__()
        self.num_layers = num_hidden_layers
        self.num_branches = num_side_branches
        self.layers = nn.ModuleList([
            VanillaBertLayer(attention_probs_dropout_prob, hidden_dropout_prob, hidden_size, intermediate_size,
                             layer_norm_eps, num_attention_heads, repl_norm_1_ident=False, repl_norm_2_ident=False)
            for _ in range(num_hidden_layers)])
        maps: Dict[str, nn.Module] = {}
        for i_b in range(num_side_branches):
            for i_ly in range(num_hidden_layers):
                maps[f"{i_b}_{i_ly}"] = nn.Linear(hidden_size, s_attn_hidden_size)
        self.s_attn_maps = nn.ModuleDict(maps)
        side: Dict[str, nn.Module] = {}
        for i_b in range(num_side_branches):
            for i_ly in range(num_hidden_layers):
                side[f"{i_b}_{i_ly}"] = VanillaBertLayer(attention_probs_dropout_prob, hidden_dropout_prob, s_attn_hidden_size,
                                                         s_attn_intermediate_size, layer_norm_eps, num_attention_heads,
                                                         repl_norm_1_ident=False, repl_norm_2_ident=False)
        self.s_attn_layers = nn.ModuleDict(side)
        self._ltt_freeze_layer = num_hidden_layers

    def ltt_freeze_layers_until(self, layer_id: int) -> None:
        self._ltt_freeze_layer = max(1, min(self.num_layers, layer_id))


class LttBertModel(nn.Module):
    """reference :352-401.  ``run`` is the HIP path."""

    def __init__(self, config: LttBertConfig, num_side_branches: int):
        super().__init__()
        self.config = config
        self.num_side_branches = num_side_branches
        self.embeddings = VanillaBertEmbeddings(config.hidden_dropout_prob, config.hidden_size, config.layer_norm_eps,
                                                config.max_position_embeddings, config.pad_token_id, config.type_vocab_size,
                                                config.vocab_size)
        self.encoder = LttBertMultiEncoder(config.attention_probs_dropout_prob, config.hidden_dropout_prob, config.hidden_size,
                                           config.intermediate_size, config.layer_norm_eps, config.num_attention_heads,
                                           config.num_hidden_layers, num_side_branches, config.s_attn_hidden_size,
                                           config.s_attn_intermediate_size)
        self._bb: Dict[int, List[engine.PackedEncoder]] = {}
        self._side: Dict[Tuple[int, str], engine.PackedEncoder] = {}
        self._maps: Dict[str, engine.PackedLinear] = {}

    def _pack(self, t: int) -> None:
        if t in self._bb:
            return
        c = self.config
        self._bb[t] = [engine.PackedEncoder([ly], L.AG_MASK_BERT_ADD, t, c.hidden_size, c.intermediate_size,
                                            c.num_attention_heads, c.layer_norm_eps) for ly in self.encoder.layers]
        for key, ly in self.encoder.s_attn_layers.items():
            self._side[(t, key)] = engine.PackedEncoder([ly], L.AG_MASK_BERT_ADD, t, c.s_attn_hidden_size,
                                                        c.s_attn_intermediate_size, c.num_attention_heads, c.layer_norm_eps)
        if not self._maps:
            for key, lin in self.encoder.s_attn_maps.items():
                self._maps[key] = engine.PackedLinear([lin.weight], [lin.bias])

    def embed(self, input_ids: Tensor, token_type_ids: Optional[Tensor], dtype: int) -> Tensor:
        """the vanilla embeddings (reference :358-366); that code only touches .embeddings / .config"""
        return VanillaBertModel.embed(self, input_ids, token_type_ids, dtype)  # type: ignore[arg-type]

    def run(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor],
            side_layer_branches: Sequence[int]) -> Tuple[Tensor, List[Tensor], Tensor, int]:
        """-> (hidden [R,T,H], [side_b [R,T,h] for b in sorted(side_layer_branches)], mask bits, R), storage dtype."""
        dtype = engine.get_precision()
        c, t = self.config, input_ids.shape[1]
        bits = engine.to_mask_bits(attention_mask, t - 1)
        rows, b = bits.shape[0], input_ids.shape[0]
        if rows % b != 0:
            raise ValueError(f"mask rows ({rows}) must be a multiple of input rows ({b})")
        self._pack(t)
        hidden = self.embed(input_ids, token_type_ids, dtype)
        branches = sorted(set(int(x) for x in side_layer_branches))
        for i_b in branches:
            if not 0 <= i_b < self.num_side_branches:
                raise ValueError(f"side branch {i_b} out of range (model has {self.num_side_branches})")
        side: Dict[int, Optional[Tensor]] = {i_b: None for i_b in branches}
        enc = self.encoder
        for i_ly in range(enc.num_layers):
            share = rows // b if i_ly == 0 else 1
            hidden = self._bb[t][i_ly].forward(hidden, rows, share, bits, False, dtype)
            if i_ly >= enc._ltt_freeze_layer:
                continue
            flat = hidden.view(rows * t, c.hidden_size)
            for i_b in branches:
                key = f"{i_b}_{i_ly}"
                w, bias = self._maps[key].get(dtype)
                if side[i_b] is None:
                    s_new = ops.gemm(flat, w, bias, L.AG_EPI_BIAS_GELU, dtype)
                else:
                    s_new = ops.gemm(flat, w, bias, L.AG_EPI_BIAS_GELU_ADD, dtype, resid=side[i_b].view(rows * t, -1),
                                     rows_per_seq=t, resid_share=1)
                side[i_b] = self._side[(t, key)].forward(s_new.view(rows, t, c.s_attn_hidden_size), rows, 1, bits, False, dtype)
        return hidden, [side[i_b] for i_b in branches], bits, rows

    def run_cls(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor],
                side_layer_branches: Sequence[int]) -> Tuple[Tensor, List[Tensor], int]:
        """CLS rows only: -> (hidden[:, 0:1] [R,1,H], [side_b[:, 0:1] [R,1,h]], R).  With token pruning (engine.PRUNE_BERT_TOKENS)
        the additively masked tokens are dropped after layer 0 from the backbone AND the ladder: a masked key has exactly
        zero weight in every backbone and side layer (models/vanilla_bert.py:523), the heads read the CLS rows only
        (models/ltt_bert.py:108-116), so the masked tokens' rows of both streams are dead."""
        dtype = engine.get_precision()
        c, t = self.config, input_ids.shape[1]
        enc = self.encoder
        if not engine.PRUNE_BERT_TOKENS or enc.num_layers < 2:
            hidden, sides, _, rows = self.run(input_ids, attention_mask, token_type_ids, side_layer_branches)
            return hidden[:, 0:1].contiguous(), [s_[:, 0:1].contiguous() for s_ in sides], rows
        bits = engine.to_mask_bits(attention_mask, t - 1)
        rows, b = bits.shape[0], input_ids.shape[0]
        if rows % b != 0:
            raise ValueError(f"mask rows ({rows}) must be a multiple of input rows ({b})")
        self._pack(t)
        branches = sorted(set(int(x) for x in side_layer_branches))
        # layer 0 on every token (its QKV is shared by the K masks of an input), then pack
        hidden = self._bb[t][0].forward(self.embed(input_ids, token_type_ids, dtype), rows, rows // b, bits, False, dtype)
        # the packed row count stays on the device: the packed section is sized for the upper bound rows * t and its kernels
        # clamp to cu[rows] (rows_dev): no host read, the whole forward can be captured into a hipGraph
        cu, src, n = ops.seq_compact_plan(bits, t, sync=False)
        n_dev = cu[rows:rows + 1]
        side: Dict[int, Optional[Tensor]] = {i_b: None for i_b in branches}
        if 0 < enc._ltt_freeze_layer:
            flat = hidden.view(rows * t, c.hidden_size)
            for i_b in branches:
                w, bias = self._maps[f"{i_b}_0"].get(dtype)
                s0 = ops.gemm(flat, w, bias, L.AG_EPI_BIAS_GELU, dtype)
                s0 = self._side[(t, f"{i_b}_0")].forward(s0.view(rows, t, c.s_attn_hidden_size), rows, 1, bits, False, dtype)
                side[i_b] = ops.gather_rows(s0, src, n, dtype, rows_dev=n_dev)
        hidden = ops.gather_rows(hidden, src, n, dtype, rows_dev=n_dev)
        for i_ly in range(1, enc.num_layers):
            hidden = self._bb[t][i_ly].forward_packed(hidden, cu, rows, n, dtype, rows_dev=n_dev)
            if i_ly >= enc._ltt_freeze_layer:
                continue
            for i_b in branches:
                key = f"{i_b}_{i_ly}"
                w, bias = self._maps[key].get(dtype)
                s_new = ops.gemm(hidden, w, bias, L.AG_EPI_BIAS_GELU_ADD, dtype, resid=side[i_b], rows_per_seq=1, resid_share=1,
                                 rows_dev=n_dev)
                side[i_b] = self._side[(t, key)].forward_packed(s_new, cu, rows, n, dtype, rows_dev=n_dev)
        engine.note_packed_rows(hidden.device, n_dev)
        h_cls = ops.gather_rows(hidden, cu, rows, dtype).view(rows, 1, c.hidden_size)
        s_cls = [ops.gather_rows(side[i_b], cu, rows, dtype).view(rows, 1, c.s_attn_hidden_size) for i_b in branches]
        return h_cls, s_cls, rows

    def forward(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor],
                side_layer_branches: List[int]) -> Tuple[Tensor, List[Tensor]]:
        hidden, outs, _, _ = self.run(input_ids, attention_mask, token_type_ids, side_layer_branches)
        return hidden, outs


class LttBertSurrogate(nn.Module, ObservableModuleMixin, _BertHead):
    """reference :69-117; returns (side probs, backbone probs)."""

    def __init__(self, config: LttBertConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.bert = LttBertModel(config=config, num_side_branches=1)
        self.bert_pooler = VanillaBertPooler(hidden_size=config.hidden_size)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.act = nn.Softmax(dim=-1)
        self.bert_s_attn_pooler = VanillaBertPooler(hidden_size=config.s_attn_hidden_size)
        self.s_attn_dropout = nn.Dropout(config.hidden_dropout_prob)
        self.s_attn_classifier = nn.Linear(config.s_attn_hidden_size, config.num_labels)
        self.s_attn_act = nn.Softmax(dim=-1)

    def train(self, mode: bool = True) -> Self:
        super().train(mode)
        freeze_model_parameters(self, "bert.embeddings")
        freeze_model_parameters(self, "bert.encoder.layers")
        freeze_model_parameters(self, "bert_pooler")
        freeze_model_parameters(self, "classifier")
        return self

    def ltt_freeze_layers_until(self, layer_id: int) -> None:
        self.bert.encoder.ltt_freeze_layers_until(layer_id)

    def forward(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        if _ag.grad_mode(self):   # side probabilities carry the gradient; the frozen backbone's do not
            return _ag.surrogate_forward(self, input_ids, attention_mask)
        dtype = engine.get_precision()
        if self.om_is_observing():
            output, (srg_output,), _, rows = self.bert.run(input_ids, attention_mask, token_type_ids, [0])
            self.om_record_features(repr_cls=output, repr_srg=srg_output)
            t = input_ids.shape[1]
        else:   # only the CLS rows are consumed (:108-116): token-pruned path
            output, (srg_output,), rows = self.bert.run_cls(input_ids, attention_mask, token_type_ids, [0])
            t = 1
        logits = self._pool_classify(output, rows, t, self.bert_pooler, self.classifier, "cls", True, dtype)
        srg_logits = self._pool_classify(srg_output, rows, t, self.bert_s_attn_pooler, self.s_attn_classifier, "side", True, dtype)
        return srg_logits, logits


class _LttBertExplainerHead(nn.Module):
    """s_attn_attention_layers + s_attn_explainer (reference :133-156, :233-256)."""

    def _build_head(self, config: LttBertConfig) -> None:
        self.s_attn_attention_layers = nn.Sequential(*[
            VanillaBertLayer(config.attention_probs_dropout_prob, config.hidden_dropout_prob, config.s_attn_hidden_size,
                             config.s_attn_intermediate_size, config.layer_norm_eps, config.num_attention_heads,
                             repl_norm_1_ident=(i == 0), repl_norm_2_ident=False)
            for i in range(config.explainer_s_attn_num_layers)])
        self.s_attn_exp_dropout = nn.Dropout(config.hidden_dropout_prob)
        w = int(config.explainer_s_head_hidden_size)
        self.s_attn_explainer = nn.Sequential(nn.Linear(config.s_attn_hidden_size, w), nn.GELU(), nn.Linear(w, w), nn.GELU(),
                                              nn.Linear(w, config.num_labels))
        self._attn_packed: Dict[int, engine.PackedEncoder] = {}
        self._mlp_packed: Optional[List[engine.PackedLinear]] = None

    def _run_head(self, exp_output: Tensor, bits: Tensor, surrogate_grand, surrogate_null, dtype: int) -> Tensor:
        config = self.config
        rows, t, h = exp_output.shape
        if t not in self._attn_packed:
            self._attn_packed[t] = engine.PackedEncoder(list(self.s_attn_attention_layers), L.AG_MASK_BERT_ADD, t, h,
                                                        config.s_attn_intermediate_size, config.num_attention_heads,
                                                        config.layer_norm_eps)
        if self._mlp_packed is None:
            m = self.s_attn_explainer
            self._mlp_packed = [engine.PackedLinear([m[i].weight], [m[i].bias]) for i in (0, 2, 4)]
        o = self._attn_packed[t].forward(exp_output.contiguous(), rows, 1, bits, False, dtype) if len(self.s_attn_attention_layers) else exp_output
        xs = engine.linear_head(o, h, rows * t, self._mlp_packed[0], L.AG_EPI_BIAS_GELU, dtype)
        xs = engine.linear_head(xs, xs.shape[1], rows * t, self._mlp_packed[1], L.AG_EPI_BIAS_GELU, dtype)
        pred = engine.linear_head(xs, xs.shape[1], rows * t, self._mlp_packed[2], L.AG_EPI_BIAS_F32, dtype)
        pred = pred.view(rows, t, config.num_labels)
        return ops.shapley_normalize(pred, surrogate_grand, surrogate_null, normalize=bool(config.explainer_normalize))


class LttBertExplainer(_LttBertExplainerHead, ObservableModuleMixin, _BertHead):
    """reference :120-218; returns (phi [B,C,P], backbone probs [B,C])."""

    def __init__(self, config: LttBertConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.bert = LttBertModel(config=config, num_side_branches=1)
        self.bert_pooler = VanillaBertPooler(hidden_size=config.hidden_size)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.act = nn.Softmax(dim=-1)
        self._build_head(config)

    def ltt_freeze_layers_until(self, layer_id: int) -> None:
        self.bert.encoder.ltt_freeze_layers_until(layer_id)

    def train(self, mode: bool = True):
        super().train(mode)
        freeze_model_parameters(self, "bert.embeddings")
        freeze_model_parameters(self, "bert.encoder.layers")
        freeze_model_parameters(self, "bert_pooler")
        freeze_model_parameters(self, "classifier")
        return self

    def forward(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor],
                surrogate_grand: Tensor, surrogate_null: Tensor) -> Tuple[Tensor, Tensor]:
        if _ag.grad_mode(self):
            return _ag.explainer_forward(self, input_ids, attention_mask, surrogate_grand, surrogate_null)
        dtype = engine.get_precision()
        output, (exp_output,), bits, rows = self.bert.run(input_ids, attention_mask, token_type_ids, [0])
        self.om_record_features(repr_cls=output, repr_exp=exp_output)
        logits = self._pool_classify(output, rows, input_ids.shape[1], self.bert_pooler, self.classifier, "cls", True, dtype)
        return self._run_head(exp_output, bits, surrogate_grand, surrogate_null, dtype), logits


class LttBertFinal(_LttBertExplainerHead, ObservableModuleMixin, _BertHead):
    """reference :221-349: one backbone pass, ladders 0 (surrogate) and 1 (explainer)."""

    def __init__(self, config: LttBertConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.bert = LttBertModel(config=config, num_side_branches=2)
        self.bert_pooler = VanillaBertPooler(hidden_size=config.hidden_size)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.act = nn.Softmax(dim=-1)
        self.bert_s_attn_pooler = VanillaBertPooler(hidden_size=config.s_attn_hidden_size)
        self.s_attn_dropout = nn.Dropout(config.hidden_dropout_prob)
        self.s_attn_classifier = nn.Linear(config.s_attn_hidden_size, config.num_labels)
        self.s_attn_act = nn.Softmax(dim=-1)
        self.surrogate_null = nn.Parameter(torch.zeros((1, config.num_labels)), requires_grad=False)
        self._build_head(config)

    def train(self, mode: bool = True):
        super().train(mode)
        freeze_model_parameters(self, "bert.embeddings")
        freeze_model_parameters(self, "bert.encoder.layers")
        freeze_model_parameters(self, "bert_pooler")
        freeze_model_parameters(self, "classifier")
        return self

    def forward(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        _no_autograd(self)
        dtype = engine.get_precision()
        t = input_ids.shape[1]
        if self.config.explainer_normalize:
            output, (srg_output, exp_output), bits, rows = self.bert.run(input_ids, attention_mask, token_type_ids, [0, 1])
            self.om_record_features(repr_cls=output, repr_srg=srg_output, repr_exp=exp_output)
            surrogate_grand = self._pool_classify(srg_output, rows, t, self.bert_s_attn_pooler, self.s_attn_classifier, "side", True, dtype)
            surrogate_null = self.surrogate_null
        else:
            output, (exp_output,), bits, rows = self.bert.run(input_ids, attention_mask, token_type_ids, [1])
            self.om_record_features(repr_cls=output, repr_exp=exp_output)
            surrogate_grand = surrogate_null = None
        logits = self._pool_classify(output, rows, t, self.bert_pooler, self.classifier, "cls", True, dtype)
        return logits, self._run_head(exp_output, bits, surrogate_grand, surrogate_null, dtype)
