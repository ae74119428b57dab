"""MI355X-native mirror of reference ``models/vanilla_bert.py`` (same class names, signatures and
state-dict keys); ``forward`` drives the HIP kernels.  Math restated from reference
models/vanilla_bert.py:307-325 (embeddings), :410-427 + :556-560 + :600-604 (post-LN block),
:503-537 (additive extended mask), :61-77 + :615-619 (pooler + head), :123-162 (explainer).

Same extension as the ViT mirror: ``input_ids`` may hold B rows while the mask holds R = B*K rows.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import pydantic
import torch
from torch import Tensor, nn
from typing_extensions import Self

from .. import _lib as L
from .. import engine, ops
from ..utils.nnmodel import ObservableModuleMixin, freeze_model_parameters
from .. import autograd as _ag
from .vanilla_vit import _no_autograd


class VanillaBertConfig(pydantic.BaseModel):
    """equiv. `transformers.BertModel` (reference models/vanilla_bert.py:16-37)"""

    attention_probs_dropout_prob: float
    explainer_attn_num_layers: int
    explainer_head_hidden_size: int
    explainer_normalize: bool
    hidden_dropout_prob: float
    hidden_size: int
    intermediate_size: int
    layer_norm_eps: float
    max_position_embeddings: int
    num_attention_heads: int
    num_hidden_layers: int
    num_labels: int
    pad_token_id: int
    type_vocab_size: int
    vocab_size: int

    @property
    def is_decoder(self) -> bool:
        return False


# ------------------------------------------------------------------------- parameter containers
class VanillaBertSelfAttention(nn.Module):
    def __init__(self, attention_probs_dropout_prob: float, hidden_size: int, num_attention_heads: int):
        super().__init__()
        if hidden_size % num_attention_heads != 0:
            raise ValueError(f"The hidden size ({hidden_size}) is not a multiple of the number of attention "
                             f"heads ({num_attention_heads})")
        self.num_attention_heads = num_attention_heads
        self.attention_head_size = hidden_size // num_attention_heads
        self.query = nn.Linear(hidden_size, hidden_size)
        self.key = nn.Linear(hidden_size, hidden_size)
        self.value = nn.Linear(hidden_size, hidden_size)
        self.dropout = nn.Dropout(attention_probs_dropout_prob)


class VanillaBertSelfOutput(nn.Module):
    def __init__(self, hidden_dropout_prob: float, hidden_size: int, layer_norm_eps: float, repl_norm_ident: bool):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.LayerNorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps) if not repl_norm_ident else nn.Identity()
        self.dropout = nn.Dropout(hidden_dropout_prob)


class VanillaBertAttention(nn.Module):
    def __init__(self, attention_probs_dropout_prob: float, hidden_dropout_prob: float, hidden_size: int,
                 layer_norm_eps: float, num_attention_heads: int, repl_norm_ident: bool):
        super().__init__()
        self.self = VanillaBertSelfAttention(attention_probs_dropout_prob, hidden_size, num_attention_heads)
        self.output = VanillaBertSelfOutput(hidden_dropout_prob, hidden_size, layer_norm_eps, repl_norm_ident)


class VanillaBertIntermediate(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int):
        super().__init__()
        self.dense = nn.Linear(hidden_size, intermediate_size)
        self.intermediate_act_fn = nn.GELU()


class VanillaBertOutput(nn.Module):
    def __init__(self, hidden_dropout_prob: float, hidden_size: int, intermediate_size: int, layer_norm_eps: float,
                 repl_norm_ident: bool):
        super().__init__()
        self.dense = nn.Linear(intermediate_size, hidden_size)
        if repl_norm_ident:
            raise NotImplementedError("output.LayerNorm = Identity is never used by the reference recipes")
        self.LayerNorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        self.dropout = nn.Dropout(hidden_dropout_prob)


class VanillaBertLayer(nn.Module):
    def __init__(self, attention_probs_dropout_prob: float, hidden_dropout_prob: float, hidden_size: int,
                 intermediate_size: int, layer_norm_eps: float, num_attention_heads: int,
                 repl_norm_1_ident: bool, repl_norm_2_ident: bool):
        super().__init__()
        self.attention = VanillaBertAttention(attention_probs_dropout_prob, hidden_dropout_prob, hidden_size,
                                              layer_norm_eps, num_attention_heads, repl_norm_1_ident)
        self.intermediate = VanillaBertIntermediate(hidden_size, intermediate_size)
        self.output = VanillaBertOutput(hidden_dropout_prob, hidden_size, intermediate_size, layer_norm_eps,
                                        repl_norm_2_ident)


class VanillaBertEncoder(nn.Module):
    def __init__(self, attention_probs_dropout_prob: float, hidden_dropout_prob: float, hidden_size: int,
                 intermediate_size: int, layer_norm_eps: float, num_attention_heads: int, num_hidden_layers: int):
        super().__init__()
        self.layers = nn.ModuleList([
            VanillaBertLayer(attention_probs_dropout_prob, hidden_dropout_prob, hidden_size, intermediate_size,
                             layer_norm_eps, num_attention_heads, False, False)
            for _ in range(num_hidden_layers)])


class VanillaBertEmbeddings(nn.Module):
    def __init__(self, hidden_dropout_prob: float, hidden_size: int, layer_norm_eps: float,
                 max_position_embeddings: int, pad_token_id: int, type_vocab_size: int, vocab_size: int):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab_size, hidden_size, padding_idx=pad_token_id)
        self.position_embeddings = nn.Embedding(max_position_embeddings, hidden_size)
        self.token_type_embeddings = nn.Embedding(type_vocab_size, hidden_size)
        self.LayerNorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        self.dropout = nn.Dropout(hidden_dropout_prob)
        self.register_buffer("position_ids", torch.arange(max_position_embeddings).expand((1, -1)), persistent=False)


class VanillaBertPooler(nn.Module):
    def __init__(self, hidden_size: int):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.activation = nn.Tanh()


class VanillaBertModel(nn.Module):
    def __init__(self, config: VanillaBertConfig):
        super().__init__()
        self.config = config
        self.embeddings = VanillaBertEmbeddings(config.hidden_dropout_prob, config.hidden_size, config.layer_norm_eps,
                                                config.max_position_embeddings, config.pad_token_id,
                                                config.type_vocab_size, config.vocab_size)
        self.encoder = VanillaBertEncoder(config.attention_probs_dropout_prob, config.hidden_dropout_prob,
                                          config.hidden_size, config.intermediate_size, config.layer_norm_eps,
                                          config.num_attention_heads, config.num_hidden_layers)
        self._packed = {}

    def packed(self, t: int) -> engine.PackedEncoder:
        if t not in self._packed:
            c = self.config
            self._packed[t] = engine.PackedEncoder(self.encoder.layers, L.AG_MASK_BERT_ADD, t, c.hidden_size,
                                                   c.intermediate_size, c.num_attention_heads, c.layer_norm_eps)
        return self._packed[t]

    def embed(self, input_ids: Tensor, token_type_ids: Optional[Tensor], dtype: int) -> Tensor:
        """reference :307-325 -> fp32 h0 [B,T,H].  The recipes always pass token_type_ids == 0
        (recipes/vanilla_bert.py:289); anything else is rejected rather than silently ignored."""
        L.require_gpu(input_ids)
        c = self.config
        ids = input_ids.contiguous().to(torch.int64)
        b, t = ids.shape
        if t > c.max_position_embeddings:
            raise ValueError(f"sequence length {t} > max_position_embeddings {c.max_position_embeddings}")
        if token_type_ids is not None and bool((token_type_ids != 0).any()):
            raise NotImplementedError("token_type_ids != 0 is not on the reference path (recipes/vanilla_bert.py:289)")
        e = self.embeddings
        h0 = torch.empty((b, t, c.hidden_size), dtype=torch.float32, device=ids.device)
        word = e.word_embeddings.weight.detach().float().contiguous()
        type0 = e.token_type_embeddings.weight.detach().float()[0].contiguous()
        pos = e.position_embeddings.weight.detach().float().contiguous()
        with L.on(ids.device):
            L.check(L.lib().ag_bert_embed(L.ptr(ids), b, t, c.hidden_size, L.ptr(word), c.vocab_size, L.ptr(type0),
                                          L.ptr(pos), L.ptr(e.LayerNorm.weight.detach().float().contiguous()),
                                          L.ptr(e.LayerNorm.bias.detach().float().contiguous()), c.layer_norm_eps,
                                          L.ptr(h0), None, dtype, L.stream()))
        return h0

    def run(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor], cls_only: bool):
        dtype = engine.get_precision()
        t = input_ids.shape[1]
        bits = engine.to_mask_bits(attention_mask, t - 1)
        rows, b = bits.shape[0], input_ids.shape[0]
        if rows % b != 0:
            raise ValueError(f"mask rows ({rows}) must be a multiple of input rows ({b})")
        h0 = self.embed(input_ids, token_type_ids, dtype)
        return self.packed(t).forward(h0, rows, rows // b, bits, cls_only, dtype), rows, bits

    def forward(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor] = None) -> Tensor:
        hidden, _, _ = self.run(input_ids, attention_mask, token_type_ids, cls_only=False)
        return hidden


class _BertHead:
    """pooler + classifier on the CLS row (reference :73-76, :615-619)."""

    def _pool_classify(self, hidden: Tensor, rows: int, t: int, pooler: VanillaBertPooler, classifier: nn.Linear,
                       cache_name: str, act: bool, dtype: int) -> Tensor:
        cache = self.__dict__.setdefault("_head_cache", {})
        if cache_name not in cache:
            cache[cache_name] = (engine.PackedLinear([pooler.dense.weight], [pooler.dense.bias]),
                                 engine.PackedLinear([classifier.weight], [classifier.bias]))
        pl, cl = cache[cache_name]
        h = hidden.shape[-1]
        # CLS rows of the [R,T,H] stream read in place (row stride T*H)
        pooled = engine.linear_head(hidden, t * h, rows, pl, L.AG_EPI_BIAS_TANH, dtype)
        logits = engine.linear_head(pooled, h, rows, cl, L.AG_EPI_BIAS_F32, dtype)
        return ops.softmax_rows(logits) if act else logits


class VanillaBertClassifier(nn.Module, ObservableModuleMixin, _BertHead):
    def __init__(self, config: VanillaBertConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.bert = VanillaBertModel(config)
        self.bert_pooler = VanillaBertPooler(hidden_size=config.hidden_size)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.act = nn.Softmax(dim=-1)

    def train(self, mode: bool = True):
        super().train(mode)
        freeze_model_parameters(self, "bert")
        freeze_model_parameters(self, "bert_pooler")
        freeze_model_parameters(self, "classifier")
        return self

    def forward(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor] = None) -> Tensor:
        """reference :61-77 -> probabilities [R,C]; differentiable under grad mode (scripts/train_surrogate.py:143-147)."""
        if _ag.grad_mode(self):
            return _ag.surrogate_forward(self, input_ids, attention_mask)[0]
        dtype = engine.get_precision()
        observing = self.om_is_observing()
        hidden, rows, _ = self.bert.run(input_ids, attention_mask, token_type_ids, cls_only=not observing)
        if observing:
            self.om_record_features(repr_cls=hidden)
        return self._pool_classify(hidden, rows, input_ids.shape[1], self.bert_pooler, self.classifier, "cls", True, dtype)


class VanillaBertSurrogate(VanillaBertClassifier):
    def train(self, mode: bool = True):
        nn.Module.train(self, mode)
        return self


class _BertExplainerHead(nn.Module):
    def _build_head(self, config) -> None:
        self.explainer_attn = nn.ModuleList([
            VanillaBertLayer(config.attention_probs_dropout_prob, config.hidden_dropout_prob, config.hidden_size,
                             config.intermediate_size, config.layer_norm_eps, config.num_attention_heads,
                             repl_norm_1_ident=(i == 0), repl_norm_2_ident=False)
            for i in range(config.explainer_attn_num_layers)])
        self.explainer_dropout = nn.Dropout(config.hidden_dropout_prob)
        w = int(config.explainer_head_hidden_size)
        self.explainer_mlp = nn.Sequential(nn.Linear(config.hidden_size, w), nn.GELU(), nn.Linear(w, w), nn.GELU(),
                                           nn.Linear(w, config.num_labels))
        self._attn_packed = {}
        self._mlp_packed: Optional[List[engine.PackedLinear]] = None

    def _run_head(self, z: Tensor, bits: Tensor, rows: int, surrogate_grand, surrogate_null, config, dtype: int) -> Tensor:
        """z = backbone output fp32 [rows,T,H] -> phi [rows,C,P]  (reference :147-161; no leading LN)."""
        t, h = z.shape[1], z.shape[2]
        if t not in self._attn_packed:
            self._attn_packed[t] = engine.PackedEncoder(self.explainer_attn, L.AG_MASK_BERT_ADD, t, h,
                                                        config.intermediate_size, config.num_attention_heads,
                                                        config.layer_norm_eps)
        if self._mlp_packed is None:
            m = self.explainer_mlp
            self._mlp_packed = [engine.PackedLinear([m[i].weight], [m[i].bias]) for i in (0, 2, 4)]
        o = self._attn_packed[t].forward(z.contiguous(), rows, 1, bits, False, dtype) if len(self.explainer_attn) else z
        xs = engine.linear_head(o, h, rows * t, self._mlp_packed[0], L.AG_EPI_BIAS_GELU, dtype)
        xs = engine.linear_head(xs, xs.shape[1], rows * t, self._mlp_packed[1], L.AG_EPI_BIAS_GELU, dtype)
        pred = engine.linear_head(xs, xs.shape[1], rows * t, self._mlp_packed[2], L.AG_EPI_BIAS_F32, dtype)
        pred = pred.view(rows, t, config.num_labels)
        return ops.shapley_normalize(pred, surrogate_grand, surrogate_null, normalize=bool(config.explainer_normalize))


class VanillaBertExplainer(_BertExplainerHead, ObservableModuleMixin):
    def __init__(self, config: VanillaBertConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.bert = VanillaBertModel(config)
        self._build_head(config)

    def forward(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor],
                surrogate_grand: Tensor, surrogate_null: Tensor) -> Tensor:
        """reference :123-162 -> phi [B,C,P]; differentiable under grad mode (scripts/train_explainer.py:183-197)."""
        if _ag.grad_mode(self):
            return _ag.explainer_forward(self, input_ids, attention_mask, surrogate_grand, surrogate_null)[0]
        dtype = engine.get_precision()
        hidden, rows, bits = self.bert.run(input_ids, attention_mask, token_type_ids, cls_only=False)
        self.om_record_features(repr_exp=hidden)
        return self._run_head(hidden, bits, rows, surrogate_grand, surrogate_null, self.config, dtype)


class VanillaBertFinal(nn.Module, ObservableModuleMixin):
    def __init__(self, config: VanillaBertConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.classifier = VanillaBertClassifier(config)
        self.surrogate = VanillaBertSurrogate(config)
        self.surrogate_null = nn.Parameter(torch.zeros((1, config.num_labels)), requires_grad=False)
        self.explainer = VanillaBertExplainer(config)

    def forward(self, input_ids: Tensor, attention_mask: Tensor, token_type_ids: Optional[Tensor] = None):
        """reference :181-215."""
        logits = self.classifier(input_ids, attention_mask, token_type_ids)
        om_repr_cls = self.classifier.om_take_observations()
        if self.config.explainer_normalize:
            surrogate_grand = self.surrogate(input_ids, attention_mask, token_type_ids)
            om_repr_srg = self.surrogate.om_take_observations()
        else:
            surrogate_grand, om_repr_srg = None, {}
        explainer = self.explainer(input_ids, attention_mask, token_type_ids, surrogate_grand, self.surrogate_null)
        om_repr_exp = self.explainer.om_take_observations()
        self.om_record_features(repr_cls=om_repr_cls.get("repr_cls", None), repr_srg=om_repr_srg.get("repr_srg", None),
                                repr_exp=om_repr_exp.get("repr_exp", None))
        return logits, explainer

    def train(self, mode: bool = True) -> Self:
        super().train(mode)
        freeze_model_parameters(self, "classifier")
        return self

    def om_retain_observations(self, flag: bool = True) -> None:
        ObservableModuleMixin.om_retain_observations(self, flag)
        self.classifier.om_retain_observations(flag)
        self.surrogate.om_retain_observations(flag)
        self.explainer.om_retain_observations(flag)
