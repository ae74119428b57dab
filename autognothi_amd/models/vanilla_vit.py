"""MI355X-native mirror of reference ``models/vanilla_vit.py``: same class names, constructor
arguments, ``forward`` signatures and ``state_dict`` keys, but ``forward``
drives hand-written HIP kernels through the C ABI instead of stock torch ops.

The nn.Modules below are parameter containers (their own ``forward`` is never called); the math of
reference models/vanilla_vit.py:207-214 (model), :242-253 (embeddings), :364-377 (layer),
:436-465 (masked attention), :51-56 (classifier head), :102-130 (explainer) lives in
``autognothi_amd/csrc``.  There is no CPU path: calling ``forward`` with CPU tensors raises.

Extension over the reference API: ``forward(x, attention_mask)`` accepts ``x`` with B rows and a
mask with R = B*K rows (row r uses input r // K, the reference's ``[b0 s0, b0 s1, b1 s0, ...]``
order, models/shapley.py:24) — the K masked copies then share embeddings and layer-0 LN/QKV
instead of being materialised K times as in scripts/train_explainer.py:159-163.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import pydantic
import torch
from torch import Tensor, nn
from typing_extensions import Self

from .. import _lib as L
from .. import autograd as _ag
from .. import engine, ops
from ..utils.nnmodel import ObservableModuleMixin, freeze_model_parameters


class VanillaViTConfig(pydantic.BaseModel):
    """differs from `transformers.ViTModel` (reference models/vanilla_vit.py:14-30)"""

    attention_probs_dropout_prob: float
    explainer_attn_num_layers: int
    explainer_head_hidden_size: int
    explainer_normalize: bool
    hidden_dropout_prob: float
    hidden_size: int
    intermediate_size: int
    layer_norm_eps: float
    num_attention_heads: int
    num_hidden_layers: int
    num_labels: int
    img_channels: int
    img_px_size: int
    img_patch_size: int


# ------------------------------------------------------------------------- parameter containers
class VanillaViTSelfAttention(nn.Module):
    def __init__(self, attention_probs_dropout_prob: float, num_attention_heads: int, hidden_size: int):
        super().__init__()
        assert hidden_size % num_attention_heads == 0
        self.num_attention_heads = num_attention_heads
        self.attention_head_size = hidden_size // num_attention_heads
        self.query = nn.Linear(hidden_size, hidden_size, bias=True)
        self.key = nn.Linear(hidden_size, hidden_size, bias=True)
        self.value = nn.Linear(hidden_size, hidden_size, bias=True)
        self.dropout = nn.Dropout(attention_probs_dropout_prob)


class VanillaViTSelfOutput(nn.Module):
    def __init__(self, hidden_dropout_prob: float, hidden_size: int):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.dropout = nn.Dropout(hidden_dropout_prob)


class VanillaViTAttention(nn.Module):
    def __init__(self, attention_probs_dropout_prob: float, hidden_dropout_prob: float, hidden_size: int,
                 num_attention_heads: int):
        super().__init__()
        self.self = VanillaViTSelfAttention(attention_probs_dropout_prob, num_attention_heads, hidden_size)
        self.output = VanillaViTSelfOutput(hidden_dropout_prob, hidden_size)


class VanillaViTIntermediate(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int):
        super().__init__()
        self.dense = nn.Linear(hidden_size, intermediate_size)
        self.intermediate_act_fn = nn.GELU()


class VanillaViTOutput(nn.Module):
    def __init__(self, hidden_dropout_prob: float, hidden_size: int, intermediate_size: int):
        super().__init__()
        self.dense = nn.Linear(intermediate_size, hidden_size)
        self.dropout = nn.Dropout(hidden_dropout_prob)


class VanillaViTLayer(nn.Module):
    def __init__(self, attention_probs_dropout_prob: float, hidden_dropout_prob: float, hidden_size: int,
                 intermediate_size: int, layer_norm_eps: float, num_attention_heads: int,
                 repl_norm_1_ident: bool, repl_norm_2_ident: bool):
        super().__init__()
        self.attention = VanillaViTAttention(attention_probs_dropout_prob, hidden_dropout_prob, hidden_size,
                                             num_attention_heads)
        self.intermediate = VanillaViTIntermediate(hidden_size, intermediate_size)
        self.output = VanillaViTOutput(hidden_dropout_prob, hidden_size, intermediate_size)
        self.layernorm_before = nn.LayerNorm(hidden_size, eps=layer_norm_eps) if not repl_norm_1_ident else nn.Identity()
        self.layernorm_after = nn.LayerNorm(hidden_size, eps=layer_norm_eps) if not repl_norm_2_ident else nn.Identity()
        if repl_norm_2_ident:
            raise NotImplementedError("layernorm_after = Identity is never used by the reference recipes")


class VanillaViTEncoder(nn.Module):
    def __init__(self, attention_probs_dropout_prob: float, hidden_dropout_prob: float, hidden_size: int,
                 intermediate_size: int, layer_norm_eps: float, num_hidden_layers: int, num_attention_heads: int):
        super().__init__()
        self.layers = nn.ModuleList([
            VanillaViTLayer(attention_probs_dropout_prob, hidden_dropout_prob, hidden_size, intermediate_size,
                            layer_norm_eps, num_attention_heads, False, False)
            for _ in range(num_hidden_layers)])


class VanillaViTPatchEmbeddings(nn.Module):
    def __init__(self, hidden_size: int, img_channels: int, img_px_size: int, img_patch_size: int):
        super().__init__()
        self.img_channels, self.img_px_size, self.img_patch_size = img_channels, img_px_size, img_patch_size
        self.n_patches = (img_px_size // img_patch_size) ** 2
        self.projection = nn.Conv2d(img_channels, hidden_size, kernel_size=img_patch_size, stride=img_patch_size)


class VanillaViTEmbeddings(nn.Module):
    def __init__(self, hidden_dropout_prob: float, hidden_size: int, img_px_size: int, img_patch_size: int,
                 img_channels: int):
        super().__init__()
        self.cls_token = nn.Parameter(torch.randn(1, 1, hidden_size))
        self.patch_embeddings = VanillaViTPatchEmbeddings(hidden_size, img_channels, img_px_size, img_patch_size)
        self.position_embeddings = nn.Parameter(torch.randn(1, self.patch_embeddings.n_patches + 1, hidden_size))
        self.dropout = nn.Dropout(hidden_dropout_prob)


class VanillaViTModel(nn.Module):
    """Backbone (reference models/vanilla_vit.py:181-214).  ``run`` is the HIP path."""

    def __init__(self, config: VanillaViTConfig):
        super().__init__()
        self.config = config
        self.embeddings = VanillaViTEmbeddings(config.hidden_dropout_prob, config.hidden_size, config.img_px_size,
                                               config.img_patch_size, config.img_channels)
        self.encoder = VanillaViTEncoder(config.attention_probs_dropout_prob, config.hidden_dropout_prob,
                                         config.hidden_size, config.intermediate_size, config.layer_norm_eps,
                                         config.num_hidden_layers, config.num_attention_heads)
        self.layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self._packed: Optional[engine.PackedEncoder] = None
        self._patch: Optional[engine.PackedLinear] = None

    @property
    def n_players(self) -> int:
        return (self.config.img_px_size // self.config.img_patch_size) ** 2

    def packed(self) -> engine.PackedEncoder:
        if self._packed is None:
            c = self.config
            self._packed = engine.PackedEncoder(self.encoder.layers, L.AG_MASK_VIT_MUL, self.n_players + 1, c.hidden_size,
                                                c.intermediate_size, c.num_attention_heads, c.layer_norm_eps)
            proj = self.embeddings.patch_embeddings.projection
            self._patch = engine.PackedLinear([proj.weight], [proj.bias])
        return self._packed

    def embed(self, pixel_values: Tensor, dtype: int) -> Tensor:
        """reference :242-253 -> fp32 h0 [B, T, H], computed once per distinct input."""
        c = self.config
        L.require_gpu(pixel_values)
        x = pixel_values.contiguous().float()
        b, ch, hh, ww = x.shape
        assert ch == c.img_channels and hh == c.img_px_size and ww == c.img_px_size  # reference :281-282
        self.packed()
        p, kdim = self.n_players, ch * c.img_patch_size ** 2
        cols = torch.empty((b * p, kdim), dtype=ops.storage_dtype(dtype), device=x.device)
        with L.on(x.device):
            L.check(L.lib().ag_vit_im2col(L.ptr(x), b, ch, hh, c.img_patch_size, L.ptr(cols), dtype, L.stream()))
            w, bias = self._patch.get(dtype)
            pe = ops.gemm(cols, w, bias, L.AG_EPI_BIAS_F32, dtype)
            h0 = torch.empty((b, p + 1, c.hidden_size), dtype=torch.float32, device=x.device)
            cls = self.embeddings.cls_token.detach().float().contiguous()
            pos = self.embeddings.position_embeddings.detach().float().contiguous()
            L.check(L.lib().ag_vit_assemble(L.ptr(pe), L.ptr(cls), L.ptr(pos), b, p, c.hidden_size, L.ptr(h0), L.stream()))
        return h0

    def run(self, pixel_values: Tensor, attention_mask: Tensor, cls_only: bool) -> Tuple[Tensor, int]:
        """-> (pre-final-LN hidden fp32 [R,T,H], R).  With cls_only only hidden[:,0] is defined."""
        dtype = engine.get_precision()
        bits = engine.to_mask_bits(attention_mask, self.n_players)
        rows, b = bits.shape[0], pixel_values.shape[0]
        if rows % b != 0:
            raise ValueError(f"mask rows ({rows}) must be a multiple of input rows ({b})")
        h0 = self.embed(pixel_values, dtype)
        return self.packed().forward(h0, rows, rows // b, bits, cls_only, dtype), rows

    def final_norm(self, hidden: Tensor, rows: int, cls_only: bool, dtype: int, want_f32: bool):
        t, h = self.n_players + 1, self.config.hidden_size
        g, bta = self.layernorm.weight.detach().float(), self.layernorm.bias.detach().float()
        if cls_only:
            return ops.layernorm(hidden, g, bta, self.config.layer_norm_eps, dtype, rows=rows, ldx=t * h, want_f32=want_f32)
        return ops.layernorm(hidden, g, bta, self.config.layer_norm_eps, dtype, rows=rows * t, ldx=h, want_f32=want_f32)

    def forward(self, pixel_values: Tensor, attention_mask: Tensor) -> Tensor:
        """reference :207-214 -> LN_final(hidden) [R,T,H] in the storage dtype."""
        hidden, rows = self.run(pixel_values, attention_mask, cls_only=False)
        z, _ = self.final_norm(hidden, rows, False, engine.get_precision(), want_f32=False)
        return z.view(rows, self.n_players + 1, self.config.hidden_size)


def _no_autograd(module: nn.Module) -> None:
    """The Final modules (and the plain classifier) are never trained by the reference's pipelines: inference only.
    Surrogates and explainers route to autognothi_amd.autograd under grad mode instead of calling this."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise RuntimeError(
            f"{type(module).__name__}.forward: this module has no training path on the HIP kernels; call it under "
            "torch.no_grad() (surrogate / explainer modules are differentiable: autognothi_amd/autograd.py).")


class VanillaViTClassifier(nn.Module, ObservableModuleMixin):
    def __init__(self, config: VanillaViTConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.vit = VanillaViTModel(config)
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.act = nn.Softmax(dim=-1)
        self._head: Optional[engine.PackedLinear] = None

    def train(self, mode: bool = True):
        super().train(mode)
        freeze_model_parameters(self, "vit")
        freeze_model_parameters(self, "classifier")
        return self

    def forward(self, x: Tensor, attention_mask: Tensor) -> Tensor:
        """reference :51-56: softmax(Linear(LN_f(h)[:,0])) -> probabilities [R, C].  Only the CLS row
        of the last layer is consumed, so that layer's out-proj/MLP run on the CLS token only.
        Under grad mode (reference scripts/train_surrogate.py:143-147) the result carries a grad_fn (autograd.py)."""
        if _ag.grad_mode(self):
            return _ag.surrogate_forward(self, x, attention_mask)[0]
        dtype = engine.get_precision()
        observing = self.om_is_observing()
        hidden, rows = self.vit.run(x, attention_mask, cls_only=not observing)
        if observing:
            zs, _ = self.vit.final_norm(hidden, rows, False, dtype, want_f32=False)
            t, h = self.vit.n_players + 1, self.config.hidden_size
            self.om_record_features(repr_cls=zs.view(rows, t, h))
            z_cls, lda = zs, t * h
        else:
            z_cls, _ = self.vit.final_norm(hidden, rows, True, dtype, want_f32=False)
            lda = self.config.hidden_size
        if self._head is None:
            self._head = engine.PackedLinear([self.classifier.weight], [self.classifier.bias])
        logits = engine.linear_head(z_cls, lda, rows, self._head, L.AG_EPI_BIAS_F32, dtype)
        return ops.softmax_rows(logits)


class VanillaViTSurrogate(VanillaViTClassifier):
    def train(self, mode: bool = True):
        nn.Module.train(self, mode)
        return self


class _ExplainerHead(nn.Module):
    """explainer_attn + explainer_mlp (+ optional normalisation) shared by vanilla / duo / froyo."""

    def _build_head(self, config) -> None:
        self.explainer_attn = nn.ModuleList([
            VanillaViTLayer(config.attention_probs_dropout_prob, config.hidden_dropout_prob, config.hidden_size,
                            config.intermediate_size, config.layer_norm_eps, config.num_attention_heads,
                            repl_norm_1_ident=(i == 0), repl_norm_2_ident=False)
            for i in range(config.explainer_attn_num_layers)])
        w = int(config.explainer_head_hidden_size)
        self.explainer_mlp = nn.Sequential(
            nn.LayerNorm(config.hidden_size), nn.Linear(config.hidden_size, w), nn.GELU(),
            nn.Linear(w, w), nn.GELU(), nn.Linear(w, config.num_labels))
        self._attn_packed: Optional[engine.PackedEncoder] = None
        self._mlp_packed: Optional[List[engine.PackedLinear]] = None

    def _run_head(self, z: Tensor, bits: Tensor, rows: int, surrogate_grand, surrogate_null, config, dtype: int) -> Tensor:
        """z = LN_final(backbone) [rows,T,H] (storage dtype) -> phi [rows, C, P]  (reference :120-129)."""
        t, h = z.shape[1], z.shape[2]
        if self._attn_packed is None:
            self._attn_packed = engine.PackedEncoder(self.explainer_attn, L.AG_MASK_VIT_MUL, t, h, config.intermediate_size,
                                                     config.num_attention_heads, config.layer_norm_eps)
            m = self.explainer_mlp
            self._mlp_packed = [engine.PackedLinear([m[i].weight], [m[i].bias]) for i in (1, 3, 5)]
        o = self._attn_packed.forward(z.contiguous(), rows, 1, bits, False, dtype) if len(self.explainer_attn) else z
        ln = self.explainer_mlp[0]
        xs, _ = ops.layernorm(o, ln.weight.detach().float(), ln.bias.detach().float(), ln.eps, dtype, rows=rows * t, ldx=h)
        xs = engine.linear_head(xs, h, rows * t, self._mlp_packed[0], L.AG_EPI_BIAS_GELU, dtype)
        xs = engine.linear_head(xs, xs.shape[1], rows * t, self._mlp_packed[1], L.AG_EPI_BIAS_GELU, dtype)
        pred = engine.linear_head(xs, xs.shape[1], rows * t, self._mlp_packed[2], L.AG_EPI_BIAS_F32, dtype)
        pred = pred.view(rows, t, config.num_labels)
        return ops.shapley_normalize(pred, surrogate_grand, surrogate_null, normalize=bool(config.explainer_normalize))


class VanillaViTExplainer(_ExplainerHead, ObservableModuleMixin):
    def __init__(self, config: VanillaViTConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.vit = VanillaViTModel(config)
        self._build_head(config)

    def forward(self, pixel_values: Tensor, attention_mask: Tensor, surrogate_grand: Tensor,
                surrogate_null: Tensor) -> Tensor:
        """reference :102-130 -> phi [B, C, P]; differentiable under grad mode (scripts/train_explainer.py:183-197)."""
        if _ag.grad_mode(self):
            return _ag.explainer_forward(self, pixel_values, attention_mask, surrogate_grand, surrogate_null)[0]
        dtype = engine.get_precision()
        bits = engine.to_mask_bits(attention_mask, self.vit.n_players)
        z = self.vit(pixel_values, bits)
        self.om_record_features(repr_exp=z)
        return self._run_head(z, bits, z.shape[0], surrogate_grand, surrogate_null, self.config, dtype)


class VanillaViTFinal(nn.Module, ObservableModuleMixin):
    def __init__(self, config: VanillaViTConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.classifier = VanillaViTClassifier(config)
        self.surrogate = VanillaViTSurrogate(config)
        self.surrogate_null = nn.Parameter(torch.zeros((1, config.num_labels)), requires_grad=False)
        self.explainer = VanillaViTExplainer(config)

    def forward(self, pixel_values: Tensor, attention_mask: Tensor) -> Tuple[Tensor, Tensor]:
        """reference :147-169."""
        logits = self.classifier(pixel_values, attention_mask)
        om_repr_cls = self.classifier.om_take_observations()
        if self.config.explainer_normalize:
            surrogate_grand = self.surrogate(pixel_values, attention_mask)
            om_repr_srg = self.surrogate.om_take_observations()
        else:
            surrogate_grand = None
            om_repr_srg = {}
        explainer = self.explainer(pixel_values, attention_mask, surrogate_grand, self.surrogate_null)
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
