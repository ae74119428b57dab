"""MI355X-native mirror of reference ``models/ltt_vit.py`` (LTT = ladder / side-network tuning): same class
names, constructor arguments, ``forward`` signatures and ``state_dict`` keys; ``forward`` drives the HIP
kernels through the C ABI.

A frozen ViT backbone is tapped after every layer i by a narrow side branch (reference :423-436):

    side = side + gelu(Linear_{H->h}(hidden_i))        # ag_gemm, epilogue AG_EPI_BIAS_GELU_ADD (one kernel)
    side = ViTLayer_h(side, mask)                      # ag_encoder_forward on the h-wide layer (heads = 12 -> d = h/12)

with h = ``s_attn_hidden_size`` (96 in the shipped config, i.e. 8-wide heads: ``attn_valu_kernel<T, 8>``).
``Surrogate`` and ``Explainer`` carry one branch, ``Final`` two (surrogate = 0, explainer = 1) off ONE backbone
pass (reference :244-249).  The nn.Modules are parameter containers; there is no CPU path.

As in ``vanilla_vit.py`` the mask may have R = B*K rows for B inputs: embeddings and the backbone's layer-0
LN/QKV are then shared by the K masks of an input.
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
from .vanilla_vit import VanillaViTConfig, VanillaViTEmbeddings, VanillaViTLayer, VanillaViTModel, _no_autograd


class LttViTConfig(pydantic.BaseModel):
    """reference models/ltt_vit.py:14-52"""

    attention_probs_dropout_prob: float
    explainer_s_attn_num_layers: int  # side head
    explainer_s_head_hidden_size: int  # side head
    explainer_normalize: bool  # side head
    hidden_dropout_prob: float
    hidden_size: int
    intermediate_size: int
    layer_norm_eps: float
    num_attention_heads: int
    num_hidden_layers: int
    num_labels: int
    s_attn_hidden_size: int  # side attention
    s_attn_intermediate_size: int  # side attention
    img_channels: int
    img_px_size: int
    img_patch_size: int

    def into(self) -> VanillaViTConfig:
        return VanillaViTConfig(
            attention_probs_dropout_prob=self.attention_probs_dropout_prob,
            explainer_attn_num_layers=self.explainer_s_attn_num_layers,
            explainer_head_hidden_size=self.explainer_s_head_hidden_size,
            explainer_normalize=self.explainer_normalize,
            hidden_dropout_prob=self.hidden_dropout_prob,
            hidden_size=self.hidden_size,
            intermediate_size=self.intermediate_size,
            layer_norm_eps=self.layer_norm_eps,
            num_attention_heads=self.num_attention_heads,
            num_hidden_layers=self.num_hidden_layers,
            num_labels=self.num_labels,
            img_channels=self.img_channels,
            img_px_size=self.img_px_size,
            img_patch_size=self.img_patch_size,
        )


class LttViTMultiEncoder(nn.Module):
    """Backbone layers + per-(branch, layer) ladder maps and side layers (reference :343-440)."""

    def __init__(self, attention_probs_dropout_prob: float, hidden_dropout_prob: float, hidden_size: int,
                 intermediate_size: int, layer_norm_eps: float, num_hidden_layers: int, num_attention_heads: int,
                 num_side_branches: int, s_attn_hidden_size: int, s_attn_intermediate_size: int):
        super().__init__()
        self.num_layers = num_hidden_layers
        self.num_branches = num_side_branches
        self.layers = nn.ModuleList([
            VanillaViTLayer(attention_probs_dropout_prob, hidden_dropout_prob, hidden_size, intermediate_size,
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
                side[f"{i_b}_{i_ly}"] = VanillaViTLayer(attention_probs_dropout_prob, hidden_dropout_prob, s_attn_hidden_size,
                                                        s_attn_intermediate_size, layer_norm_eps, num_attention_heads,
                                                        repl_norm_1_ident=False, repl_norm_2_ident=False)
        self.s_attn_layers = nn.ModuleDict(side)
        self._ltt_freeze_layer = num_hidden_layers  # training hack of the reference (:399-405)

    def ltt_freeze_layers_until(self, layer_id: int) -> None:
        self._ltt_freeze_layer = max(1, min(len(self.layers), layer_id))


class LttViTModel(nn.Module):
    """reference :290-340.  ``run`` is the HIP path."""

    def __init__(self, config: LttViTConfig, num_side_branches: int):
        super().__init__()
        self.config = config
        self.num_side_branches = num_side_branches
        self.embeddings = VanillaViTEmbeddings(config.hidden_dropout_prob, config.hidden_size, config.img_px_size,
                                               config.img_patch_size, config.img_channels)
        self.encoder = LttViTMultiEncoder(config.attention_probs_dropout_prob, config.hidden_dropout_prob,
                                          config.hidden_size, config.intermediate_size, config.layer_norm_eps,
                                          config.num_hidden_layers, config.num_attention_heads, num_side_branches,
                                          config.s_attn_hidden_size, config.s_attn_intermediate_size)
        self.layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.s_attn_layernorm = nn.ModuleList([nn.LayerNorm(config.s_attn_hidden_size, eps=config.layer_norm_eps)
                                               for _ in range(num_side_branches)])
        self._bb: Optional[List[engine.PackedEncoder]] = None
        self._side: Dict[str, engine.PackedEncoder] = {}
        self._maps: Dict[str, engine.PackedLinear] = {}
        self._patch: Optional[engine.PackedLinear] = None

    @property
    def n_players(self) -> int:
        return (self.config.img_px_size // self.config.img_patch_size) ** 2

    def packed(self) -> None:
        """called by the borrowed ``VanillaViTModel.embed``; packs the ladder as well."""
        self._pack()

    def _pack(self) -> None:
        if self._bb is not None:
            return
        c, t = self.config, self.n_players + 1
        # one single-layer encoder per backbone layer: the ladder taps the stream after each of them
        self._bb = [engine.PackedEncoder([ly], L.AG_MASK_VIT_MUL, t, c.hidden_size, c.intermediate_size,
                                         c.num_attention_heads, c.layer_norm_eps) for ly in self.encoder.layers]
        for key, ly in self.encoder.s_attn_layers.items():
            self._side[key] = engine.PackedEncoder([ly], L.AG_MASK_VIT_MUL, t, c.s_attn_hidden_size, c.s_attn_intermediate_size,
                                                   c.num_attention_heads, c.layer_norm_eps)
        for key, lin in self.encoder.s_attn_maps.items():
            self._maps[key] = engine.PackedLinear([lin.weight], [lin.bias])
        proj = self.embeddings.patch_embeddings.projection
        self._patch = engine.PackedLinear([proj.weight], [proj.bias])

    def embed(self, pixel_values: Tensor, dtype: int) -> Tensor:
        """the vanilla embeddings (reference :296-302); that code only touches .embeddings / .config / ._patch"""
        return VanillaViTModel.embed(self, pixel_values, dtype)  # type: ignore[arg-type]

    def run(self, pixel_values: Tensor, attention_mask: Tensor, side_layer_branches: Sequence[int]) -> Tuple[Tensor, List[Tensor], Tensor, int]:
        """-> (LN_f(hidden) [R,T,H], [LN_b(side_b) [R,T,h] for b in sorted(side_layer_branches)], mask bits, R);
        all in the storage dtype (reference :323-340, :407-440)."""
        dtype = engine.get_precision()
        c, t = self.config, self.n_players + 1
        bits = engine.to_mask_bits(attention_mask, self.n_players)
        rows, b = bits.shape[0], pixel_values.shape[0]
        if rows % b != 0:
            raise ValueError(f"mask rows ({rows}) must be a multiple of input rows ({b})")
        self._pack()
        hidden = self.embed(pixel_values, dtype)
        branches = sorted(set(int(x) for x in side_layer_branches))
        for i_b in branches:
            if not 0 <= i_b < self.num_side_branches:
                raise ValueError(f"side branch {i_b} out of range (model has {self.num_side_branches})")
        side: Dict[int, Optional[Tensor]] = {i_b: None for i_b in branches}
        enc = self.encoder
        # LayerNorm-fold row statistics of the stream, produced by layer i's fc2 epilogue for layer i+1's QKV (one call per layer)
        chain = [ops.new_row_stats(rows * t, self.config.hidden_size, hidden.device), False, True]
        for i_ly in range(enc.num_layers):
            share = rows // b if i_ly == 0 else 1
            chain[2] = i_ly + 1 < enc.num_layers
            hidden = self._bb[i_ly].forward(hidden, rows, share, bits, False, dtype, chain=chain)
            if i_ly >= enc._ltt_freeze_layer:
                continue
            flat = hidden.view(rows * t, c.hidden_size)
            for i_b in branches:
                key = f"{i_b}_{i_ly}"
                w, bias = self._maps[key].get(dtype)
                if side[i_b] is None:   # the reference starts from the scalar 0.0
                    s_new = ops.gemm(flat, w, bias, L.AG_EPI_BIAS_GELU, dtype)
                else:
                    s_new = ops.gemm(flat, w, bias, L.AG_EPI_BIAS_GELU_ADD, dtype, resid=side[i_b].view(rows * t, -1),
                                     rows_per_seq=t, resid_share=1)
                side[i_b] = self._side[key].forward(s_new.view(rows, t, c.s_attn_hidden_size), rows, 1, bits, False, dtype)
        z, _ = ops.layernorm(hidden, self.layernorm.weight.detach().float(), self.layernorm.bias.detach().float(),
                             c.layer_norm_eps, dtype, rows=rows * t, ldx=c.hidden_size)
        outs = []
        for i_b in branches:
            ln = self.s_attn_layernorm[i_b]
            s, _ = ops.layernorm(side[i_b], ln.weight.detach().float(), ln.bias.detach().float(), c.layer_norm_eps, dtype,
                                 rows=rows * t, ldx=c.s_attn_hidden_size)
            outs.append(s.view(rows, t, c.s_attn_hidden_size))
        return z.view(rows, t, c.hidden_size), outs, bits, rows

    def forward(self, pixel_values: Tensor, attention_mask: Tensor, side_layer_branches: List[int]) -> Tuple[Tensor, List[Tensor]]:
        z, outs, _, _ = self.run(pixel_values, attention_mask, side_layer_branches)
        return z, outs


def _cls_probs(z: Tensor, lin: engine.PackedLinear, dtype: int) -> Tensor:
    """softmax(Linear(z[:, 0, :])) -> fp32 [R, C]; the CLS rows are read in place (row stride T*H)."""
    rows, t, h = z.shape
    return ops.softmax_rows(engine.linear_head(z, t * h, rows, lin, L.AG_EPI_BIAS_F32, dtype))


class LttViTSurrogate(nn.Module, ObservableModuleMixin):
    """reference :55-94: backbone classifier + a side-branch classifier; returns (side probs, backbone probs)."""

    def __init__(self, config: LttViTConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.vit = LttViTModel(config, num_side_branches=1)
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.act = nn.Softmax(dim=-1)
        self.s_attn_classifier = nn.Linear(config.s_attn_hidden_size, config.num_labels)
        self.s_attn_act = nn.Softmax(dim=-1)
        self._heads: Optional[Tuple[engine.PackedLinear, engine.PackedLinear]] = None

    def train(self, mode: bool = True):
        super().train(mode)
        freeze_model_parameters(self, "vit.embeddings")
        freeze_model_parameters(self, "vit.encoder.layers")
        freeze_model_parameters(self, "vit.layernorm")
        freeze_model_parameters(self, "classifier")
        return self

    def ltt_freeze_layers_until(self, layer_id: int) -> None:
        self.vit.encoder.ltt_freeze_layers_until(layer_id)

    def forward(self, pixel_values: Tensor, attention_mask: Tensor) -> Tuple[Tensor, Tensor]:
        if _ag.grad_mode(self):   # side probabilities carry the gradient; the frozen backbone's do not
            return _ag.surrogate_forward(self, pixel_values, attention_mask)
        dtype = engine.get_precision()
        output, (srg_output,), _, _ = self.vit.run(pixel_values, attention_mask, [0])
        self.om_record_features(repr_cls=output, repr_srg=srg_output)
        if self._heads is None:
            self._heads = (engine.PackedLinear([self.classifier.weight], [self.classifier.bias]),
                           engine.PackedLinear([self.s_attn_classifier.weight], [self.s_attn_classifier.bias]))
        return _cls_probs(srg_output, self._heads[1], dtype), _cls_probs(output, self._heads[0], dtype)


class _LttExplainerHead(nn.Module):
    """s_explainer_attn + s_explainer_mlp (reference :107-130, :203-224)."""

    def _build_head(self, config: LttViTConfig) -> None:
        self.s_explainer_attn = nn.ModuleList([
            VanillaViTLayer(config.attention_probs_dropout_prob, config.hidden_dropout_prob, config.s_attn_hidden_size,
                            config.s_attn_intermediate_size, config.layer_norm_eps, config.num_attention_heads,
                            repl_norm_1_ident=(i == 0), repl_norm_2_ident=False)
            for i in range(config.explainer_s_attn_num_layers)])
        w = int(config.explainer_s_head_hidden_size)
        self.s_explainer_mlp = nn.Sequential(
            nn.LayerNorm(config.s_attn_hidden_size), nn.Linear(config.s_attn_hidden_size, w), nn.GELU(),
            nn.Linear(w, w), nn.GELU(), nn.Linear(w, config.num_labels))
        self._attn_packed: Optional[engine.PackedEncoder] = None
        self._mlp_packed: Optional[List[engine.PackedLinear]] = None

    def _run_head(self, exp_output: Tensor, bits: Tensor, surrogate_grand, surrogate_null, dtype: int) -> Tensor:
        """LN_1(side_1) [rows,T,h] -> phi [rows, C, P]  (reference :169-181)."""
        config = self.config
        rows, t, h = exp_output.shape
        if self._attn_packed is None:
            self._attn_packed = engine.PackedEncoder(self.s_explainer_attn, L.AG_MASK_VIT_MUL, t, h, config.s_attn_intermediate_size,
                                                     config.num_attention_heads, config.layer_norm_eps)
            m = self.s_explainer_mlp
            self._mlp_packed = [engine.PackedLinear([m[i].weight], [m[i].bias]) for i in (1, 3, 5)]
        o = self._attn_packed.forward(exp_output.contiguous(), rows, 1, bits, False, dtype) if len(self.s_explainer_attn) else exp_output
        ln = self.s_explainer_mlp[0]
        xs, _ = ops.layernorm(o, ln.weight.detach().float(), ln.bias.detach().float(), ln.eps, dtype, rows=rows * t, ldx=h)
        xs = engine.linear_head(xs, h, rows * t, self._mlp_packed[0], L.AG_EPI_BIAS_GELU, dtype)
        xs = engine.linear_head(xs, xs.shape[1], rows * t, self._mlp_packed[1], L.AG_EPI_BIAS_GELU, dtype)
        pred = engine.linear_head(xs, xs.shape[1], rows * t, self._mlp_packed[2], L.AG_EPI_BIAS_F32, dtype)
        pred = pred.view(rows, t, config.num_labels)
        return ops.shapley_normalize(pred, surrogate_grand, surrogate_null, normalize=bool(config.explainer_normalize))


class LttViTExplainer(_LttExplainerHead, ObservableModuleMixin):
    """reference :97-183; returns (phi [B,C,P], backbone probs [B,C])."""

    def __init__(self, config: LttViTConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.vit = LttViTModel(config, num_side_branches=1)
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.act = nn.Softmax(dim=-1)
        self._build_head(config)
        self._cls: Optional[engine.PackedLinear] = None

    def train(self, mode: bool = True):
        super().train(mode)
        freeze_model_parameters(self, "vit.embeddings")
        freeze_model_parameters(self, "vit.encoder.layers")
        freeze_model_parameters(self, "vit.layernorm")
        freeze_model_parameters(self, "classifier")
        return self

    def ltt_freeze_layers_until(self, layer_id: int) -> None:
        self.vit.encoder.ltt_freeze_layers_until(layer_id)

    def forward(self, pixel_values: Tensor, attention_mask: Tensor, surrogate_grand: Tensor,
                surrogate_null: Tensor) -> Tuple[Tensor, Tensor]:
        if _ag.grad_mode(self):
            return _ag.explainer_forward(self, pixel_values, attention_mask, surrogate_grand, surrogate_null)
        dtype = engine.get_precision()
        output, (exp_output,), bits, _ = self.vit.run(pixel_values, attention_mask, [0])
        self.om_record_features(repr_cls=output, repr_exp=exp_output)
        if self._cls is None:
            self._cls = engine.PackedLinear([self.classifier.weight], [self.classifier.bias])
        logits = _cls_probs(output, self._cls, dtype)
        return self._run_head(exp_output, bits, surrogate_grand, surrogate_null, dtype), logits


class LttViTFinal(_LttExplainerHead, ObservableModuleMixin):
    """reference :186-287: ONE backbone pass feeds both ladders (surrogate = branch 0, explainer = branch 1)."""

    def __init__(self, config: LttViTConfig):
        nn.Module.__init__(self)
        ObservableModuleMixin.__init__(self)
        self.config = config
        self.vit = LttViTModel(config, num_side_branches=2)
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.act = nn.Softmax(dim=-1)
        self.s_attn_classifier = nn.Linear(config.s_attn_hidden_size, config.num_labels)
        self.s_attn_act = nn.Softmax(dim=-1)
        self.surrogate_null = nn.Parameter(torch.zeros((1, config.num_labels)), requires_grad=False)
        self._build_head(config)
        self._heads: Optional[Tuple[engine.PackedLinear, engine.PackedLinear]] = None

    def train(self, mode: bool = True) -> Self:
        super().train(mode)
        freeze_model_parameters(self, ...)
        return self

    def forward(self, pixel_values: Tensor, attention_mask: Tensor) -> Tuple[Tensor, Tensor]:
        _no_autograd(self)
        dtype = engine.get_precision()
        if self._heads is None:
            self._heads = (engine.PackedLinear([self.classifier.weight], [self.classifier.bias]),
                           engine.PackedLinear([self.s_attn_classifier.weight], [self.s_attn_classifier.bias]))
        if self.config.explainer_normalize:
            output, (srg_output, exp_output), bits, _ = self.vit.run(pixel_values, attention_mask, [0, 1])
            self.om_record_features(repr_cls=output, repr_srg=srg_output, repr_exp=exp_output)
            surrogate_grand = _cls_probs(srg_output, self._heads[1], dtype)
            surrogate_null = self.surrogate_null
        else:
            output, (exp_output,), bits, _ = self.vit.run(pixel_values, attention_mask, [1])
            self.om_record_features(repr_cls=output, repr_exp=exp_output)
            surrogate_grand = surrogate_null = None
        logits = _cls_probs(output, self._heads[0], dtype)
        return logits, self._run_head(exp_output, bits, surrogate_grand, surrogate_null, dtype)
