"""MIRROR pre-training model on hand-written gfx950 kernels — host-side mirror of the reference's
`models/mirror.py` (same class names, constructor kwargs, forward signatures, 15-tuple order and
state-dict keys), with every arithmetic step dispatched to libmirror_hip.so through
`mirror_amd.functional`.  nn.Linear / nn.LayerNorm / nn.Conv2d objects are used as *parameter
containers only* (identical keys, shapes and default initialisation); their forward() is never called.

Build-only extensions (default = reference behaviour):
  * `rna_num_heads` kwarg (the reference hard-wires 12, models/mirror.py:392/:798-811, which rejects D=256/512);
  * `noise=` dict for the four random draws of MIRROR.forward (reproducible parity runs);
  * `precision` attribute / `mirror_amd.set_precision()` : "fp32" | "bf16" | "bf16_pinv32" | None (follow autocast).
"""
from __future__ import annotations

import logging
import math
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import nn

from .. import functional as Fn
from ..functional import FP32, BF16, POLICIES, Precision
from .._lib import ACT_NONE, MirrorHipError

_logger = logging.getLogger(__name__)
f32 = torch.float32

_DEFAULT_PRECISION: Optional[str] = None

# program order of the two encoder branches in MIRROR.forward (see there): 1 = WSI encoder launches first (default)
_RNA_LATE = True      # (test hook)
# 1 (default) = the alignment / style heads run on the RNA branch's helper stream, 0 = on the caller's stream (A/B switch)
_HEADS_SIDE = True      # (test hook)
_LM_MASKED = True      # (test hook, round 5) the landmark-row LayerNorm + to_qkv node also under a key-padding mask (BASELINE config 4)
_OWN_NOISE = True      # (test hook, round 5) the step's four random draws as one launch on the dropout stream instead of torch's generator
# (measured and removed: the four noise draws + the prototype renorm on the RNA stream cost 0.5 - 1 % of the step)


def set_precision(name: Optional[str]) -> None:
    """Process-wide default policy for modules whose `.precision` is None."""
    global _DEFAULT_PRECISION
    if name is not None and name not in POLICIES:
        raise ValueError(f"unknown precision {name!r}; choose from {sorted(POLICIES)}")
    _DEFAULT_PRECISION = name


def resolve_precision(pref: Optional[str]) -> Precision:
    name = pref or _DEFAULT_PRECISION
    if name is not None:
        return POLICIES[name]
    if torch.is_autocast_enabled():  # train_mirror.py:759-761 wraps the step in torch.autocast
        return BF16
    return FP32


def _trunc_normal_(t: torch.Tensor, std: float) -> None:
    nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


class _Mlp(nn.Module):
    """Parameter layout of timm.layers.Mlp: fc1 -> act -> drop1 -> norm -> fc2 -> drop2 ([3P], models/mirror.py:217-224)."""

    def __init__(self, in_features, hidden_features, out_features, norm_eps: Optional[float], drop: float):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.norm = nn.LayerNorm(hidden_features, eps=norm_eps) if norm_eps is not None else nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = drop

    def forward(self, x, prec: Precision, extra_bias: Optional[torch.Tensor] = None, out_dtype=None, residual=None):
        """residual: the pre-norm block's stream; `residual + drop2(fc2(...))` then comes back as one fused op."""
        h = Fn.gelu(Fn.linear(x, self.fc1.weight, self.fc1.bias, prec=prec))
        h = Fn.dropout(h, self.drop, self.training)
        if isinstance(self.norm, nn.LayerNorm):
            h = Fn.layer_norm(h, self.norm.weight, self.norm.bias, self.norm.eps, out_dtype=prec.act)
        b2 = self.fc2.bias if extra_bias is None else Fn.add(self.fc2.bias, extra_bias.reshape(-1), f32)
        y = Fn.linear(h, self.fc2.weight, b2, prec=prec, out_dtype=out_dtype)
        if residual is not None:
            return Fn.dropout_add(residual, y, self.drop, self.training)
        return Fn.dropout(y, self.drop, self.training)


# ===========================================
#  Transformer for Transcriptomics Data (RNA)
# ===========================================
class Attention(nn.Module):
    """models/mirror.py:50-102: on a 2-D [B, D] input the SDPA runs over the heads axis."""

    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False, proj_drop: float = 0.0):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = proj_drop

    def forward(self, x, prec: Precision, residual=None):
        qkv = Fn.linear(x, self.qkv.weight, self.qkv.bias, prec=prec)
        o = Fn.HeadAttnFn.apply(qkv, self.num_heads)
        y = Fn.linear(o, self.proj.weight, self.proj.bias, prec=prec)
        if residual is not None:
            return Fn.dropout_add(residual, y, self.proj_drop, self.training)
        return Fn.dropout(y, self.proj_drop, self.training)


class Block(nn.Module):
    """models/mirror.py:105-152 with LayerScale / DropPath at their identity settings."""

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = False,
                 proj_drop: float = 0.0, norm_eps: float = 1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=norm_eps)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, proj_drop=proj_drop)
        self.norm2 = nn.LayerNorm(dim, eps=norm_eps)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio), dim, None, proj_drop)

    def forward(self, x, prec: Precision):
        y = Fn.rna_block(x, self, prec, self.training)          # one fused call per direction (csrc/rna_block.hip) when it fits
        if y is not None:
            return y
        h = Fn.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, out_dtype=prec.act)
        x = self.attn(h, prec, residual=x)                          # x + drop(proj(...)); x feeds exactly norm1 and this add
        h = Fn.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, out_dtype=prec.act)
        return self.mlp(h, prec, residual=x)


class TransFormer(nn.Module):
    """models/mirror.py:155-289 (embedding = Mlp(G -> 2D -> D, LayerNorm(2D)), learnt gene_embed, blocks, norm)."""

    def __init__(self, input_dim: int, embed_dim: int = 768, depth: int = 2, num_heads: int = 12,
                 mlp_ratio: float = 4.0, qkv_bias: bool = True, gene_embed: str = "learn",
                 embed_drop_rate: float = 0.0, pos_drop_rate: float = 0.0, proj_drop_rate: float = 0.0,
                 norm_eps: float = 1e-6, **unused):
        super().__init__()
        assert gene_embed in ("", "none", "learn")
        self.num_features = self.head_hidden_size = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.norm_eps = norm_eps
        self.embedding = _Mlp(input_dim, embed_dim * 2, embed_dim, norm_eps, embed_drop_rate)
        if not gene_embed or gene_embed == "none":
            self.gene_embed = None
        else:
            self.gene_embed = nn.Parameter(torch.randn(1, embed_dim) * 0.02)
            _trunc_normal_(self.gene_embed, std=0.02)
        self.pos_drop_rate = pos_drop_rate
        self.blocks = nn.Sequential(*[
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, proj_drop_rate, norm_eps) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=norm_eps)
        self.precision: Optional[str] = None

    def forward(self, x):
        prec = resolve_precision(self.precision)
        x = self.embedding(x, prec, extra_bias=self.gene_embed, out_dtype=f32)
        x = Fn.dropout(x, self.pos_drop_rate, self.training)
        for blk in self.blocks:
            x = blk(x, prec)
        return Fn.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps, out_dtype=f32)


# ===========================================
#  TransMIL for Histopathology Data (WSI)
# ===========================================
class NystromAttention(nn.Module):
    """Parameter layout of [3P] nystrom_attention.NystromAttention as configured at models/mirror.py:299-309."""

    def __init__(self, dim, dim_head=64, heads=8, num_landmarks=256, pinv_iterations=6, residual=True,
                 residual_conv_kernel=33, eps=1e-8, dropout=0.0):
        super().__init__()
        assert residual, "the reference always enables the residual conv (models/mirror.py:306)"
        inner = heads * dim_head
        self.heads, self.num_landmarks, self.pinv_iterations = heads, num_landmarks, pinv_iterations
        self.eps, self.drop = eps, dropout
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))
        ks = residual_conv_kernel
        self.res_conv = nn.Conv2d(heads, heads, (ks, 1), padding=(ks // 2, 0), groups=heads, bias=False)


class TransLayer(nn.Module):
    """models/mirror.py:295-314: x + NystromAttention(LayerNorm(x)); dim_head = D//8, m = D//2 landmarks."""

    def __init__(self, dim: int = 512):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.attn = NystromAttention(dim=dim, dim_head=dim // 8, heads=8, num_landmarks=dim // 2,
                                     pinv_iterations=6, residual=True, dropout=0.1)

    def forward(self, x, prec: Precision, mask=None):
        """mask: optional [B, n] bool key-padding mask (True = real token) or an Fn.KeyMask, the `mask` argument of [3P]
        NystromAttention.forward that the reference never passes (models/mirror.py:312); BASELINE config 4 uses it for
        variable-length slides.  It is front-padded with False like the sequence."""
        a = self.attn
        n, m = x.shape[1], a.num_landmarks
        pad = (m - n % m) % m
        l = math.ceil(n / m)  # noqa: E741
        lm = kmask = mrow = None
        if mask is not None:
            km = mask if isinstance(mask, Fn.KeyMask) else Fn.KeyMask(mask.to(x.device))
            if tuple(km.shape) != tuple(x.shape[:2]):
                raise ValueError(f"key-padding mask must be {tuple(x.shape[:2])}, got {tuple(km.shape)}")
            kmask = km.plan(pad, l)     # (row mask, landmark valid flag, l / valid count): one launch, shared by layers of one geometry
            mrow = kmask[0]
        if (mask is None or _LM_MASKED) and Fn.layer_norm_landmarks_ok(x, n, pad, l, prec):
            # the norm also leaves the landmark means of its output behind the padded sequence; to_qkv is linear and bias-free, so the
            # q | k landmarks of [3P] NystromAttention are the same projection's result for those extra rows (Fn.NormQkvLmFn).  With a
            # key-padding mask (round 5) the same launch zeroes the masked rows and keeps them out of the landmark sums: no row-scale
            # pass over the norm's output, no landmark pass over q | k (NystromCoreFn scales the sums by l / valid count)
            lsc = None if kmask is None else kmask[2]
            qkv, lm = Fn.NormQkvLmFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps, n, pad, l, a.to_qkv.weight, prec, mrow, lsc)
            if kmask is not None:
                kmask = (kmask[0], kmask[1], None)      # the landmark rows are masked means already
        else:
            xp = Fn.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps, pad=pad, out_dtype=prec.act,
                               q8_key=Fn.fp8_site_key(a.to_qkv.weight, prec) if prec.fp8_fwd else None)
            if mask is not None:
                xp = Fn.RowScaleFn.apply(xp, mrow)           # to_qkv has no bias: zero rows in, zero q / k / v rows out
            qkv = Fn.linear(xp, a.to_qkv.weight, None, prec=prec, defer_from=2 * a.to_qkv.weight.shape[1])
        core = Fn.NystromCoreFn.apply(qkv, a.res_conv.weight, a.heads, l, a.pinv_iterations, prec, kmask,
                                      Fn.fp8_site_key(a.to_out[0].weight, prec) if prec.fp8_fwd else None, lm)
        # to_out(...)[:, -n:], its Dropout and the residual add: one launch when the shapes allow (Fn.to_out_dropout_add).
        # x feeds exactly self.norm and this add
        return Fn.to_out_dropout_add(x, core, a.to_out[0].weight, a.to_out[0].bias, pad, n, a.drop, self.training, prec)


class PPEG(nn.Module):
    """models/mirror.py:317-331 (three depthwise convs + identity, merged into one kernel)."""

    def __init__(self, dim: int = 512):
        super().__init__()
        self.proj = nn.Conv2d(dim, dim, 7, 1, 7 // 2, groups=dim)
        self.proj1 = nn.Conv2d(dim, dim, 5, 1, 5 // 2, groups=dim)
        self.proj2 = nn.Conv2d(dim, dim, 3, 1, 3 // 2, groups=dim)

    def forward(self, x, H, W):  # noqa: N803
        assert H == W and x.shape[1] == 1 + H * W
        return Fn.PPEGFn.apply(x, self.proj.weight, self.proj.bias, self.proj1.weight, self.proj1.bias,
                               self.proj2.weight, self.proj2.bias, H)


class FeatureTransMIL(nn.Module):
    """models/mirror.py:334-380 (downstream encoder; forward returns the normalised cls token)."""

    def __init__(self, input_dim: int = 1024, embed_dim: int = 512):
        super().__init__()
        self.input_dim, self.embed_dim = input_dim, embed_dim
        self.pos_layer = PPEG(dim=embed_dim)
        self._fc1 = nn.Sequential(nn.Linear(input_dim, embed_dim), nn.ReLU())
        self.cls_token = nn.Parameter(torch.randn(1, 1, embed_dim))
        self.layer1 = TransLayer(dim=embed_dim)
        self.layer2 = TransLayer(dim=embed_dim)
        self.norm = nn.LayerNorm(embed_dim)
        self.precision: Optional[str] = None

    def _encode(self, h, keep_rows: Optional[int], mask: Optional[torch.Tensor] = None):
        """mask: optional [B, N] bool, True = real patch (BASELINE config 4; the reference has no such argument).  The
        sequence [cls, x_0..x_{N-1}, x_0..x_{add-1}] carries [True, mask, mask[:, :add]]."""
        prec = resolve_precision(self.precision)
        if not h.is_cuda:
            raise MirrorHipError("mirror_amd models run on MI355X only (no CPU fallback): move the inputs to the GPU")
        n_tok = h.shape[1]
        side = int(np.ceil(np.sqrt(n_tok)))
        add = side * side - n_tok
        seq = Fn.probe_point(Fn.fc1_seq(h, self._fc1[0].weight, self._fc1[0].bias, self.cls_token, add, prec), "wsi_fc1_out")
        smask = None
        if mask is not None:
            smask = Fn.KeyMask(mask.to(h.device, torch.bool), lead=1, wrap=add)
        seq = self.layer1(seq, prec, smask)
        seq = self.pos_layer(seq, side, side)
        seq = self.layer2(seq, prec, smask)
        rows = seq.shape[1] - add if keep_rows is None else keep_rows
        # the pre-training encoder output also feeds the retention decoder's bf16 projection: the norm writes that copy itself
        return Fn.layer_norm(seq, self.norm.weight, self.norm.bias, self.norm.eps, rows=rows, out_dtype=f32,
                             bf16_copy=(keep_rows is None and prec.act == torch.bfloat16 and torch.is_grad_enabled()))

    def forward(self, h):
        return self._encode(h, keep_rows=1)[:, 0]


# ===========================================
#  Hybrid (pre-training) encoders
# ===========================================
class TransFormerHybrid(TransFormer):
    """models/mirror.py:386-569."""

    def __init__(self, input_dim: int, embed_dim: int = 768, depth: int = 2, num_heads: int = 12,
                 mlp_ratio: float = 4.0, qkv_bias: bool = True, gene_embed: str = "learn",
                 embed_drop_rate: float = 0.0, pos_drop_rate: float = 0.0, proj_drop_rate: float = 0.0,
                 norm_eps: float = 1e-6, retention_decoder_depth: int = 1, **unused):
        super().__init__(input_dim, embed_dim, depth, num_heads, mlp_ratio, qkv_bias, gene_embed,
                         embed_drop_rate, pos_drop_rate, proj_drop_rate, norm_eps)
        self.alignment_head = nn.Linear(embed_dim, embed_dim)
        self.retention_embed = nn.Linear(embed_dim, embed_dim)
        self.mask_token = nn.Parameter(torch.zeros(1, 1))
        self.retention_gene_embed = nn.Parameter(torch.randn(1, embed_dim) * 0.02)
        self.retention_blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, proj_drop_rate, norm_eps)
            for _ in range(retention_decoder_depth)])
        self.retention_norm = nn.LayerNorm(embed_dim, eps=norm_eps)
        self.retention_head = nn.Linear(embed_dim, embed_dim)
        nn.init.normal_(self.mask_token, std=0.02)
        _trunc_normal_(self.retention_gene_embed, std=0.02)
        with torch.no_grad():  # models/mirror.py:503-508
            for layer_id, layer in enumerate(self.retention_blocks):
                layer.attn.proj.weight.div_(math.sqrt(2.0 * (layer_id + 1)))
                layer.mlp.fc2.weight.div_(math.sqrt(2.0 * (layer_id + 1)))

    def forward_encoder(self, x):
        if not x.is_cuda:
            raise MirrorHipError("mirror_amd models run on MI355X only (no CPU fallback): move the inputs to the GPU")
        return TransFormer.forward(self, x)

    def forward_alignment_head(self, x):
        prec = resolve_precision(self.precision)
        y = Fn.L2NormRowFn.apply(x, 1e-12, f32)
        return Fn.linear(y, self.alignment_head.weight, self.alignment_head.bias, prec=prec, out_dtype=f32)

    def _draw_mask(self, x, mask_ratio: float, noise: Optional[torch.Tensor] = None):
        B, N = x.shape  # noqa: N806
        len_keep = int(N * (1 - mask_ratio))
        if noise is None:
            noise = torch.rand(B, N, device=x.device)
        return Fn.rank_mask(noise, len_keep)

    def random_masking(self, x, mask_ratio: float, noise: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """models/mirror.py:510-533: (x with the masked CHANNELS replaced by the scalar mask token, mask [B, D]).  The
        retention head below fuses the same select with the `+ retention_gene_embed` that follows it."""
        mask = self._draw_mask(x, mask_ratio, noise)
        zero_pos = torch.zeros(x.shape[1], device=x.device, dtype=f32)
        return Fn.MaskApplyFn.apply(x, mask, self.mask_token, zero_pos, 0, True), mask

    def forward_retention_head(self, x, mask_ratio: float, noise: Optional[torch.Tensor] = None):
        prec = resolve_precision(self.precision)
        r = Fn.linear(x, self.retention_embed.weight, self.retention_embed.bias, prec=prec, out_dtype=f32)
        mask = self._draw_mask(r, mask_ratio, noise)
        r = Fn.MaskApplyFn.apply(r, mask, self.mask_token, self.retention_gene_embed, 0, True)
        for blk in self.retention_blocks:
            r = blk(r, prec)
        r = Fn.layer_norm(r, self.retention_norm.weight, self.retention_norm.bias, self.retention_norm.eps,
                          out_dtype=prec.act)
        r = Fn.linear(r, self.retention_head.weight, self.retention_head.bias, prec=prec, out_dtype=f32)
        return r, mask

    def forward_decoders(self, x, mask_ratio: float, noise: Optional[torch.Tensor] = None):
        alignment_x = self.forward_alignment_head(x)
        retention_x, mask = self.forward_retention_head(x, mask_ratio, noise)
        return alignment_x, retention_x, mask

    def forward(self, x, mask_ratio: float = 0.75):
        x = self.forward_encoder(x)
        alignment_x, retention_x, mask = self.forward_decoders(x, mask_ratio)
        return alignment_x, retention_x, x, mask


class FeatureTransMILHybrid(FeatureTransMIL):
    """models/mirror.py:575-714."""

    def __init__(self, input_dim: int = 1024, embed_dim: int = 512, num_tokens: int = 2048,
                 retention_decoder_depth: int = 1):
        super().__init__(input_dim, embed_dim)
        self.num_tokens = num_tokens
        self.alignment_head = nn.Linear(embed_dim, embed_dim)
        self.retention_embed = nn.Linear(embed_dim, embed_dim)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.retention_gene_embed = nn.Parameter(torch.randn(1, num_tokens + 1, embed_dim) * 0.02)
        self.retention_blocks = nn.ModuleList([TransLayer(dim=embed_dim) for _ in range(retention_decoder_depth)])
        self.retention_norm = nn.LayerNorm(embed_dim)
        self.retention_head = nn.Linear(embed_dim, embed_dim)
        self.init_weights()

    def init_weights(self) -> None:  # models/mirror.py:609-622
        nn.init.normal_(self.mask_token, std=0.02)
        nn.init.normal_(self.cls_token, std=0.02)
        _trunc_normal_(self.retention_gene_embed, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

    def forward_encoder(self, h, mask: Optional[torch.Tensor] = None):
        return self._encode(h, keep_rows=None, mask=mask)

    def forward_alignment_head(self, h):
        prec = resolve_precision(self.precision)
        y = Fn.L2NormRowFn.apply(h, 1e-12, f32)         # only the cls row is consumed (models/mirror.py:684)
        return Fn.linear(y, self.alignment_head.weight, self.alignment_head.bias, prec=prec, out_dtype=f32)

    def _draw_mask(self, h, mask_ratio: float, noise: Optional[torch.Tensor] = None):
        B, N = h.shape[0], h.shape[1]  # noqa: N806
        len_keep = int(N * (1 - mask_ratio))
        if noise is None:
            noise = torch.rand(B, N, device=h.device)
        return Fn.rank_mask(noise, len_keep)

    def random_masking(self, h, mask_ratio: float, noise: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """models/mirror.py:624-649: (h [B, N, C] with the masked TOKENS replaced by mask_token, mask [B, N]).  The
        retention head below fuses the same select with the cls concat and the `+ retention_gene_embed` that follow it."""
        mask = self._draw_mask(h, mask_ratio, noise)
        zero_pos = torch.zeros(h.shape[1] * h.shape[2], device=h.device, dtype=f32)
        return Fn.MaskApplyFn.apply(h, mask, self.mask_token, zero_pos, 0, False), mask

    def forward_retention_head(self, h, mask_ratio: float, noise: Optional[torch.Tensor] = None,
                               mask: Optional[torch.Tensor] = None, key_padding_mask: Optional[torch.Tensor] = None,
                               target: Optional[torch.Tensor] = None):
        """`mask`: a mask already drawn for this batch (MIRROR.forward ranks the noise on a side stream up front).
        `key_padding_mask` [B, N] bool: attention mask of the decoder layers (config 4; cls is always attended)."""
        prec = resolve_precision(self.precision)
        if h.shape[1] != self.num_tokens + 1:
            raise ValueError(f"wsi_num_tokens={self.num_tokens} but the batch has {h.shape[1] - 1} tokens")
        # activation dtype out (what autocast gives the reference); MaskApplyFn restarts the f32 residual stream
        if mask is None:
            mask = self._draw_mask(h[:, 1:], mask_ratio, noise)
        # retention_embed, the mask-token select and `+ retention_gene_embed` (models/mirror.py:690-693): one launch when possible
        r = Fn.embed_mask_pos(h, self.retention_embed.weight, self.retention_embed.bias, mask, self.mask_token,
                              self.retention_gene_embed, 1, prec)
        kp = None
        if key_padding_mask is not None:
            kp = Fn.KeyMask(key_padding_mask.to(h.device, torch.bool), lead=1)
        for blk in self.retention_blocks:
            r = blk(r, prec, kp)
        r = Fn.layer_norm(r, self.retention_norm.weight, self.retention_norm.bias, self.retention_norm.eps,
                          out_dtype=prec.act)
        # retention_head(...)[:, 1:]: the cls row is sliced away, so it is never computed.  When the caller names the retention
        # target (MIRROR.forward does), the same launch accumulates the masked squared error MIRRORLoss needs (Fn.head_sqerr)
        r = Fn.head_sqerr(r, self.retention_head.weight, self.retention_head.bias, 1, r.shape[1] - 1, prec, target, mask)
        return r, mask

    def forward_decoders(self, h, mask_ratio: float, noise: Optional[torch.Tensor] = None):
        alignment_h = self.forward_alignment_head(h)
        retention_h, mask = self.forward_retention_head(h, mask_ratio, noise)
        return alignment_h, retention_h, mask

    def forward(self, h, mask_ratio: float = 0.75):
        h = self.forward_encoder(h)
        alignment_h, retention_h, mask = self.forward_decoders(h, mask_ratio)
        return alignment_h, retention_h, h[:, 1:, :], mask


# ===========================================
#  MIRROR for Pre-training
# ===========================================
class MIRROR(nn.Module):
    """models/mirror.py:720-915."""

    def __init__(self, wsi_embed_dim: int, rna_embed_dim: int, embed_dim: int, wsi_num_tokens: int = 2048,
                 wsi_retention_decoder_depth: int = 1, rna_encoder_depth: int = 2, rna_gene_embed: str = "learn",
                 rna_mlp_ratio: float = 2.572, rna_pos_drop_rate: float = 0.0, rna_proj_drop_rate: float = 0.1,
                 rna_attn_drop_rate: float = 0.0, rna_drop_path_rate: float = 0.0, rna_norm_layer=None,
                 rna_act_layer=None, rna_retention_decoder_depth: int = 1,
                 init_logit_scale: float = float(np.log(1 / 0.07)), style_mlp_hidden_dim: int = 512,
                 style_mlp_out_dim: int = 256, style_norm_layer=None, style_act_layer=None,
                 style_latent_dim: int = 128, num_prototypes: int = 3000, rna_num_heads: int = 12):
        super().__init__()
        for nm, v, ok in (("rna_norm_layer", rna_norm_layer, (None, "layernorm")), ("rna_act_layer", rna_act_layer, (None, "gelu")),
                          ("style_norm_layer", style_norm_layer, (None,)), ("style_act_layer", style_act_layer, (None, "gelu"))):
            if v not in ok:
                raise NotImplementedError(f"{nm}={v!r}: only {ok} have HIP kernels")
        if rna_attn_drop_rate or rna_drop_path_rate:
            raise NotImplementedError("attention dropout / drop-path are 0 in every reference config")
        self.wsi_embed_dim, self.rna_embed_dim, self.embed_dim = wsi_embed_dim, rna_embed_dim, embed_dim
        self.wsi_num_tokens = wsi_num_tokens
        self.style_latent_dim = style_latent_dim
        self.logit_scale = nn.Parameter(torch.ones([]) * init_logit_scale)
        self.wsi_encoder = FeatureTransMILHybrid(wsi_embed_dim, embed_dim, wsi_num_tokens, wsi_retention_decoder_depth)
        self.rna_encoder = TransFormerHybrid(
            input_dim=rna_embed_dim, embed_dim=embed_dim, depth=rna_encoder_depth, num_heads=rna_num_heads,
            mlp_ratio=rna_mlp_ratio, gene_embed=rna_gene_embed, pos_drop_rate=rna_pos_drop_rate,
            proj_drop_rate=rna_proj_drop_rate, retention_decoder_depth=rna_retention_decoder_depth)
        self.style_encoder_mlp = _Mlp(embed_dim, style_mlp_hidden_dim, style_mlp_out_dim, None, 0.0)
        self.style_mu = nn.Linear(style_mlp_out_dim, style_latent_dim)
        self.style_logstd = nn.Linear(style_mlp_out_dim, style_latent_dim)
        self.style_decoder = nn.Linear(style_latent_dim, embed_dim)
        self.prototypes = nn.Linear(embed_dim, num_prototypes, bias=False)
        nn.init.orthogonal_(self.prototypes.weight)
        self._precision: Optional[str] = None
        self._rna_drop_n: Dict[tuple, int] = {}      # dropout offsets the RNA branch consumes, per (shape, mode): see forward()

    @property
    def precision(self) -> Optional[str]:
        return self._precision

    @precision.setter
    def precision(self, name: Optional[str]) -> None:
        if name is not None and name not in POLICIES:
            raise ValueError(f"unknown precision {name!r}")
        self._precision = name
        self.wsi_encoder.precision = name
        self.rna_encoder.precision = name

    def reparameterize(self, mu, logstd, eps: Optional[torch.Tensor] = None):
        if eps is None:
            eps = torch.randn_like(mu)
        return Fn.ReparamFn.apply(mu, logstd, eps)[0]

    def _style_branch(self, emb, eps, prec):
        h = self.style_encoder_mlp(emb, prec)
        mu, logstd = Fn.linear_pair(h, self.style_mu.weight, self.style_mu.bias, self.style_logstd.weight, self.style_logstd.bias,
                                    prec=prec, out_dtype=f32)      # one node: h's gradient needs no add launch
        if eps is None:
            eps = torch.randn_like(mu)
        z, mu, logstd = Fn.ReparamFn.apply(mu, logstd, eps)      # mu / logstd pass through: their KL gradient comes back to this node
        z = Fn.linear(z, self.style_decoder.weight, self.style_decoder.bias, prec=prec)
        score = Fn.linear(z, self.prototypes.weight, None, prec=prec, out_dtype=f32)
        return score, mu, logstd

    def forward_style_clustering(self, wsi_emb, rna_emb, wsi_eps=None, rna_eps=None):
        prec = resolve_precision(self._precision)
        wsi_score, wsi_mu, wsi_logstd = self._style_branch(wsi_emb, wsi_eps, prec)
        rna_score, rna_mu, rna_logstd = self._style_branch(rna_emb, rna_eps, prec)
        return wsi_score, wsi_mu, wsi_logstd, rna_score, rna_mu, rna_logstd

    def rna_branch(self, rna_emb, rna_noise, rna_mask_ratio: float):
        """The RNA side of forward() up to the loss inputs: encoder output, alignment / retention heads, token mask."""
        Fn.probe("rna_enc_start")
        rna_emb = Fn.probe_point(self.rna_encoder.forward_encoder(rna_emb), "rna_enc_out")
        a, r, mask = self.rna_encoder.forward_decoders(rna_emb, mask_ratio=rna_mask_ratio, noise=rna_noise)
        Fn.probe("rna_branch_end")
        return rna_emb, a, r, mask

    def forward(self, wsi_emb, rna_emb, wsi_mask_ratio: float = 0.75, rna_mask_ratio: float = 0.75,
                noise: Optional[Dict[str, torch.Tensor]] = None, wsi_key_padding_mask: Optional[torch.Tensor] = None):
        """`noise` (build-only) pins the random draws; `wsi_key_padding_mask` (build-only, BASELINE config 4): [B, N] bool,
        True = real patch, False = padding of a slide shorter than N — handed to every Nystrom layer as the package's
        key-padding `mask`."""
        noise = dict(noise or {})
        if not wsi_emb.is_cuda:
            raise MirrorHipError("mirror_amd models run on MI355X only (no CPU fallback): move the inputs to the GPU")
        Fn._res_grads.clear()
        Fn.K.shared_chip = False   # a forward that raised between a chain fork and its join must not leave the hint set
        Fn._deferred.clear()       # hand-over slots of a backward that never completed must not meet this step's tensors
        Fn._pending_lm_merge.clear()
        # the reference draws: rand(B,N) -> rand(B,D) -> eps_wsi -> eps_rna (models/mirror.py:630, :516, :832-833);
        # draw them up front in that order so the two encoders can then run on different streams
        B, dev = wsi_emb.shape[0], wsi_emb.device
        # The RNA encoder is ~100 launch-bound [B, D] kernels (0.06 % of the FLOPs): it runs on a side stream
        # underneath the WSI encoder; autograd replays each backward node on its forward stream, so the RNA
        # backward overlaps the WSI backward as well.
        main = torch.cuda.current_stream()
        side = Fn._side_stream(dev, 1)
        wsi_in, rna_in = wsi_emb, rna_emb
        # the reference draws them in this order: rand(B,N) -> rand(B,D) -> eps_wsi -> eps_rna.  None pinned (a training step): all four
        # come from ONE launch on the dropout stream, issued on the side stream that consumes them (round 5: as torch draws they were four
        # launches in front of the WSI encoder's first GEMM plus two generator-state fills in front of every graph replay).  Eval mode keeps
        # torch's generator: validate() runs outside the step protocol that advances the dropout stream's device base
        own_draws = _OWN_NOISE and _HEADS_SIDE and self.training and not any(k in noise for k in ("wsi_mask", "rna_mask", "wsi_eps", "rna_eps"))
        if not own_draws:
            if "wsi_mask" not in noise:
                noise["wsi_mask"] = torch.rand(B, wsi_emb.shape[1], device=dev)
            if "rna_mask" not in noise:
                noise["rna_mask"] = torch.rand(B, self.embed_dim, device=dev)
            if "wsi_eps" not in noise:
                noise["wsi_eps"] = torch.randn(B, self.style_latent_dim, device=dev)
            if "rna_eps" not in noise:
                noise["rna_eps"] = torch.randn(B, self.style_latent_dim, device=dev)
        fork = main.record_event()   # the side stream ranks / applies these draws: it has to start behind them

        def run_side():
            side.wait_event(fork)
            with torch.cuda.stream(side):
                if own_draws:
                    noise["wsi_mask"], noise["rna_mask"], noise["wsi_eps"], noise["rna_eps"] = Fn.noise_draws(
                        B, wsi_in.shape[1], self.embed_dim, self.style_latent_dim, dev)
                # the WSI token mask depends on the noise alone: rank it here, long before the retention decoder needs it
                n_tok = wsi_in.shape[1]
                mask_ = Fn.rank_mask(noise["wsi_mask"], int(n_tok * (1 - wsi_mask_ratio)))
                ready_ = side.record_event()
                g = getattr(self, "_rna_graph", None)      # TrainEngine's HIP-graph replay of this branch (graphed.py)
                if (g is not None and g[1] == rna_mask_ratio and self.training and torch.is_grad_enabled()
                        and not torch.cuda.is_current_stream_capturing() and Fn._dropout_state["offset"] == 0   # offsets baked at 0
                        and g[0].matches((rna_in, noise["rna_mask"]))):
                    outs_ = g[0](rna_in, noise["rna_mask"])
                else:
                    outs_ = self.rna_branch(rna_in, noise["rna_mask"], rna_mask_ratio)
            return mask_, ready_, outs_

        # Launch ORDER matters even though the two branches are independent: launches (and the nodes of a captured graph) reach
        # the GPU in program order at a few microseconds each, so ~70 tiny RNA kernels issued first keep the first WSI GEMM
        # waiting for ~0.5 ms, while issued after the WSI encoder's launches they run underneath its long kernels.  Autograd
        # replays later-created nodes first, so the RNA backward is then issued before the WSI backward and overlaps it
        # instead of trailing it.
        # The dropout offsets stay those of the RNA-first order (the RNA branch's HIP-graph replay has them baked in, and the
        # masks do not depend on the launch order): the RNA range is reserved up front once its length is known.
        st = Fn._dropout_state
        key = (tuple(rna_emb.shape), self.training, torch.is_grad_enabled(), rna_mask_ratio)
        n_rna = self._rna_drop_n.get(key) if _RNA_LATE else None
        if n_rna is not None:
            off0 = st["offset"]
            st["offset"] = off0 + n_rna
            Fn.probe("wsi_enc_start")
            wsi_emb = Fn.probe_point(self.wsi_encoder.forward_encoder(wsi_in, wsi_key_padding_mask), "wsi_enc_out")
            off_w, st["offset"] = st["offset"], off0
            wsi_mask, mask_ready, (rna_emb, rna_alignment_emb, rna_retention_emb, rna_mask) = run_side()
            if st["offset"] != off0 + n_rna:
                raise MirrorHipError("the RNA branch consumed a different number of dropout offsets than on its first run")
            st["offset"] = off_w
        else:
            off0 = st["offset"]
            wsi_mask, mask_ready, (rna_emb, rna_alignment_emb, rna_retention_emb, rna_mask) = run_side()
            self._rna_drop_n[key] = st["offset"] - off0
            wsi_emb = self.wsi_encoder.forward_encoder(wsi_in, wsi_key_padding_mask)
        # the encoder output has three consumers (decoder input, retention target, cls row): one node sums their gradients
        wsi_full, wsi_retention_target, wsi_cls = Fn.enc_fanout(wsi_emb)
        # The alignment head of the cls row and the two style / prototype branches are ~25 forward and ~50 backward launches on
        # [B, D] rows (they share their weights, so they stay on ONE stream: their weight gradients accumulate in place): on the
        # RNA branch's stream they run beside the retention decoder (a whole TransLayer) instead of in front of its backward.
        # (A third helper stream that waits for both the main and the RNA stream before its first kernel makes
        # hipStreamEndCapture of the whole-step graph segfault on ROCm 7.2.)
        heads = side if _HEADS_SIDE else main
        heads.wait_event(main.record_event())
        wsi_cls.record_stream(heads)
        with torch.cuda.stream(heads):
            Fn.probe("heads_start")
            wsi_alignment_emb = self.wsi_encoder.forward_alignment_head(wsi_cls)
            wsi_score, wsi_mu, wsi_logstd, rna_score, rna_mu, rna_logstd = self.forward_style_clustering(
                wsi_cls, rna_emb, noise.get("wsi_eps"), noise.get("rna_eps"))
            # `logit_scale.exp()` (models/mirror.py:911) here, beside the retention decoder: at the end of forward() its one-thread
            # launch (and its backward's, which autograd replays on this stream) would stand on the main stream between the last
            # forward GEMM and the loss
            logit_scale = Fn.exp(self.logit_scale)
            Fn.probe("heads_end")
        ag = getattr(self, "_align_gather", None)      # TrainEngine: the loss contrasts against all ranks (gather_distributed)
        if ag is not None:
            # the [B, 2D] all-gather of the global-batch InfoNCE starts now, on a communication stream, under the retention decoder
            from ..losses.mirror_loss import prefetch_alignment_gather
            with torch.cuda.stream(heads):
                prefetch_alignment_gather(wsi_alignment_emb, rna_alignment_emb, ag[0])
        main.wait_event(mask_ready)
        wsi_mask.record_stream(main)
        Fn.probe("decoder_start")
        wsi_retention_emb, wsi_mask = self.wsi_encoder.forward_retention_head(
            wsi_full, mask_ratio=wsi_mask_ratio, mask=wsi_mask, key_padding_mask=wsi_key_padding_mask,
            target=wsi_retention_target if self.training else None)
        Fn.probe("decoder_end")
        main.wait_stream(side)
        Fn.probe("fwd_joined")
        for t in (rna_emb, rna_alignment_emb, rna_retention_emb, rna_mask, wsi_alignment_emb, wsi_score, wsi_mu, wsi_logstd,
                  rna_score, rna_mu, rna_logstd, logit_scale):
            t.record_stream(main)       # allocated in a helper stream's pool, consumed on the main stream
        rna_retention_target = rna_emb
        if own_draws:
            Fn.noise_draws_advance()      # the next step's draws must differ even if this one had no dropout site (Fn.noise_draws)
        return (wsi_alignment_emb, wsi_retention_emb, wsi_retention_target, wsi_mask, wsi_score, wsi_mu, wsi_logstd,
                rna_alignment_emb, rna_retention_emb, rna_retention_target, rna_mask, rna_score, rna_mu, rna_logstd,
                logit_scale)


_ACCEPTED = {
    "wsi_embed_dim", "rna_embed_dim", "embed_dim", "wsi_num_tokens", "wsi_retention_decoder_depth",
    "rna_encoder_depth", "rna_gene_embed", "rna_mlp_ratio", "rna_pos_drop_rate", "rna_proj_drop_rate",
    "rna_attn_drop_rate", "rna_drop_path_rate", "rna_norm_layer", "rna_act_layer", "rna_retention_decoder_depth",
    "init_logit_scale", "style_mlp_hidden_dim", "style_mlp_out_dim", "style_norm_layer", "style_act_layer",
    "style_latent_dim", "num_prototypes",
    "rna_num_heads",  # build-only extension
}


# ===========================================
#  Downstream classifier (inference / fine-tuning encoders)
# ===========================================
class MIRRORClassifier(nn.Module):
    """models/mirror.py:921-1015: FeatureTransMIL (+ TransFormer) -> add / concat fusion -> linear head.  Same kwargs,
    same state-dict keys (`wsi_encoder.*`, `rna_encoder.*`, `head.*`), so the checkpoints that tools/split_weights.py
    cuts out of a pre-trained MIRROR load with strict=False exactly as in train_subtyping.py:740-763."""

    def __init__(self, wsi_embed_dim: int, rna_embed_dim: int, embed_dim: int, num_classes: int, rna_encoder_depth: int = 2,
                 rna_gene_embed: str = "learn", rna_mlp_ratio: float = 2.572, rna_pos_drop_rate: float = 0.0,
                 rna_proj_drop_rate: float = 0.1, rna_attn_drop_rate: float = 0.0, rna_drop_path_rate: float = 0.0,
                 rna_norm_layer=None, rna_act_layer=None, fusion: str = "concat", rna_num_heads: int = 12):
        super().__init__()
        assert fusion in ["add", "concat"], "Fusion must be either add or concat"
        if rna_attn_drop_rate or rna_drop_path_rate:
            raise NotImplementedError("attention dropout / stochastic depth are 0 in every reference config")
        self.embed_dim, self.num_classes, self.fusion = embed_dim, num_classes, fusion
        self.wsi_encoder = FeatureTransMIL(input_dim=wsi_embed_dim, embed_dim=embed_dim)
        self.rna_encoder = TransFormer(input_dim=rna_embed_dim, embed_dim=embed_dim, depth=rna_encoder_depth,
                                       num_heads=rna_num_heads, mlp_ratio=rna_mlp_ratio, gene_embed=rna_gene_embed,
                                       pos_drop_rate=rna_pos_drop_rate, proj_drop_rate=rna_proj_drop_rate)
        self.head = nn.Linear(embed_dim if fusion == "add" else embed_dim * 2, num_classes)
        self.precision: Optional[str] = None

    def forward(self, wsi_emb: torch.Tensor, rna_emb: Optional[torch.Tensor] = None) -> torch.Tensor:
        prec = resolve_precision(self.precision)
        self.wsi_encoder.precision = self.rna_encoder.precision = self.precision
        w = self.wsi_encoder(wsi_emb)                                     # [B, D] f32
        if rna_emb is None:
            fused = w
        else:
            r = self.rna_encoder(rna_emb)
            fused = Fn.add(w, r, f32) if self.fusion == "add" else torch.cat((w, r), dim=1)
        return Fn.linear(fused.contiguous(), self.head.weight, self.head.bias, prec=prec, out_dtype=f32)


def mirror_classifier(**kwargs) -> MIRRORClassifier:
    """models/mirror.py:1056-1085 (kwarg filter with the same warning; `rna_num_heads` is a build-only extra)."""
    accepted = {"wsi_embed_dim", "rna_embed_dim", "embed_dim", "rna_encoder_depth", "rna_gene_embed", "rna_mlp_ratio",
                "rna_pos_drop_rate", "rna_proj_drop_rate", "rna_attn_drop_rate", "rna_drop_path_rate", "rna_norm_layer",
                "rna_act_layer", "num_classes", "fusion", "rna_num_heads"}
    dropped = [k for k in kwargs if k not in accepted]
    if dropped:
        _logger.warning("Filtered model kwargs: %s", ", ".join(dropped))
    return MIRRORClassifier(**{k: v for k, v in kwargs.items() if k in accepted})


def mirror(**kwargs) -> MIRROR:
    """Registry entry point (models/mirror.py:1018-1053): unknown kwargs (e.g. timm's pretrained*) are dropped with a warning."""
    kept = {k: v for k, v in kwargs.items() if k in _ACCEPTED}
    dropped = [k for k in kwargs if k not in _ACCEPTED]
    if dropped:
        _logger.warning("Filtered model kwargs: %s", ", ".join(dropped))
    return MIRROR(**kept)
