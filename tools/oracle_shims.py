"""Stand-ins for the two pip dependencies of the reference that are NOT installed here.

TEST INFRASTRUCTURE ONLY (used by tools/make_golden.py in the build container).

The reference's `models/mirror.py:28-39` imports `nystrom_attention.NystromAttention`
(pinned `nystrom_attention~=0.0.14`, requirements.txt:3) and a handful of symbols from
`timm` (`timm~=1.0.15`, requirements.txt:2).  Neither package is vendored under
/root/reference nor importable in this image, so the arithmetic that lives in them is
restated here from the published algorithms:

* Nystromformer attention (Xiong et al., AAAI 2021; lucidrains/nystrom-attention 0.0.14):
  front zero-padding to a multiple of the landmark count, segment-mean landmarks,
  three softmax kernels, the iterative Moore-Penrose pseudo-inverse with a tensor-wide
  max in its initial scaling, `(a1 @ a2inv) @ (a3 @ v)`, a 33-tap depthwise residual
  convolution of the values, `to_out = Linear + Dropout`, last-n slice.
* timm.layers.Mlp op order fc1 -> act -> drop1 -> norm -> fc2 -> drop2, timm LayerNorm
  eps 1e-6, exact-erf GELU, DropPath / LayerScale identities at the reference's settings.

PARITY STATUS: "parity unpinned" at this third-party boundary — the reference ships no
tests or golden vectors for it.  `tools/make_golden.py --use-installed`
prefers the real packages when importable so the fixtures can be regenerated and diffed
on a machine that has them.
"""
from __future__ import annotations

import math
import sys
import types
from typing import Callable, Optional, Type, Union

import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------- timm
class DropPath(nn.Module):
    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        r = x.new_empty(shape).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            r.div_(keep)
        return x * r


LayerType = Union[str, Callable, Type[nn.Module]]


class LayerNorm(nn.LayerNorm):
    """timm.layers.LayerNorm: nn.LayerNorm with eps defaulting to 1e-6."""

    def __init__(self, num_channels, eps=1e-6, affine=True):
        super().__init__(num_channels, eps=eps, elementwise_affine=affine)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None,
                 act_layer=nn.GELU, norm_layer=None, bias=True, drop=0.0, use_conv=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        x = self.fc1(x)
        x = self.act(x)
        x = self.drop1(x)
        x = self.norm(x)
        x = self.fc2(x)
        x = self.drop2(x)
        return x


def get_act_layer(name=None):
    if name is None:
        return None
    if not isinstance(name, str):
        return name
    if not name:
        return None
    table = {"gelu": nn.GELU, "relu": nn.ReLU, "silu": nn.SiLU, "tanh": nn.Tanh}
    return table[name.lower()]


def get_norm_layer(name=None):
    if name is None:
        return None
    if not isinstance(name, str):
        return name
    if not name:
        return None
    table = {"layernorm": LayerNorm, "ln": LayerNorm}
    return table[name.replace("_", "").lower()]


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def use_fused_attn(experimental: bool = False) -> bool:
    return True


def register_model(fn):
    return fn


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5, inplace=False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x.mul_(self.gamma) if self.inplace else x * self.gamma


# ------------------------------------------------------------- nystrom_attention
def moore_penrose_iter_pinv(x, iters=6):
    abs_x = torch.abs(x)
    col = abs_x.sum(dim=-1)
    row = abs_x.sum(dim=-2)
    z = x.transpose(-1, -2) / (torch.max(col) * torch.max(row))
    eye = torch.eye(x.shape[-1], device=x.device, dtype=x.dtype).unsqueeze(0)
    for _ in range(iters):
        xz = x @ z
        z = 0.25 * z @ (13 * eye - (xz @ (15 * eye - (xz @ (7 * eye - xz)))))
    return z


class NystromAttention(nn.Module):
    def __init__(self, dim, dim_head=64, heads=8, num_landmarks=256, pinv_iterations=6,
                 residual=True, residual_conv_kernel=33, eps=1e-8, dropout=0.0):
        super().__init__()
        self.eps = eps
        inner_dim = heads * dim_head
        self.num_landmarks = num_landmarks
        self.pinv_iterations = pinv_iterations
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))
        self.residual = residual
        if residual:
            ks = residual_conv_kernel
            self.res_conv = nn.Conv2d(heads, heads, (ks, 1), padding=(ks // 2, 0),
                                      groups=heads, bias=False)

    def forward(self, x, mask=None, return_attn=False):
        b, n, _ = x.shape
        h, m, iters, eps = self.heads, self.num_landmarks, self.pinv_iterations, self.eps
        remainder = n % m
        if remainder > 0:
            padding = m - remainder
            x = F.pad(x, (0, 0, padding, 0), value=0)
            if mask is not None:
                mask = F.pad(mask, (padding, 0), value=False)
        q, k, v = self.to_qkv(x).chunk(3, dim=-1)

        def split(t):
            return t.reshape(b, t.shape[1], h, -1).permute(0, 2, 1, 3)

        q, k, v = split(q), split(k), split(v)
        if mask is not None:
            mask = mask[:, None, :]
            q, k, v = (t * mask[..., None] for t in (q, k, v))
        q = q * self.scale
        l = math.ceil(n / m)  # noqa: E741
        n_p = q.shape[2]
        q_l = q.reshape(b, h, n_p // l, l, -1).sum(dim=3)
        k_l = k.reshape(b, h, n_p // l, l, -1).sum(dim=3)
        divisor = l
        if mask is not None:
            mls = mask.reshape(b, 1, n_p // l, l).sum(dim=-1)
            divisor = mls[..., None] + eps
            mask_l = mls > 0
        q_l = q_l / divisor
        k_l = k_l / divisor
        sim1 = q @ k_l.transpose(-1, -2)
        sim2 = q_l @ k_l.transpose(-1, -2)
        sim3 = q_l @ k.transpose(-1, -2)
        if mask is not None:
            mv = -torch.finfo(q.dtype).max
            sim1.masked_fill_(~(mask[..., None] * mask_l[..., None, :]), mv)
            sim2.masked_fill_(~(mask_l[..., None] * mask_l[..., None, :]), mv)
            sim3.masked_fill_(~(mask_l[..., None] * mask[..., None, :]), mv)
        a1, a2, a3 = (t.softmax(dim=-1) for t in (sim1, sim2, sim3))
        a2_inv = moore_penrose_iter_pinv(a2, iters)
        out = (a1 @ a2_inv) @ (a3 @ v)
        if self.residual:
            out = out + self.res_conv(v)
        out = out.permute(0, 2, 1, 3).reshape(b, n_p, -1)
        out = self.to_out(out)
        out = out[:, -n:]
        if return_attn:
            return out, a1 @ a2_inv @ a3
        return out


def install(use_installed: bool = False) -> dict:
    """Register the stand-ins in sys.modules (unless the real packages import)."""
    used = {}
    have_timm = have_nys = False
    if use_installed:
        try:
            import timm  # noqa: F401
            have_timm = True
        except Exception:
            pass
        try:
            import nystrom_attention  # noqa: F401
            have_nys = True
        except Exception:
            pass
    if not have_timm:
        timm = types.ModuleType("timm")
        layers = types.ModuleType("timm.layers")
        for name in ("DropPath", "LayerType", "Mlp", "get_act_layer", "get_norm_layer",
                     "trunc_normal_", "use_fused_attn", "LayerNorm"):
            setattr(layers, name, globals()[name])
        models = types.ModuleType("timm.models")
        models.register_model = register_model
        vit = types.ModuleType("timm.models.vision_transformer")
        vit.LayerScale = LayerScale
        models.vision_transformer = vit
        timm.layers, timm.models = layers, models
        sys.modules.update({"timm": timm, "timm.layers": layers, "timm.models": models,
                            "timm.models.vision_transformer": vit})
    if not have_nys:
        nys = types.ModuleType("nystrom_attention")
        nys.NystromAttention = NystromAttention
        sys.modules["nystrom_attention"] = nys
    used["timm"] = "installed" if have_timm else "stand-in"
    used["nystrom_attention"] = "installed" if have_nys else "stand-in"
    return used
