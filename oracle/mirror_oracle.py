"""CPU oracle for the MIRROR pre-training hot path.  TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product path (`mirror_amd`) never does: it runs hand-written HIP kernels and
fails loudly when the HIP library is missing.

This is a *functional* fp32 restatement (plain torch CPU ops on a flat state-dict) of what
the reference computes on the path `train_mirror.py` -> `models.mirror.MIRROR.forward`
-> `losses.MIRRORLoss.forward` (+ `losses.InfoNCE`).  Every function cites the reference
file:line (relative to /root/reference) it follows.  The four random draws of the forward
(`models/mirror.py:630`, `:516`, `:832-833` twice) are explicit `noise` inputs so that results
are reproducible.  Dropout is eval-mode (identity) by default; train-mode numerics are
modelled by handing the keep-multipliers in: `noise["dropout"] = {"wsi": [...], "rna": [...]}`,
one f32 tensor (0 or 1/(1-p)) per nn.Dropout site in program order (see `_drop`).

PINNING: checked against golden vectors produced by importing the reference itself in the
build container (`tools/make_golden.py` -> `tests/golden/*.npz`; `tests/test_oracle.py`).
`losses/*` is exercised by the real reference code.  `models/mirror.py` is exercised by
the real reference code with stand-ins for two pip packages that are absent from this
image (`timm~=1.0.15`, `nystrom_attention~=0.0.14`; `tools/oracle_shims.py`).  The
arithmetic inside those two packages (Nystrom attention, timm `Mlp`/LayerNorm-eps) is
therefore **parity unpinned**: restated from the published algorithm, not verified against
the pinned wheels (the reference has no tests or golden vectors of its own).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def exact_cpu_convs():
    """Context manager for parity runs: torch 2.10's oneDNN CPU conv2d weight-gradient is wrong for some
    depthwise shapes (33-tap res_conv at n_p=256: error O(10) against an explicit sum; found while pinning the
    HIP path).  With oneDNN off ATen's native kernels are used, which agree with the explicit sum."""
    return torch.backends.mkldnn.flags(enabled=False)


@dataclass
class Cfg:
    """Model hyper-parameters (names follow `models/mirror.py:721-745`)."""
    wsi_embed_dim: int          # F
    rna_embed_dim: int          # G
    embed_dim: int              # D
    wsi_num_tokens: int = 2048  # N
    wsi_retention_decoder_depth: int = 1
    rna_encoder_depth: int = 2
    rna_mlp_ratio: float = 2.572
    rna_retention_decoder_depth: int = 1
    rna_num_heads: int = 12     # fixed at 12 in the reference (models/mirror.py:392, :798-811)
    style_mlp_hidden_dim: int = 512
    style_mlp_out_dim: int = 256
    style_latent_dim: int = 128
    num_prototypes: int = 3000
    rna_norm_eps: float = 1e-6  # timm LayerNorm / partial(nn.LayerNorm, eps=1e-6), models/mirror.py:210
    wsi_heads: int = 8          # models/mirror.py:302
    pinv_iterations: int = 6    # models/mirror.py:304
    res_conv_kernel: int = 33   # nystrom_attention default


# ----------------------------------------------------------------------------- helpers
def _ln(x: Tensor, sd: SD, p: str, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _linear(x: Tensor, sd: SD, p: str) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _drop(x: Tensor, drop) -> Tensor:
    """nn.Dropout in train mode with the mask handed in: `drop` is an iterator over keep-multipliers (0 or 1/(1-p), shaped
    like x) in program order, or None (eval mode: identity)."""
    if drop is None:
        return x
    m = next(drop)
    assert m.shape == x.shape, (tuple(m.shape), tuple(x.shape))
    return x * m


# ------------------------------------------------------------------- Nystrom attention
def pinv_iter(x: Tensor, iters: int) -> Tensor:
    """[3P nystrom_attention] moore_penrose_iter_pinv: tensor-wide max in the initial scale."""
    ax = x.abs()
    z = x.transpose(-1, -2) / (ax.sum(-1).max() * ax.sum(-2).max())
    eye = torch.eye(x.shape[-1], dtype=x.dtype).expand_as(x)
    for _ in range(iters):
        xz = x @ z
        z = 0.25 * z @ (13 * eye - xz @ (15 * eye - xz @ (7 * eye - xz)))
    return z


def nystrom_attention(x: Tensor, sd: SD, p: str, cfg: Cfg, mask: Optional[Tensor] = None, drop=None) -> Tensor:
    """[3P] NystromAttention.forward as configured at models/mirror.py:299-309.

    dim_head = D//8, heads = 8, landmarks m = D//2, 6 pinv iterations, residual 33-tap
    depthwise conv on v, to_out = Linear (+Dropout, off here).  `mask` ([B, n] bool) is the
    package's optional key-padding mask (never passed by the reference, models/mirror.py:312).
    """
    b, n, d = x.shape
    h, m = cfg.wsi_heads, d // 2
    dh = d // h
    pad = (m - n % m) % m
    if pad:
        x = F.pad(x, (0, 0, pad, 0))
        if mask is not None:
            mask = F.pad(mask, (pad, 0), value=False)
    n_p = n + pad
    qkv = F.linear(x, sd[p + ".to_qkv.weight"])
    q, k, v = (t.reshape(b, n_p, h, dh).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
    if mask is not None:
        mk = mask[:, None, :, None].to(q.dtype)
        q, k, v = q * mk, k * mk, v * mk
    q = q * dh ** -0.5
    l = math.ceil(n / m)  # noqa: E741
    q_l = q.reshape(b, h, n_p // l, l, dh).sum(3)
    k_l = k.reshape(b, h, n_p // l, l, dh).sum(3)
    if mask is None:
        q_l, k_l = q_l / l, k_l / l
    else:
        cnt = mask.reshape(b, 1, n_p // l, l).sum(-1).to(q.dtype)
        q_l, k_l = q_l / (cnt[..., None] + 1e-8), k_l / (cnt[..., None] + 1e-8)
        ml = cnt > 0
    s1 = q @ k_l.transpose(-1, -2)
    s2 = q_l @ k_l.transpose(-1, -2)
    s3 = q_l @ k.transpose(-1, -2)
    if mask is not None:
        neg = -torch.finfo(q.dtype).max
        mb = mask[:, None, :]
        s1 = s1.masked_fill(~(mb[..., None] & ml[..., None, :]), neg)
        s2 = s2.masked_fill(~(ml[..., None] & ml[..., None, :]), neg)
        s3 = s3.masked_fill(~(ml[..., None] & mb[..., None, :]), neg)
    a1, a2, a3 = s1.softmax(-1), s2.softmax(-1), s3.softmax(-1)
    out = (a1 @ pinv_iter(a2, cfg.pinv_iterations)) @ (a3 @ v)
    ks = cfg.res_conv_kernel
    out = out + F.conv2d(v, sd[p + ".res_conv.weight"], padding=(ks // 2, 0), groups=h)
    out = out.transpose(1, 2).reshape(b, n_p, h * dh)
    out = _linear(out, sd, p + ".to_out.0")
    # to_out = Sequential(Linear, Dropout(0.1)) (models/mirror.py:308): the package drops on the padded sequence and slices
    # afterwards; the masks are i.i.d. per element, so a mask drawn for the surviving rows only is the same experiment
    return _drop(out[:, -n:], drop)


def trans_layer(x: Tensor, sd: SD, p: str, cfg: Cfg, mask: Optional[Tensor] = None, drop=None) -> Tensor:
    """TransLayer.forward, models/mirror.py:311-314 (nn.LayerNorm eps 1e-5, :296-298)."""
    return x + nystrom_attention(_ln(x, sd, p + ".norm", 1e-5), sd, p + ".attn", cfg, mask, drop)


def ppeg(x: Tensor, sd: SD, p: str, hh: int, ww: int) -> Tensor:
    """PPEG.forward, models/mirror.py:324-331."""
    b, _, c = x.shape
    cls, feat = x[:, :1], x[:, 1:]
    g = feat.transpose(1, 2).reshape(b, c, hh, ww)
    y = (F.conv2d(g, sd[p + ".proj.weight"], sd[p + ".proj.bias"], padding=3, groups=c) + g
         + F.conv2d(g, sd[p + ".proj1.weight"], sd[p + ".proj1.bias"], padding=2, groups=c)
         + F.conv2d(g, sd[p + ".proj2.weight"], sd[p + ".proj2.bias"], padding=1, groups=c))
    return torch.cat([cls, y.flatten(2).transpose(1, 2)], dim=1)


# ---------------------------------------------------------------------------- WSI side
def wsi_forward_encoder(wsi: Tensor, sd: SD, cfg: Cfg, p: str = "wsi_encoder", mask: Optional[Tensor] = None, drop=None) -> Tensor:
    """FeatureTransMILHybrid.forward_encoder, models/mirror.py:651-679.  `mask` ([B, N] bool, True = real patch) is the
    BASELINE config-4 extension: the reference has no such argument; the sequence [cls, x, x[:add]] carries
    [True, mask, mask[:, :add]] into the package's key-padding `mask` of both Nystrom layers."""
    h = F.relu(_linear(wsi.float(), sd, p + "._fc1.0"))
    n = h.shape[1]
    side = int(math.ceil(math.sqrt(n)))
    add = side * side - n
    h = torch.cat([h, h[:, :add]], dim=1)
    h = torch.cat([sd[p + ".cls_token"].expand(h.shape[0], -1, -1), h], dim=1)
    smask = None if mask is None else torch.cat([torch.ones_like(mask[:, :1]), mask, mask[:, :add]], dim=1)
    h = trans_layer(h, sd, p + ".layer1", cfg, smask, drop)
    h = ppeg(h, sd, p + ".pos_layer", side, side)
    h = trans_layer(h, sd, p + ".layer2", cfg, smask, drop)
    h = _ln(h, sd, p + ".norm", 1e-5)
    return h[:, : h.shape[1] - add]


def rank_mask(noise: Tensor, len_keep: int) -> Tensor:
    """mask = 1 where argsort-rank(noise) >= len_keep (models/mirror.py:632-647, :518-531)."""
    rank = torch.argsort(torch.argsort(noise, dim=1), dim=1)
    return (rank >= len_keep).to(torch.float32)


def wsi_forward_decoders(h: Tensor, sd: SD, cfg: Cfg, ratio: float, noise: Tensor,
                         p: str = "wsi_encoder", kp_mask: Optional[Tensor] = None, drop=None) -> Tuple[Tensor, Tensor, Tensor]:
    """forward_decoders, models/mirror.py:701-706 (+ :681-699, :624-649)."""
    align = _linear(F.normalize(h, dim=-1, p=2, eps=1e-12)[:, 0], sd, p + ".alignment_head")
    r = _linear(h, sd, p + ".retention_embed")
    n = r.shape[1] - 1
    mask = rank_mask(noise, int(n * (1 - ratio)))
    tok = torch.where(mask[..., None] > 0, sd[p + ".mask_token"].expand(r.shape[0], n, -1), r[:, 1:])
    r = torch.cat([r[:, :1], tok], dim=1) + sd[p + ".retention_gene_embed"]
    kp = None if kp_mask is None else torch.cat([torch.ones_like(kp_mask[:, :1]), kp_mask], dim=1)
    for i in range(cfg.wsi_retention_decoder_depth):
        r = trans_layer(r, sd, f"{p}.retention_blocks.{i}", cfg, kp, drop)
    r = _linear(_ln(r, sd, p + ".retention_norm", 1e-5), sd, p + ".retention_head")
    return align, r[:, 1:], mask


# ---------------------------------------------------------------------------- RNA side
def rna_attention(x: Tensor, sd: SD, p: str, heads: int) -> Tensor:
    """Attention.forward on a 2-D [B, D] input, models/mirror.py:77-102.

    q,k,v are [B, heads, head_dim]; SDPA therefore attends over the *heads* axis and the
    result is permuted by transpose(1,2).reshape(B, D) before `proj`.
    """
    b, d = x.shape
    hd = d // heads
    q, k, v = _linear(x, sd, p + ".qkv").reshape(b, 3, heads, hd).unbind(1)
    a = ((q * hd ** -0.5) @ k.transpose(-1, -2)).softmax(-1)
    o = (a @ v).transpose(1, 2).reshape(b, d)
    return _linear(o, sd, p + ".proj")


def rna_block(x: Tensor, sd: SD, p: str, cfg: Cfg, drop=None) -> Tensor:
    """Block.forward, models/mirror.py:149-152 (LayerScale/DropPath are identities).  Train mode: proj_drop behind the
    attention's `proj` (:101) and timm Mlp's drop1 (behind the activation) / drop2 (behind fc2), all at rna_proj_drop_rate
    (:128-129, :142); attn_drop is 0 (:733)."""
    e = cfg.rna_norm_eps
    x = x + _drop(rna_attention(_ln(x, sd, p + ".norm1", e), sd, p + ".attn", cfg.rna_num_heads), drop)
    y = _drop(F.gelu(_linear(_ln(x, sd, p + ".norm2", e), sd, p + ".mlp.fc1")), drop)
    return x + _drop(_linear(y, sd, p + ".mlp.fc2"), drop)


def rna_forward_encoder(rna: Tensor, sd: SD, cfg: Cfg, p: str = "rna_encoder", drop=None) -> Tensor:
    """TransFormer.forward, models/mirror.py:283-289; embedding = timm Mlp(G->2D->D, norm=LN(2D))."""
    x = F.gelu(_linear(rna, sd, p + ".embedding.fc1"))
    x = _linear(_ln(x, sd, p + ".embedding.norm", cfg.rna_norm_eps), sd, p + ".embedding.fc2")
    x = x + sd[p + ".gene_embed"]
    for i in range(cfg.rna_encoder_depth):
        x = rna_block(x, sd, f"{p}.blocks.{i}", cfg, drop)
    return _ln(x, sd, p + ".norm", cfg.rna_norm_eps)


def rna_forward_decoders(x: Tensor, sd: SD, cfg: Cfg, ratio: float, noise: Tensor,
                         p: str = "rna_encoder", drop=None) -> Tuple[Tensor, Tensor, Tensor]:
    """forward_decoders, models/mirror.py:556-561 (+ :538-554, :510-533): channel masking."""
    align = _linear(F.normalize(x, dim=-1, p=2, eps=1e-12), sd, p + ".alignment_head")
    r = _linear(x, sd, p + ".retention_embed")
    mask = rank_mask(noise, int(r.shape[1] * (1 - ratio)))
    r = torch.where(mask > 0, sd[p + ".mask_token"].expand_as(r), r) + sd[p + ".retention_gene_embed"]
    for i in range(cfg.rna_retention_decoder_depth):
        r = rna_block(r, sd, f"{p}.retention_blocks.{i}", cfg, drop)
    r = _linear(_ln(r, sd, p + ".retention_norm", cfg.rna_norm_eps), sd, p + ".retention_head")
    return align, r, mask


# ------------------------------------------------------------------------- style / heads
def style_branch(x: Tensor, sd: SD, eps: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """One modality of forward_style_clustering, models/mirror.py:845-850 (+ :830-833)."""
    hdn = _linear(F.gelu(_linear(x, sd, "style_encoder_mlp.fc1")), sd, "style_encoder_mlp.fc2")
    mu, logstd = _linear(hdn, sd, "style_mu"), _linear(hdn, sd, "style_logstd")
    z = mu + eps * torch.exp(0.5 * logstd)
    score = F.linear(_linear(z, sd, "style_decoder"), sd["prototypes.weight"])
    return score, mu, logstd


def mirror_forward(sd: SD, cfg: Cfg, wsi: Tensor, rna: Tensor, noise: Dict[str, Tensor],
                   wsi_mask_ratio: float = 0.75, rna_mask_ratio: float = 0.75, wsi_key_padding_mask: Optional[Tensor] = None):
    """MIRROR.forward, models/mirror.py:860-915.  noise keys: wsi_mask [B,N], rna_mask [B,D],
    wsi_eps [B,latent], rna_eps [B,latent] (the draw order of the reference).  `wsi_key_padding_mask` ([B, N] bool): the
    BASELINE config-4 extension (variable-length slides), see wsi_forward_encoder."""
    dm = noise.get("dropout")          # train mode: {"wsi": [2 encoder layers, decoder blocks], "rna": [3 per Block, encoder then decoder]}
    dw = None if dm is None else iter(dm["wsi"])
    dr = None if dm is None else iter(dm["rna"])
    w = wsi_forward_encoder(wsi, sd, cfg, mask=wsi_key_padding_mask, drop=dw)
    w_align, w_ret, w_mask = wsi_forward_decoders(w, sd, cfg, wsi_mask_ratio, noise["wsi_mask"], kp_mask=wsi_key_padding_mask, drop=dw)
    r = rna_forward_encoder(rna, sd, cfg, drop=dr)
    r_align, r_ret, r_mask = rna_forward_decoders(r, sd, cfg, rna_mask_ratio, noise["rna_mask"], drop=dr)
    if dm is not None:
        assert next(dw, None) is None and next(dr, None) is None, "unused dropout masks"
    w_score, w_mu, w_ls = style_branch(w[:, 0], sd, noise["wsi_eps"])
    r_score, r_mu, r_ls = style_branch(r, sd, noise["rna_eps"])
    return (w_align, w_ret, w[:, 1:], w_mask, w_score, w_mu, w_ls,
            r_align, r_ret, r, r_mask, r_score, r_mu, r_ls, sd["logit_scale"].exp())


OUTPUT_NAMES = ("wsi_alignment_emb", "wsi_retention_emb", "wsi_retention_target", "wsi_mask",
                "wsi_score", "wsi_mu", "wsi_logstd", "rna_alignment_emb", "rna_retention_emb",
                "rna_retention_target", "rna_mask", "rna_score", "rna_mu", "rna_logstd",
                "logit_scale")
LOSS_NAMES = ("total", "alignment", "wsi_retention", "rna_retention", "style", "cluster")


# -------------------------------------------------------------------------------- losses
def classifier_forward(sd: SD, cfg: Cfg, wsi: Tensor, rna: Optional[Tensor], fusion: str = "concat") -> Tensor:
    """MIRRORClassifier.forward, models/mirror.py:998-1015: FeatureTransMIL (cls token of the normalised sequence,
    :345-380) [+ TransFormer, :283-289] -> add / concat -> head."""
    w = wsi_forward_encoder(wsi, sd, cfg)[:, 0]
    if rna is None:
        return _linear(w, sd, "head")
    r = rna_forward_encoder(rna, sd, cfg)
    fused = w + r if fusion == "add" else torch.cat([w, r], dim=1)
    return _linear(fused, sd, "head")


def dataset_getitem(wsi_feature: Tensor, rna_row, num_tokens: int):
    """TCGAWSIRNAPretrainDataset.__getitem__, datasets/dataset_pretrain.py:150-167: sample `num_tokens` patch rows with numpy's
    GLOBAL RNG (`np.random.choice`, with replacement only when the slide is shorter than num_tokens), gather them, and
    return the slide's RNA row as float32.  Also returns the drawn indices."""
    import numpy as np
    n = wsi_feature.shape[0]
    is_replace = not n >= num_tokens
    idx = np.random.choice(n, num_tokens, replace=is_replace)
    return wsi_feature[idx], torch.tensor(np.asarray(rna_row), dtype=torch.float32), idx


def clip_loss(w: Tensor, r: Tensor, scale: Tensor) -> Tensor:
    """ClipLoss.forward, losses/mirror_loss.py:37-52."""
    lab = torch.arange(w.shape[0])
    return 0.5 * (F.cross_entropy(scale * w @ r.T, lab) + F.cross_entropy(scale * r @ w.T, lab))


def mirror_loss(outs: Sequence[Tensor], weights: Sequence[float] = (0.5, 0.1, 0.1, 0.1, 0.2)):
    """MIRRORLoss.forward, losses/mirror_loss.py:74-135; default weights :59-63."""
    (w_al, w_ret, w_tgt, w_mask, w_sc, w_mu, w_ls,
     r_al, r_ret, r_tgt, r_mask, r_sc, r_mu, r_ls, scale) = outs
    align = clip_loss(w_al, r_al, scale)
    wret = (((w_ret - w_tgt) ** 2).mean(-1) * w_mask).sum() / w_mask.sum()
    rret = (((r_ret - r_tgt) ** 2) * r_mask).sum() / r_mask.sum()
    style = 0.5 * ((w_ls.exp() + w_mu ** 2 - 1 - w_ls).sum(1).mean()
                   + (r_ls.exp() + r_mu ** 2 - 1 - r_ls).sum(1).mean())
    lw, lr = F.log_softmax(w_sc, -1), F.log_softmax(r_sc, -1)
    bsz = w_sc.shape[0]
    cluster = 0.5 * ((lr.exp() * (lr - lw)).sum() / bsz + (lw.exp() * (lw - lr)).sum() / bsz)
    total = (weights[0] * align + weights[1] * wret + weights[2] * rret
             + weights[3] * style + weights[4] * cluster)
    return total, align, wret, rret, style, cluster


def info_nce(query: Tensor, positive_key: Tensor, temperature: float = 0.1,
             reduction: str = "mean", symmetric: bool = False) -> Tensor:
    """InfoNCE.info_nce with negative_keys=None, losses/info_nce.py:122-124, :144-164."""
    q, k = F.normalize(query, dim=-1), F.normalize(positive_key, dim=-1)
    lab = torch.arange(q.shape[0])
    loss = F.cross_entropy(q @ k.T / temperature, lab, reduction=reduction)
    if symmetric:
        loss = 0.5 * loss + 0.5 * F.cross_entropy(k @ q.T / temperature, lab, reduction=reduction)
    return loss
