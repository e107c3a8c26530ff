"""Deterministic synthetic state-dicts / inputs for parity tests.  TEST INFRASTRUCTURE.

The torch CPU generator is bit-reproducible for a fixed torch version (the build container
and the GPU box run the same image), so large fixtures (BASELINE config 1) are regenerated
from seeds instead of being committed; `tests/golden/*.npz` store a checksum of what the
generator produced when the golden outputs were recorded, and tests verify it.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch

from .mirror_oracle import Cfg


def param_shapes(cfg: Cfg) -> List[Tuple[str, Tuple[int, ...]]]:
    """State-dict keys and shapes of `models.mirror.MIRROR` (SURVEY.md §8b key contract)."""
    d, f, g, n = cfg.embed_dim, cfg.wsi_embed_dim, cfg.rna_embed_dim, cfg.wsi_num_tokens
    hh = int(d * cfg.rna_mlp_ratio)
    out: List[Tuple[str, Tuple[int, ...]]] = [("logit_scale", ())]

    def lin(p, o, i, bias=True):
        out.append((p + ".weight", (o, i)))
        if bias:
            out.append((p + ".bias", (o,)))

    def ln(p, c):
        out.extend([(p + ".weight", (c,)), (p + ".bias", (c,))])

    def trans(p):
        ln(p + ".norm", d)
        lin(p + ".attn.to_qkv", 3 * d, d, bias=False)
        lin(p + ".attn.to_out.0", d, d)
        out.append((p + ".attn.res_conv.weight", (cfg.wsi_heads, 1, cfg.res_conv_kernel, 1)))

    def block(p):
        ln(p + ".norm1", d)
        lin(p + ".attn.qkv", 3 * d, d)
        lin(p + ".attn.proj", d, d)
        ln(p + ".norm2", d)
        lin(p + ".mlp.fc1", hh, d)
        lin(p + ".mlp.fc2", d, hh)

    w = "wsi_encoder"
    out.append((w + ".cls_token", (1, 1, d)))
    for nm, k in (("proj", 7), ("proj1", 5), ("proj2", 3)):
        out.extend([(f"{w}.pos_layer.{nm}.weight", (d, 1, k, k)), (f"{w}.pos_layer.{nm}.bias", (d,))])
    lin(w + "._fc1.0", d, f)
    trans(w + ".layer1")
    trans(w + ".layer2")
    ln(w + ".norm", d)
    lin(w + ".alignment_head", d, d)
    lin(w + ".retention_embed", d, d)
    out.append((w + ".mask_token", (1, 1, d)))
    out.append((w + ".retention_gene_embed", (1, n + 1, d)))
    for i in range(cfg.wsi_retention_decoder_depth):
        trans(f"{w}.retention_blocks.{i}")
    ln(w + ".retention_norm", d)
    lin(w + ".retention_head", d, d)

    r = "rna_encoder"
    lin(r + ".embedding.fc1", 2 * d, g)
    ln(r + ".embedding.norm", 2 * d)
    lin(r + ".embedding.fc2", d, 2 * d)
    out.append((r + ".gene_embed", (1, d)))
    for i in range(cfg.rna_encoder_depth):
        block(f"{r}.blocks.{i}")
    ln(r + ".norm", d)
    lin(r + ".alignment_head", d, d)
    lin(r + ".retention_embed", d, d)
    out.append((r + ".mask_token", (1, 1)))
    out.append((r + ".retention_gene_embed", (1, d)))
    for i in range(cfg.rna_retention_decoder_depth):
        block(f"{r}.retention_blocks.{i}")
    ln(r + ".retention_norm", d)
    lin(r + ".retention_head", d, d)

    lin("style_encoder_mlp.fc1", cfg.style_mlp_hidden_dim, d)
    lin("style_encoder_mlp.fc2", cfg.style_mlp_out_dim, cfg.style_mlp_hidden_dim)
    lin("style_mu", cfg.style_latent_dim, cfg.style_mlp_out_dim)
    lin("style_logstd", cfg.style_latent_dim, cfg.style_mlp_out_dim)
    lin("style_decoder", d, cfg.style_latent_dim)
    out.append(("prototypes.weight", (cfg.num_prototypes, d)))
    return out


def classifier_param_shapes(cfg: Cfg, num_classes: int, fusion: str = "concat") -> List[Tuple[str, Tuple[int, ...]]]:
    """State-dict keys and shapes of `models.mirror.MIRRORClassifier` (models/mirror.py:921-996): the encoder keys of
    MIRROR without the pre-training heads, plus `head`."""
    drop = ("alignment_head", "retention_", "mask_token", "style_", "prototypes", "logit_scale")
    out = [(k, sh) for k, sh in param_shapes(cfg) if not any(d in k for d in drop)]
    d = cfg.embed_dim
    out.append(("head.weight", (num_classes, 2 * d if fusion == "concat" else d)))
    out.append(("head.bias", (num_classes,)))
    return out


def synth_state_dict(shapes: Sequence[Tuple[str, Tuple[int, ...]]], seed: int) -> Dict[str, torch.Tensor]:
    """Trained-like magnitudes: fan-in scaled matrices, LN gains near 1, small biases/tokens,
    unit-norm prototype rows (train_mirror.py:1133-1136), logit_scale = ln(1/0.07)."""
    sd: Dict[str, torch.Tensor] = {}
    for idx, (key, shape) in enumerate(shapes):
        g = torch.Generator().manual_seed(seed * 100003 + idx)
        shape = tuple(int(s) for s in shape)
        if key == "logit_scale":
            t = torch.tensor(math.log(1 / 0.07), dtype=torch.float32)
        elif key == "prototypes.weight":
            t = torch.nn.functional.normalize(torch.randn(shape, generator=g), dim=1)
        elif key.endswith("res_conv.weight"):
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(shape[2])
        elif "pos_layer" in key and key.endswith(".weight"):
            t = (torch.rand(shape, generator=g) * 2 - 1) / shape[-1]
        elif key.endswith(".weight") and len(shape) == 2:
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        elif key.endswith(".weight") and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif key.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:  # tokens / embeddings
            t = 0.02 * torch.randn(shape, generator=g)
        sd[key] = t.to(torch.float32)
    return sd


def synth_batch(cfg: Cfg, batch: int, seed: int):
    """Inputs with the dataset's contract (datasets/dataset_pretrain.py:150-167): wsi [B,N,F] f32,
    rna [B,G] f32; plus the four noise draws of the forward in the reference's order."""
    g = torch.Generator().manual_seed(seed)
    wsi = torch.randn(batch, cfg.wsi_num_tokens, cfg.wsi_embed_dim, generator=g)
    rna = torch.randn(batch, cfg.rna_embed_dim, generator=g)
    noise = {
        "wsi_mask": torch.rand(batch, cfg.wsi_num_tokens, generator=g),
        "rna_mask": torch.rand(batch, cfg.embed_dim, generator=g),
        "wsi_eps": torch.randn(batch, cfg.style_latent_dim, generator=g),
        "rna_eps": torch.randn(batch, cfg.style_latent_dim, generator=g),
    }
    return wsi, rna, noise


def checksum(tensors) -> float:
    """Order-dependent scalar fingerprint (float64) of a list of tensors."""
    acc = 0.0
    for i, t in enumerate(tensors):
        t64 = t.detach().double().flatten()
        w = torch.cos(torch.arange(t64.numel(), dtype=torch.float64) * 0.37 + i)
        acc += float((t64 * w).sum())
    return acc
