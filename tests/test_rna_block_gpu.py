"""GPU: the fused RNA Block (mh_rna_block_fwd / mh_rna_block_bwd, csrc/rna_block.hip) against the oracle's `rna_block`
(models/mirror.py:149-152, :77-102) and against the composed HIP ops it replaces."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

from mirror_amd import functional as Fn  # noqa: E402
from oracle import mirror_oracle as O  # noqa: E402

import importlib  # noqa: E402
mm = importlib.import_module("mirror_amd.models.mirror")


def _block(D, H, ratio, drop, seed):
    torch.manual_seed(seed)
    blk = mm.Block(D, H, ratio, True, drop, 1e-6)
    with torch.no_grad():                                   # non-trivial LayerNorm parameters and biases
        for p in blk.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    return blk


def _sd(blk, prefix="b"):
    m = {"norm1": "norm1", "norm2": "norm2", "attn.qkv": "attn.qkv", "attn.proj": "attn.proj", "mlp.fc1": "mlp.fc1", "mlp.fc2": "mlp.fc2"}
    sd = {}
    for k, v in blk.state_dict().items():
        sd[f"{prefix}.{k}"] = v.detach().clone().float()
    return sd


@pytest.mark.parametrize("B,D,H,ratio", [(16, 512, 8, 4.0), (8, 256, 8, 4.0), (3, 96, 12, 2.0), (32, 128, 8, 1.0)])
def test_fused_block_matches_oracle_forward_and_backward(B, D, H, ratio):
    """Eval mode (no dropout), bf16 policy: output within bf16 rounding of the f32 oracle block; the input gradient and all
    twelve parameter gradients within a cosine / norm band (bf16 operands against f32 arithmetic)."""
    blk = _block(D, H, ratio, 0.1, seed=B + D).to(DEV).eval()
    sd = _sd(blk)
    leaf = {k: v.cpu().clone().requires_grad_(True) for k, v in sd.items()}
    cfg = O.Cfg(wsi_embed_dim=8, rna_embed_dim=8, embed_dim=D, rna_num_heads=H, rna_mlp_ratio=ratio)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, D, generator=g)
    dy = torch.randn(B, D, generator=g)
    xr = x.clone().requires_grad_(True)
    want = O.rna_block(xr, leaf, "b", cfg)
    want.backward(dy)
    xg = x.to(DEV).requires_grad_(True)
    prec = Fn.POLICIES["bf16"]
    got = Fn.rna_block(xg, blk, prec, False)
    assert got is not None, "the fused path did not take this geometry"
    got.backward(dy.to(DEV))
    err = float((got.detach().cpu() - want.detach()).abs().max()) / float(want.detach().abs().max())
    assert err < 2e-2, err

    def band(a, b, name):
        a, b = a.detach().cpu().double().reshape(-1), b.detach().double().reshape(-1)
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))
        ratio_ = float(a.norm() / (b.norm() + 1e-300))
        assert cos > 0.995 and 0.97 < ratio_ < 1.03, (name, cos, ratio_)
    band(xg.grad, xr.grad, "x")
    for k, p in blk.named_parameters():
        band(p.grad, leaf["b." + k].grad, k)


def test_fused_block_equals_composed_ops_with_dropout(monkeypatch):
    """Train mode, p = 0.1: the fused Block and the composed op sequence draw the SAME Philox masks (same seed, same offsets
    in the same order: proj output, fc1 activation, fc2 output), so outputs and gradients agree to bf16 rounding of the
    intermediates; and the dropout offset advances by the same amount."""
    B, D, H = 16, 512, 8
    prec = Fn.POLICIES["bf16"]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, D, generator=g).to(DEV)
    dy = torch.randn(B, D, generator=g).to(DEV)
    res = []
    for fused in (True, False):
        blk = _block(D, H, 4.0, 0.1, seed=11).to(DEV).train()
        monkeypatch.setattr(Fn, "_RNA_FUSED", fused)
        Fn.manual_seed(77)
        Fn._dropout_state["base"] = None
        Fn._dropout_state["offset"] = 64                      # not at the start of the stream
        xg = x.clone().requires_grad_(True)
        y = blk(xg, prec)
        off = Fn._dropout_state["offset"]
        y.backward(dy)
        res.append((y.detach(), xg.grad, {k: p.grad.clone() for k, p in blk.named_parameters()}, off))
    (ya, dxa, ga, offa), (yb, dxb, gb, offb) = res
    assert offa == offb == 64 + 2 * B * D + B * 4 * D
    assert float((ya - yb).abs().max()) < 3e-2 * float(yb.abs().max())
    assert float((dxa - dxb).norm()) < 2e-2 * float(dxb.norm())
    for k in ga:
        assert float((ga[k] - gb[k]).norm()) < 3e-2 * float(gb[k].norm()) + 1e-6, k


def test_fused_block_under_device_dropout_base_redraws_masks():
    """Engine protocol (graph-safe dropout): the per-step base lives on the device; two calls with the same host offset but a
    bumped device base must draw different masks, and backward must regenerate the mask of its own forward."""
    B, D, H = 8, 256, 8
    prec = Fn.POLICIES["bf16"]
    blk = _block(D, H, 4.0, 0.5, seed=5).to(DEV).train()
    x = torch.randn(B, D, device=DEV)
    dev0 = x.device                     # indexed device: dropout_step_begin keeps its base only for an identical device
    Fn.manual_seed(9)
    Fn.dropout_step_begin(dev0)
    try:
        xg = x.clone().requires_grad_(True)
        y1 = blk(xg, prec)
        y1.sum().backward()
        g1 = xg.grad.clone()
        Fn.dropout_step_end()
        assert int(Fn._dropout_state["base"]) == 2 * B * D + B * 4 * D
        Fn.dropout_step_begin(dev0)
        y2 = blk(x, prec)
        Fn.dropout_step_end()
        assert not torch.equal(y1.detach(), y2.detach())
        assert torch.isfinite(g1).all() and float(g1.abs().sum()) > 0
    finally:
        Fn.dropout_device_base_off()
        Fn.manual_seed(0x5EED)


def test_whole_rna_branch_launch_count_and_parity_with_composed(monkeypatch):
    """c2's RNA branch (G = 2048, D = 512, depth 6 + 1 decoder block): with the fused blocks the branch's outputs equal the
    composed path's to bf16 rounding."""
    import mirror_amd.models as M
    torch.manual_seed(1)
    m = M.mirror(wsi_embed_dim=64, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=16, rna_encoder_depth=6, rna_mlp_ratio=4.0,
                 rna_num_heads=8, num_prototypes=10).to(DEV).eval()
    m.precision = "bf16"
    rna = torch.randn(16, 2048, device=DEV)
    noise = torch.rand(16, 512, device=DEV)
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(Fn, "_RNA_FUSED", fused)
        with torch.no_grad():
            outs.append([t.float() for t in m.rna_branch(rna, noise, 0.75)])
    for a, b in zip(*outs):
        assert float((a - b).abs().max()) <= 3e-2 * float(b.abs().max()) + 1e-6
