"""GPU: the hot path at BASELINE.json's FULL sizes (configs[1]: 4096 patch tokens x 1024-d + 2048 genes, D = 512, RNA depth
6; configs[3]: 8192 x 768-d padded slides) — a live oracle run at the benchmark shapes (B = 2: the CPU restatement needs a
few seconds) and size-independent properties where the oracle would be too slow: padding invariance of the mask path,
symmetry of the InfoNCE terms, linearity of the backward pass, run-to-run agreement of the graph-replayed training step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

import mirror_amd.models as M  # noqa: E402
from mirror_amd.losses import ClipLoss, MIRRORLoss  # noqa: E402
from oracle import mirror_oracle as O, synth  # noqa: E402
from tests.golden_util import DEFAULT_W  # noqa: E402

C2 = O.Cfg(wsi_embed_dim=1024, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=4096, rna_encoder_depth=6,
           rna_mlp_ratio=4.0, rna_num_heads=8)
C4 = O.Cfg(wsi_embed_dim=768, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=8192, rna_encoder_depth=6,
           rna_mlp_ratio=4.0, rna_num_heads=8)
W_KW = ("alignment_loss_weight", "wsi_retention_loss_weight", "rna_retention_loss_weight", "style_loss_weight",
        "cluster_loss_weight")
LOSS_RTOL = 1e-4        # BASELINE.json north_star: loss matching the CPU reference to 1e-4 rel (fp32 policy)
EMB_TOL = 2e-4          # projected embeddings, relative to the tensor's largest magnitude


def _model(cfg, sd, precision):
    m = M.mirror(wsi_embed_dim=cfg.wsi_embed_dim, rna_embed_dim=cfg.rna_embed_dim, embed_dim=cfg.embed_dim,
                 wsi_num_tokens=cfg.wsi_num_tokens, rna_encoder_depth=cfg.rna_encoder_depth, rna_mlp_ratio=cfg.rna_mlp_ratio,
                 rna_norm_layer="layernorm", rna_act_layer="gelu", rna_num_heads=cfg.rna_num_heads)
    m.load_state_dict(sd, strict=True)
    m.precision = precision
    return m.to(DEV).eval()


def test_c2_shapes_fp32_policy_matches_live_oracle():
    """configs[1] shapes, B = 2, fp32 policy, dropout off, the four noise draws injected: all 15 outputs and the six loss
    terms of the HIP path against the CPU oracle run on the same tensors."""
    sd = synth.synth_state_dict(synth.param_shapes(C2), 4242)
    wsi, rna, noise = synth.synth_batch(C2, 2, 5252)
    with O.exact_cpu_convs():
        ref = O.mirror_forward(sd, C2, wsi, rna, noise, 0.75, 0.75)
    ref_loss = [float(x) for x in O.mirror_loss(ref, DEFAULT_W)]
    outs = _model(C2, sd, "fp32")(wsi.to(DEV), rna.to(DEV), wsi_mask_ratio=0.75, rna_mask_ratio=0.75,
                                  noise={k: v.to(DEV) for k, v in noise.items()})
    for nm, a, b in zip(O.OUTPUT_NAMES, outs, ref):
        scale = max(float(b.abs().max()), 1e-6)
        err = float((a.detach().float().cpu() - b.float()).abs().max()) / scale
        assert err <= EMB_TOL, f"{nm}: {err:.3e}"
    got = [float(x.detach()) for x in MIRRORLoss()(*outs)]
    np.testing.assert_allclose(got, ref_loss, rtol=LOSS_RTOL)
    # the policies the bench runs in (bf16 MFMA; fp8 forward projections): reported accuracy band, not the parity gate
    for pol, band in (("bf16", 3e-2), ("fp8", 8e-2)):
        o = _model(C2, sd, pol)(wsi.to(DEV).to(torch.bfloat16), rna.to(DEV), wsi_mask_ratio=0.75, rna_mask_ratio=0.75,
                                noise={k: v.to(DEV) for k, v in noise.items()})
        rel = np.abs(np.array([float(x.detach()) for x in MIRRORLoss()(*o)]) - np.array(ref_loss)) / np.abs(np.array(ref_loss))
        assert (rel < band).all(), f"{pol}: loss rel err {rel}"


def test_c2_shapes_fp32_backward_matches_live_oracle():
    """Same shapes, now through the loss and back: the gradient of the total loss w.r.t. EVERY parameter (hand-written HIP
    backward kernels, pinv chain, fused optimizer glue excluded) against torch-CPU autograd through the oracle.  Norms to
    2e-3 relative (the tolerance of the golden-vector cases), the largest tensors also element-wise."""
    sd = synth.synth_state_dict(synth.param_shapes(C2), 99)
    wsi, rna, noise = synth.synth_batch(C2, 2, 100)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    with O.exact_cpu_convs():
        ref = O.mirror_forward(leaf, C2, wsi, rna, noise, 0.75, 0.75)
        O.mirror_loss(ref, DEFAULT_W)[0].backward()
    m = _model(C2, sd, "fp32")
    outs = m(wsi.to(DEV), rna.to(DEV), wsi_mask_ratio=0.75, rna_mask_ratio=0.75, noise={k: v.to(DEV) for k, v in noise.items()})
    MIRRORLoss()(*outs)[0].backward()
    bad = []
    gmax = max(float(v.grad.norm()) for v in leaf.values() if v.grad is not None)
    for k, p in m.named_parameters():
        r = leaf[k].grad
        rn = 0.0 if r is None else float(r.double().norm())
        gn = 0.0 if p.grad is None else float(p.grad.double().norm())
        if abs(gn - rn) > 2e-3 * rn + 1e-6 * gmax:
            bad.append(f"{k}: {gn:.6g} vs {rn:.6g}")
        elif r is not None and r.numel() >= 512 * 512:
            d = float((p.grad.cpu().double() - r.double()).norm())
            if d > 3e-3 * rn + 1e-6 * gmax:
                bad.append(f"{k}: |diff| {d:.3g} of {rn:.3g}")
    assert not bad, "; ".join(bad[:10])


def test_c4_shapes_masked_rows_do_not_reach_valid_rows_of_a_nystrom_layer():
    """configs[3] shapes (cls + 8192 + 89 wrap-around tokens = 8282 rows, 512-d, bf16 policy, fused mask-aware kernels): in a
    TransLayer the key-padding mask zeroes the normalised rows in front of to_qkv and masks all three similarity matrices,
    so whatever sits in the masked rows of the residual stream must not change any VALID row of the output — bit for bit.
    (The encoder as a whole does not have this property: PPEG's 7x7 convolution mixes neighbouring tokens, padded or not,
    exactly as in the reference.)"""
    from mirror_amd.functional import POLICIES
    sd = synth.synth_state_dict(synth.param_shapes(C4), 7)
    layer = _model(C4, sd, "bf16").wsi_encoder.layer1
    T = 1 + 8192 + 89
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, T, 512, generator=g).to(DEV)
    lens = torch.tensor([T - 3001, 2048 + 77])
    mask = (torch.arange(T)[None, :] < lens[:, None]).to(DEV)
    junk = torch.where(mask[..., None], x, torch.full_like(x, 37.5))
    with torch.no_grad():
        a = layer(x, POLICIES["bf16"], mask)
        b = layer(junk, POLICIES["bf16"], mask)
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert torch.equal(a[mask], b[mask])
    assert not torch.equal(a[~mask], b[~mask])           # the masked rows themselves keep their residual content


def test_clip_loss_is_symmetric_in_the_two_modalities_at_benchmark_batch():
    """ClipLoss (losses/mirror_loss.py:8-45) averages the image->text and text->image cross entropies: swapping the
    modalities must not change it (B = 16, D = 512: the logits product of the bench step)."""
    g = torch.Generator().manual_seed(3)
    x = torch.nn.functional.normalize(torch.randn(16, 512, generator=g), dim=1).to(DEV)
    y = torch.nn.functional.normalize(torch.randn(16, 512, generator=g), dim=1).to(DEV)
    s = torch.tensor(14.2857, device=DEV)
    a, b = ClipLoss()(x, y, s), ClipLoss()(y, x, s)
    assert abs(float(a) - float(b)) <= 1e-6 * abs(float(a))


def test_c2_backward_is_linear_in_the_loss_weights():
    """Reverse mode is linear: doubling every loss weight doubles every parameter gradient.  fp32 policy at the full c2
    shapes (B = 2) — every hand-written backward kernel of the step takes part."""
    sd = synth.synth_state_dict(synth.param_shapes(C2), 11)
    wsi, rna, noise = synth.synth_batch(C2, 2, 12)
    nz = {k: v.to(DEV) for k, v in noise.items()}
    grads = []
    for scale in (1.0, 2.0):
        m = _model(C2, sd, "fp32")
        outs = m(wsi.to(DEV), rna.to(DEV), noise=nz)
        MIRRORLoss(**{k: scale * w for k, w in zip(W_KW, DEFAULT_W)})(*outs)[0].backward()
        grads.append({k: p.grad.double() for k, p in m.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 100
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        tol = 2e-4 * float(b.abs().max()) + 1e-12          # f32 atomics order differs between the two runs
        assert float((2 * a - b).abs().max()) <= tol, k


def test_c2_training_step_replays_agree_between_runs():
    """Two engines, same seeds, the bench configuration (B = 16, bf16 policy, whole step as one HIP graph): the loss
    sequences agree to the noise of the f32 atomics (dropout masks, noise draws and Adam state are all device-resident and
    seeded, so nothing else may differ)."""
    from mirror_amd import functional as Fn
    from mirror_amd.engine import TrainEngine
    seqs = []
    for _ in range(2):
        torch.manual_seed(42)
        m = M.mirror(wsi_embed_dim=1024, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=4096, rna_encoder_depth=6,
                     rna_mlp_ratio=4.0, rna_norm_layer="layernorm", rna_act_layer="gelu", rna_num_heads=8).to(DEV).train()
        eng = TrainEngine(m, MIRRORLoss(), lr=2e-5, precision="bf16")
        Fn.manual_seed(1234)
        g = torch.Generator(device=DEV).manual_seed(1)
        wsi = torch.randn(16, 4096, 1024, device=DEV, generator=g).to(torch.bfloat16)
        rna = torch.randn(16, 2048, device=DEV, generator=g)
        torch.manual_seed(99)                        # the four noise draws of every step
        seq = [[float(x) for x in eng.step(wsi, rna)] for _ in range(5)]
        assert eng._graph is not None
        seqs.append(np.array(seq))
        del eng, m
        torch.cuda.empty_cache()
    assert np.isfinite(seqs[0]).all()
    np.testing.assert_allclose(seqs[0][:2], seqs[1][:2], rtol=2e-3)      # eager warm steps: same torch RNG stream
