"""CPU: pin the oracle (oracle/mirror_oracle.py) against golden vectors recorded from the
reference itself (tools/make_golden.py).  fp32 vs fp32: tolerances are rounding-level."""
import os

import numpy as np
import pytest
import torch

from oracle import mirror_oracle as O
from tests.golden_util import GOLDEN, ModelCase, TEMPLATE_W, DEFAULT_W


@pytest.mark.parametrize("name", ["tiny", "h12", "mid", "c1"])
def test_model_forward_and_loss_match_reference(name):
    case = ModelCase(name)
    torch.set_num_threads(8)
    sd = {k: v.clone().requires_grad_(True) for k, v in case.sd.items()}
    with O.exact_cpu_convs():
        _check_case(case, sd)


def _check_case(case, sd):
    outs = O.mirror_forward(sd, case.cfg, case.wsi, case.rna, case.noise, *case.ratios)
    case.check_outputs(outs, rtol=2e-4)
    lt = O.mirror_loss(outs, TEMPLATE_W)
    ld = O.mirror_loss([o.detach() for o in outs], DEFAULT_W)
    np.testing.assert_allclose([float(x.detach()) for x in lt], case.z["loss_template"], rtol=2e-5)
    np.testing.assert_allclose([float(x) for x in ld], case.z["loss_default"], rtol=2e-5)
    # gradients of the total (template-weighted) loss w.r.t. every parameter
    lt[0].backward()
    gn = np.array([0.0 if sd[k].grad is None else float(sd[k].grad.double().norm()) for k in case.keys])
    ref = case.z["grad_norm"]
    np.testing.assert_allclose(gn, ref, rtol=2e-3, atol=1e-6 * float(ref.max()))
    for k in case.keys:
        if f"grad/{k}" in case.z.files:
            g = case.z[f"grad/{k}"]
            tol = 2e-3 * max(float(np.abs(g).max()), 1e-7)
            np.testing.assert_allclose(sd[k].grad.numpy(), g, atol=tol, rtol=0)


def test_mirror_loss_matches_reference():
    z = np.load(os.path.join(GOLDEN, "golden_losses.npz"))
    for tag, w in (("default", DEFAULT_W), ("template", TEMPLATE_W)):
        ins = [torch.from_numpy(z[f"in/{nm}"]).clone() for nm in O.OUTPUT_NAMES]
        for t in ins:
            if f"grad_{tag}/x" is not None:
                t.requires_grad_(t.dtype.is_floating_point)
        out = O.mirror_loss(ins, w)
        np.testing.assert_allclose([float(x) for x in out], z[f"loss_{tag}"], rtol=1e-5)
        out[0].backward()
        for nm, t in zip(O.OUTPUT_NAMES, ins):
            key = f"grad_{tag}/{nm}"
            if key in z.files:
                g = z[key]
                np.testing.assert_allclose(t.grad.numpy(), g, atol=1e-5 * max(np.abs(g).max(), 1e-6), rtol=1e-4)
    w = torch.from_numpy(z["in/wsi_alignment_emb"])
    r = torch.from_numpy(z["in/rna_alignment_emb"])
    s = torch.from_numpy(z["in/logit_scale"])
    np.testing.assert_allclose(float(O.clip_loss(w, r, s)), float(z["clip_loss"]), rtol=1e-5)


def test_info_nce_matches_reference():
    z = np.load(os.path.join(GOLDEN, "golden_infonce.npz"))
    q0, k0 = torch.from_numpy(z["q"]), torch.from_numpy(z["k"])
    n = 0
    for key in z.files:
        if not key.startswith("loss/"):
            continue
        tag = key[5:]
        sym, red, tau = tag.split("_")
        q, k = q0.clone().requires_grad_(True), k0.clone().requires_grad_(True)
        out = O.info_nce(q, k, float(tau), red, sym == "sym1")
        np.testing.assert_allclose(out.detach().numpy(), z[key], rtol=2e-5, atol=1e-6)
        out.sum().backward()
        np.testing.assert_allclose(q.grad.numpy(), z[f"gq/{tag}"], rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(k.grad.numpy(), z[f"gk/{tag}"], rtol=1e-3, atol=1e-6)
        n += 1
    assert n == 12


def test_pinv_converges_to_inverse():
    """Property with no oracle needed (SURVEY §8c iii): a2 @ pinv(a2) -> I as iterations grow."""
    g = torch.Generator().manual_seed(3)
    a = (torch.randn(2, 3, 24, 24, generator=g) * 2).softmax(-1) + 0.5 * torch.eye(24)
    a = a / a.sum(-1, keepdim=True)
    e6 = (a @ O.pinv_iter(a, 6) - torch.eye(24)).abs().max()
    e20 = (a @ O.pinv_iter(a, 20) - torch.eye(24)).abs().max()
    assert e20 < 1e-4 and e20 <= e6


def test_nystrom_equals_softmax_attention_when_landmarks_are_tokens():
    """With l == 1 (m == n) and an exact inverse, Nystrom attention reduces to
    softmax(qk^T)v + res_conv(v); here checked as a1 @ pinv(a2) @ a3 ~= softmax(q k^T)."""
    g = torch.Generator().manual_seed(5)
    q = torch.randn(1, 1, 12, 8, generator=g) * 0.3
    k = torch.randn(1, 1, 12, 8, generator=g) * 0.3
    a = (q @ k.transpose(-1, -2)).softmax(-1)
    approx = a @ O.pinv_iter(a, 30) @ a
    assert (approx - a).abs().max() < 1e-3


def test_classifier_restatement_matches_reference():
    """Downstream MIRRORClassifier: oracle vs the reference's recorded predictions."""
    import os
    import numpy as np
    from oracle import synth
    from oracle import mirror_oracle as O
    from tools.make_golden import CLS_CFG
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_classifier.npz"))
    wsi, rna = torch.from_numpy(z["in/wsi"]), torch.from_numpy(z["in/rna"])
    for fusion in ("concat", "add"):
        keys = [str(k) for k in z[f"keys/{fusion}"]]
        assert keys == [k for k, _ in synth.classifier_param_shapes(CLS_CFG, 5, fusion)]
        sd = {k: torch.from_numpy(z[f"sd/{fusion}/{k}" if f"sd/{fusion}/{k}" in z.files else f"sd/concat/{k}"]) for k in keys}
        pred = O.classifier_forward(sd, CLS_CFG, wsi, rna, fusion)
        np.testing.assert_allclose(pred.numpy(), z[f"pred/{fusion}"], rtol=1e-5, atol=1e-6)
        if fusion == "add":
            np.testing.assert_allclose(O.classifier_forward(sd, CLS_CFG, wsi, None, fusion).numpy(), z["pred/add_wsi_only"],
                                       rtol=1e-5, atol=1e-6)


def test_dataset_getitem_restatement_matches_reference():
    """the fixed-N resampling of TCGAWSIRNAPretrainDataset.__getitem__ recorded from the reference
    (tools/make_golden.py gen_datafeed), numpy's global RNG seeded the same way."""
    z = np.load(os.path.join(GOLDEN, "golden_datafeed.npz"))
    N = int(z["num_tokens"])
    slides = [torch.from_numpy(z[f"slide/{k}"]) for k in range(3)]
    rna = z["rna"]
    np.random.seed(int(z["seed"]))
    for j, k in enumerate(z["order"]):
        w, r, idx = O.dataset_getitem(slides[int(k)], rna[int(k)].astype(np.float64), N)
        assert np.array_equal(idx, z[f"out/{j}/idx"])
        assert np.array_equal(w.numpy(), z[f"out/{j}/wsi"]) and np.array_equal(r.numpy(), z[f"out/{j}/rna"])
        n = slides[int(k)].shape[0]
        assert (len(set(idx.tolist())) == N) == (n >= N) or n < N      # without replacement <=> no duplicates when long enough


def test_masked_nystrom_restatement_matches_package_standin():
    """BASELINE config 4 (variable-length slides): the package's key-padding `mask` path.  The oracle's restatement against
    the literal stand-in of nystrom_attention 0.0.14 (tools/oracle_shims.py) — partially masked rows, a fully masked
    landmark group, a fully masked query row (uniform attention), front padding."""
    from tools.oracle_shims import NystromAttention
    torch.manual_seed(3)
    D, h, n = 32, 8, 37
    cfg = O.Cfg(wsi_embed_dim=8, rna_embed_dim=8, embed_dim=D)
    mod = NystromAttention(dim=D, dim_head=D // h, heads=h, num_landmarks=D // 2, pinv_iterations=6, residual=True, dropout=0.1).eval()
    sd = {"attn." + k: v.detach() for k, v in mod.state_dict().items()}
    x = torch.randn(3, n, D)
    lens = [37, 20, 5]
    mask = torch.stack([torch.arange(n) < L for L in lens])
    with torch.no_grad():
        want = mod(x, mask=mask)
        got = O.nystrom_attention(x, sd, "attn", cfg, mask)
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), float((got - want).abs().max())
        assert not torch.allclose(got, O.nystrom_attention(x, sd, "attn", cfg, None), atol=1e-3)      # the mask matters
        # (an all-True mask is NOT the unmasked path: with a mask the front-padding rows are masked out of the landmark
        #  means and the three softmaxes, without one they take part as zero rows — the package's quirk, kept)


def test_oracle_train_mode_dropout_masks_are_consumed_in_program_order():
    """noise["dropout"]: unit multipliers reproduce the eval-mode run exactly; real masks change the result; the RNA Block's
    three sites equal torch's own nn.Dropout arithmetic (x * keep / (1 - p)) when applied by hand (models/mirror.py:101, :142)."""
    from oracle import synth
    cfg = O.Cfg(wsi_embed_dim=16, rna_embed_dim=12, embed_dim=32, wsi_num_tokens=20, rna_encoder_depth=2, rna_num_heads=8,
                style_mlp_hidden_dim=16, style_mlp_out_dim=8, style_latent_dim=4, num_prototypes=9)
    sd = synth.synth_state_dict(synth.param_shapes(cfg), seed=3)
    wsi, rna, noise = synth.synth_batch(cfg, batch=2, seed=4)
    base = O.mirror_forward(sd, cfg, wsi, rna, noise)
    B, D, Hh = 2, 32, sd["rna_encoder.blocks.0.mlp.fc1.weight"].shape[0]
    n = 20 + 1 + (25 - 20)
    shapes_w = [(B, n, D), (B, n, D), (B, 21, D)]
    shapes_r = [(B, D), (B, Hh), (B, D)] * 3
    ones = {"wsi": [torch.ones(s) for s in shapes_w], "rna": [torch.ones(s) for s in shapes_r]}
    same = O.mirror_forward(sd, cfg, wsi, rna, dict(noise, dropout=ones))
    for a, b in zip(base, same):
        assert torch.equal(a, b)
    g = torch.Generator().manual_seed(5)
    keep = lambda s: (torch.rand(s, generator=g) >= 0.1).float() / 0.9      # noqa: E731
    masks = {"wsi": [keep(s) for s in shapes_w], "rna": [keep(s) for s in shapes_r]}
    drop = O.mirror_forward(sd, cfg, wsi, rna, dict(noise, dropout=masks))
    assert not torch.allclose(drop[1], base[1], atol=1e-4) and not torch.allclose(drop[8], base[8], atol=1e-4)
    # one Block by hand
    x = torch.randn(B, D, generator=g)
    p = "rna_encoder.blocks.0"
    m3 = masks["rna"][:3]
    got = O.rna_block(x, sd, p, cfg, iter(m3))
    h = x + O.rna_attention(O._ln(x, sd, p + ".norm1", 1e-6), sd, p + ".attn", 8) * m3[0]
    y = torch.nn.functional.gelu(O._linear(O._ln(h, sd, p + ".norm2", 1e-6), sd, p + ".mlp.fc1")) * m3[1]
    want = h + O._linear(y, sd, p + ".mlp.fc2") * m3[2]
    assert torch.allclose(got, want, atol=1e-6)
    with pytest.raises(AssertionError):                                       # a mask too many is an error, not silently ignored
        O.mirror_forward(sd, cfg, wsi, rna, dict(noise, dropout={"wsi": masks["wsi"] + [torch.ones(1)], "rna": masks["rna"]}))
