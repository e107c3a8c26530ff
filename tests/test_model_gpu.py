"""GPU parity proper: the HIP path (through the C ABI) vs golden vectors recorded from the reference and
vs the CPU oracle on the same seeded inputs.  fp32 policy, eval mode (dropout off), injected noise.

Tolerances (SURVEY.md §8d / BASELINE.json): every loss term within 1e-4 relative; embeddings within
1e-4 x max|ref| (max-abs); gradients within 2e-3 (norms) of the reference's autograd."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mirror_amd.models as M  # noqa: E402
from mirror_amd.losses import ClipLoss, InfoNCE, MIRRORLoss  # noqa: E402
from oracle import mirror_oracle as O  # noqa: E402
from tests.golden_util import GOLDEN, ModelCase, TEMPLATE_W, DEFAULT_W  # noqa: E402

DEV = "cuda"
LOSS_RTOL = 1e-4
EMB_TOL = 1e-4
W_KW = ("alignment_loss_weight", "wsi_retention_loss_weight", "rna_retention_loss_weight", "style_loss_weight",
        "cluster_loss_weight")


def build(case: ModelCase, precision="fp32"):
    c = case.cfg
    model = M.mirror(
        wsi_embed_dim=c.wsi_embed_dim, rna_embed_dim=c.rna_embed_dim, embed_dim=c.embed_dim,
        wsi_num_tokens=c.wsi_num_tokens, wsi_retention_decoder_depth=c.wsi_retention_decoder_depth,
        rna_encoder_depth=c.rna_encoder_depth, rna_mlp_ratio=c.rna_mlp_ratio, rna_norm_layer="layernorm",
        rna_act_layer="gelu", rna_retention_decoder_depth=c.rna_retention_decoder_depth,
        style_mlp_hidden_dim=c.style_mlp_hidden_dim, style_mlp_out_dim=c.style_mlp_out_dim,
        style_latent_dim=c.style_latent_dim, num_prototypes=c.num_prototypes, rna_num_heads=c.rna_num_heads)
    model.load_state_dict(case.sd, strict=True)
    model.precision = precision
    return model.to(DEV).eval()


def run(case, model):
    noise = {k: v.to(DEV) for k, v in case.noise.items()}
    return model(case.wsi.to(DEV), case.rna.to(DEV), wsi_mask_ratio=case.ratios[0], rna_mask_ratio=case.ratios[1],
                 noise=noise)


@pytest.mark.parametrize("name", ["tiny", "h12", "mid", "c1"])
def test_forward_loss_backward_match_reference_golden(name):
    case = ModelCase(name)
    model = build(case)
    outs = run(case, model)
    assert len(outs) == 15
    case.check_outputs(outs, rtol=EMB_TOL)
    lt = MIRRORLoss(**dict(zip(W_KW, TEMPLATE_W)))(*outs)
    ld = MIRRORLoss()(*[o.detach() for o in outs])
    got_t = np.array([float(x.detach()) for x in lt])
    got_d = np.array([float(x.detach()) for x in ld])
    np.testing.assert_allclose(got_t, case.z["loss_template"], rtol=LOSS_RTOL)
    np.testing.assert_allclose(got_d, case.z["loss_default"], rtol=LOSS_RTOL)
    lt[0].backward()
    params = dict(model.named_parameters())
    gn = np.array([0.0 if params[k].grad is None else float(params[k].grad.double().norm()) for k in case.keys])
    ref = case.z["grad_norm"]
    bad = np.abs(gn - ref) > 2e-3 * ref + 1e-6 * ref.max()
    assert not bad.any(), "grad-norm mismatch: " + ", ".join(
        f"{k}: {a:.6g} vs {b:.6g}" for k, a, b, f in zip(case.keys, gn, ref, bad) if f)
    for k in case.keys:
        if f"grad/{k}" in case.z.files:
            g = case.z[f"grad/{k}"]
            tol = 2e-3 * max(float(np.abs(g).max()), 1e-7)
            np.testing.assert_allclose(params[k].grad.cpu().numpy(), g, atol=tol, rtol=0, err_msg=k)


def test_hip_path_matches_cpu_oracle_live():
    """Same seeded inputs through the oracle (CPU) and the HIP path, no fixture in between."""
    case = ModelCase("mid")
    outs_ref = O.mirror_forward(case.sd, case.cfg, case.wsi, case.rna, case.noise, *case.ratios)
    model = build(case)
    outs = run(case, model)
    for nm, a, b in zip(O.OUTPUT_NAMES, outs, outs_ref):
        scale = max(float(b.abs().max()), 1e-6)
        err = float((a.detach().cpu() - b).abs().max()) / scale
        assert err <= EMB_TOL, f"{nm}: {err:.3e}"
    got = [float(x.detach()) for x in MIRRORLoss()(*outs)]
    ref = [float(x) for x in O.mirror_loss(outs_ref, DEFAULT_W)]
    np.testing.assert_allclose(got, ref, rtol=LOSS_RTOL)


def test_mirror_loss_module_matches_reference_golden():
    z = np.load(os.path.join(GOLDEN, "golden_losses.npz"))
    for tag, w in (("default", DEFAULT_W), ("template", TEMPLATE_W)):
        ins = []
        for nm in O.OUTPUT_NAMES:
            t = torch.from_numpy(z[f"in/{nm}"]).to(DEV)
            ins.append(t.requires_grad_(f"grad_{tag}/{nm}" in z.files))
        out = MIRRORLoss(**dict(zip(W_KW, w)))(*ins)
        np.testing.assert_allclose([float(x.detach()) for x in out], z[f"loss_{tag}"], rtol=LOSS_RTOL)
        out[0].backward()
        for nm, t in zip(O.OUTPUT_NAMES, ins):
            key = f"grad_{tag}/{nm}"
            if key in z.files:
                g = z[key]
                np.testing.assert_allclose(t.grad.cpu().numpy(), g, atol=2e-4 * max(np.abs(g).max(), 1e-6), rtol=1e-3,
                                           err_msg=f"{tag}:{nm}")
    cl = ClipLoss()(torch.from_numpy(z["in/wsi_alignment_emb"]).to(DEV), torch.from_numpy(z["in/rna_alignment_emb"]).to(DEV),
                    torch.from_numpy(z["in/logit_scale"]).to(DEV))
    np.testing.assert_allclose(float(cl), float(z["clip_loss"]), rtol=LOSS_RTOL)


def _loss_inputs(B, N, F, G, D, P, S, seed, pred_dtype=torch.float32):
    g = torch.Generator(device=DEV).manual_seed(seed)
    r = lambda *sh: torch.randn(*sh, device=DEV, generator=g)
    nrm = lambda t: t / t.norm(dim=-1, keepdim=True)
    ins = [nrm(r(B, D)), r(B, N, F).to(pred_dtype), r(B, N, F), (torch.rand(B, N, device=DEV, generator=g) < 0.75).float(), r(B, P),
           r(B, S), 0.3 * r(B, S), nrm(r(B, D)), r(B, G), r(B, G), (torch.rand(B, G, device=DEV, generator=g) < 0.75).float(), r(B, P),
           r(B, S), 0.3 * r(B, S), torch.tensor(14.3, device=DEV)]
    grad = (0, 1, 2, 4, 5, 6, 7, 8, 9, 11, 12, 13, 14)        # everything but the masks
    return [t.requires_grad_(i in grad) for i, t in enumerate(ins)], grad


@pytest.mark.parametrize("shape", [(16, 64, 32, 2048, 512, 3000, 128), (5, 7, 12, 77, 33, 70, 9), (32, 3, 8, 300, 256, 257, 16),
                                   (1, 4, 4, 5, 8, 3, 2), (3, 2, 4, 9, 16, 4200, 4)])
def test_one_launch_loss_terms_match_the_composed_terms(shape, monkeypatch):
    """mh_loss_terms_fwd / _bwd (alignment + RNA retention + style + cluster + total in one launch each way) against the
    per-term kernels that the golden-vector and oracle tests pin: all six results and every input gradient."""
    from mirror_amd import functional as Fn
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(Fn, "_LOSS_FUSED", fused)
        ins, grad = _loss_inputs(*shape, seed=3, pred_dtype=torch.bfloat16 if shape[0] == 16 else torch.float32)
        out = MIRRORLoss(**dict(zip(W_KW, TEMPLATE_W)))(*ins)
        assert (out[0].grad_fn.name().startswith("MirrorLossTermsFn")) == fused
        (out[0] * 0.5).backward()
        res[fused] = ([float(x.detach()) for x in out], [ins[i].grad.float().cpu().numpy() for i in grad])
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=2e-6, atol=1e-7)
    for i, a, b in zip(grad, res[True][1], res[False][1]):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6 * max(float(np.abs(b).max()), 1e-30), err_msg=O.OUTPUT_NAMES[i])


def test_one_launch_loss_terms_single_term_and_external_alignment(monkeypatch):
    """The two side doors of the fused loss: a backward through ONE returned term (the reference returns attached terms,
    losses/mirror_loss.py:129-135), and the gathered-batch form where the caller supplies the alignment term."""
    from mirror_amd import functional as Fn
    shape = (6, 5, 8, 40, 64, 50, 8)
    got = {}
    for fused in (True, False):
        monkeypatch.setattr(Fn, "_LOSS_FUSED", fused)
        ins, grad = _loss_inputs(*shape, seed=11)
        out = MIRRORLoss()(*ins)
        (out[0] + 3.0 * out[1] + 0.5 * out[3] + 2.0 * out[5] + 0.25 * out[2]).backward()
        got[fused] = [ins[i].grad.cpu().numpy() for i in grad]
    for i, a, b in zip(grad, got[True], got[False]):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6 * max(float(np.abs(b).max()), 1e-30), err_msg=O.OUTPUT_NAMES[i])
    # external alignment term: total and gradients equal the local form when the caller's term IS the local ClipLoss
    monkeypatch.setattr(Fn, "_LOSS_FUSED", True)
    ins, grad = _loss_inputs(*shape, seed=11)
    w = (0.5, 0.1, 0.1, 0.1, 0.1, 0.2)
    ext = ClipLoss()(ins[0], ins[7], ins[14]).reshape(())
    out = Fn.MirrorLossTermsFn.apply(w, None, None, None, ext, ins[1], ins[2], ins[3], ins[1].shape[-1], None, ins[8], ins[9], ins[10],
                                     ins[5], ins[6], ins[12], ins[13], ins[4], ins[11])
    ins2, _ = _loss_inputs(*shape, seed=11)
    ref = MIRRORLoss()(*ins2)
    np.testing.assert_allclose([float(x.detach()) for x in out], [float(x.detach()) for x in ref], rtol=2e-6)
    out[0].backward()
    ref[0].backward()
    for i in grad:
        np.testing.assert_allclose(ins[i].grad.cpu().numpy(), ins2[i].grad.cpu().numpy(), rtol=2e-5,
                                   atol=2e-6 * max(float(ins2[i].grad.abs().max()), 1e-30), err_msg=O.OUTPUT_NAMES[i])


def test_info_nce_module_matches_reference_golden():
    z = np.load(os.path.join(GOLDEN, "golden_infonce.npz"))
    q0, k0 = torch.from_numpy(z["q"]).to(DEV), torch.from_numpy(z["k"]).to(DEV)
    n = 0
    for key in z.files:
        if not key.startswith("loss/"):
            continue
        tag = key[5:]
        sym, red, tau = tag.split("_")
        q, k = q0.clone().requires_grad_(True), k0.clone().requires_grad_(True)
        out = InfoNCE(temperature=float(tau), reduction=red, symmetric=(sym == "sym1"))(q, k)
        np.testing.assert_allclose(out.detach().cpu().numpy(), z[key], rtol=LOSS_RTOL, atol=1e-6, err_msg=tag)
        out.sum().backward()
        np.testing.assert_allclose(q.grad.cpu().numpy(), z[f"gq/{tag}"], rtol=2e-3, atol=2e-6, err_msg=tag)
        np.testing.assert_allclose(k.grad.cpu().numpy(), z[f"gk/{tag}"], rtol=2e-3, atol=2e-6, err_msg=tag)
        n += 1
    assert n == 12
    with pytest.raises(ValueError):
        InfoNCE()(q0[:, None], k0)
    with pytest.raises(ValueError):
        InfoNCE()(q0[:5], k0)


def test_bf16_policy_is_close_and_runs_train_mode():
    """bf16 MFMA policy: reported accuracy band (not the fp32 parity gate) + a train-mode step with dropout."""
    case = ModelCase("c1")
    model = build(case, precision="bf16")
    outs = run(case, model)
    got = np.array([float(x.detach()) for x in MIRRORLoss(**dict(zip(W_KW, TEMPLATE_W)))(*outs)])
    ref = case.z["loss_template"]
    rel = np.abs(got - ref) / np.abs(ref)
    assert (rel < 5e-2).all(), f"bf16 loss rel err {rel}"
    model.train()
    outs = run(case, model)
    loss = MIRRORLoss()(*outs)[0]
    loss.backward()
    assert torch.isfinite(loss)
    for k, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k


def test_model_refuses_cpu_tensors():
    case = ModelCase("tiny")
    model = build(case)
    from mirror_amd import MirrorHipError
    with pytest.raises(MirrorHipError):
        model(case.wsi, case.rna)


@pytest.mark.parametrize("fusion", ["concat", "add"])
def test_classifier_matches_reference_golden(fusion):
    """MIRRORClassifier (downstream encoders), fp32 policy, eval mode: predictions and per-parameter
    gradient norms of sum(pred^2) against the reference's recording; state-dict keys load strictly."""
    import os
    import numpy as np
    import mirror_amd.models as M
    from tools.make_golden import CLS_CFG as cfg
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_classifier.npz"))
    keys = [str(k) for k in z[f"keys/{fusion}"]]
    sd = {k: torch.from_numpy(z[f"sd/{fusion}/{k}" if f"sd/{fusion}/{k}" in z.files else f"sd/concat/{k}"]) for k in keys}
    model = M.create_model("mirror_classifier", wsi_embed_dim=cfg.wsi_embed_dim, rna_embed_dim=cfg.rna_embed_dim,
                           embed_dim=cfg.embed_dim, num_classes=5, rna_encoder_depth=cfg.rna_encoder_depth,
                           rna_mlp_ratio=cfg.rna_mlp_ratio, rna_norm_layer="layernorm", rna_act_layer="gelu", fusion=fusion,
                           an_unknown_kwarg=1)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    model.precision = "fp32"
    wsi, rna = torch.from_numpy(z["in/wsi"]).cuda(), torch.from_numpy(z["in/rna"]).cuda()
    pred = model(wsi, rna)
    want = z[f"pred/{fusion}"]
    err = float(np.abs(pred.detach().cpu().numpy() - want).max()) / float(np.abs(want).max())
    assert err <= 1e-4, err
    if fusion == "add":
        pw = model(wsi, None).detach().cpu().numpy()
        assert float(np.abs(pw - z["pred/add_wsi_only"]).max()) <= 1e-4 * float(np.abs(z["pred/add_wsi_only"]).max())
    pred.square().sum().backward()
    params = dict(model.named_parameters())
    for k, gn in zip(keys, z[f"grad_norm/{fusion}"]):
        got = float(params[k].grad.double().norm())
        assert abs(got - gn) <= 2e-3 * max(gn, 1e-3), (k, got, gn)


def _kp_mask(case, seed=5):
    g = torch.Generator().manual_seed(seed)
    N = case.cfg.wsi_num_tokens
    lens = torch.randint(max(1, N // 8), N + 1, (case.batch,), generator=g)
    lens[0] = N                                                  # one full-length slide
    return torch.arange(N)[None, :] < lens[:, None]


@pytest.mark.parametrize("name", ["tiny", "mid"])
def test_key_padding_mask_path_matches_oracle(name):
    """BASELINE config 4 (variable-length slides, padded + bool mask): fp32 HIP path vs the CPU oracle carrying the
    package's key-padding mask through every Nystrom layer — outputs, losses and gradients."""
    case = ModelCase(name)
    mask = _kp_mask(case)
    sd = {k: v.clone().requires_grad_(True) for k, v in case.sd.items()}
    with O.exact_cpu_convs():       # torch's oneDNN depthwise-conv weight gradient is wrong at some shapes (tools/make_golden.py)
        outs_ref = O.mirror_forward(sd, case.cfg, case.wsi, case.rna, case.noise, *case.ratios, wsi_key_padding_mask=mask)
        loss_ref = O.mirror_loss(outs_ref, DEFAULT_W)
        loss_ref[0].backward()
    model = build(case)
    noise = {k: v.to(DEV) for k, v in case.noise.items()}
    outs = model(case.wsi.to(DEV), case.rna.to(DEV), wsi_mask_ratio=case.ratios[0], rna_mask_ratio=case.ratios[1], noise=noise,
                 wsi_key_padding_mask=mask.to(DEV))
    for nm, a, b in zip(O.OUTPUT_NAMES, outs, outs_ref):
        scale = max(float(b.abs().max()), 1e-6)
        err = float((a.detach().cpu() - b.detach()).abs().max()) / scale
        assert err <= EMB_TOL, f"{nm}: {err:.3e}"
    losses = MIRRORLoss()(*outs)
    np.testing.assert_allclose([float(x.detach()) for x in losses], [float(x.detach()) for x in loss_ref], rtol=LOSS_RTOL)
    losses[0].backward()
    params = dict(model.named_parameters())
    for k in case.keys:
        ref = sd[k].grad
        if ref is None:
            continue
        got = params[k].grad
        tol = 2e-3 * max(float(ref.abs().max()), 1e-7)
        err = float((got.cpu() - ref).abs().max())
        assert err <= tol, f"{k}: max-abs grad error {err:.3e} > {tol:.3e}"
    # the unmasked outputs differ: the mask is really applied
    plain = run(case, build(case))
    assert float((plain[2] - outs[2]).abs().max()) > 1e-3


def test_key_padding_mask_bf16_policy_runs_and_is_close():
    case = ModelCase("c1")
    mask = _kp_mask(case)
    outs_ref = O.mirror_forward(case.sd, case.cfg, case.wsi, case.rna, case.noise, *case.ratios, wsi_key_padding_mask=mask)
    ref = [float(x) for x in O.mirror_loss(outs_ref, DEFAULT_W)]
    model = build(case, "bf16")
    noise = {k: v.to(DEV) for k, v in case.noise.items()}
    outs = model(case.wsi.to(DEV).bfloat16(), case.rna.to(DEV), wsi_mask_ratio=case.ratios[0], rna_mask_ratio=case.ratios[1],
                 noise=noise, wsi_key_padding_mask=mask.to(DEV))
    got = [float(x.detach()) for x in MIRRORLoss()(*outs)]
    np.testing.assert_allclose(got, ref, rtol=5e-2)
    MIRRORLoss()(*outs)[0].backward()


def test_fp8_forward_policy_is_close_and_trains():
    """BASELINE config 5: e4m3 MFMA operands for the forward WSI projections, bf16 backward.  Loss terms stay near the
    oracle (reported, not the parity gate) and a training step runs."""
    from mirror_amd.engine import TrainEngine
    case = ModelCase("c1")
    outs_ref = O.mirror_forward(case.sd, case.cfg, case.wsi, case.rna, case.noise, *case.ratios)
    ref = np.array([float(x) for x in O.mirror_loss(outs_ref, DEFAULT_W)])
    model = build(case, "fp8")
    noise = {k: v.to(DEV) for k, v in case.noise.items()}
    used = []
    orig = M.mirror.__globals__["Fn"].K.gemm_fp8
    M.mirror.__globals__["Fn"].K.gemm_fp8 = lambda *a, **k: (used.append(1), orig(*a, **k))[1]
    try:
        outs = model(case.wsi.to(DEV).bfloat16(), case.rna.to(DEV), wsi_mask_ratio=case.ratios[0], rna_mask_ratio=case.ratios[1],
                     noise=noise)
    finally:
        M.mirror.__globals__["Fn"].K.gemm_fp8 = orig
    assert len(used) >= 6, f"fp8 products launched: {len(used)}"       # _fc1, 3 x to_qkv, 3 x to_out, retention embed / head
    got = np.array([float(x.detach()) for x in MIRRORLoss()(*outs)])
    np.testing.assert_allclose(got, ref, rtol=8e-2)
    eng = TrainEngine(model.train(), MIRRORLoss(), lr=1e-4, precision="fp8", graph=False)
    l0 = eng.step(case.wsi.to(DEV).bfloat16(), case.rna.to(DEV))
    assert all(torch.isfinite(x) for x in l0)
    # delayed scaling (one-pass quantisation with the previous step's amax) takes over from the third step of a call site: the
    # loss trajectory must stay where exact per-step scaling puts it (same weights, same dropout seeds, same injected noise)
    from mirror_amd import functional as Fn

    def run(delayed: bool):
        m = build(case, "fp8")
        e = TrainEngine(m.train(), MIRRORLoss(), lr=1e-4, precision="fp8", graph=False, seed=5)
        launches, fused = [], []
        orig_d, orig_s, orig_l = Fn.K.quant_fp8_delayed, Fn.fp8_delayed_scaling, Fn.K.layernorm_fwd_q8
        Fn.K.quant_fp8_delayed = lambda *a, **k: (launches.append(1), orig_d(*a, **k))[1]
        Fn.K.layernorm_fwd_q8 = lambda *a, **k: (fused.append(1), orig_l(*a, **k))[1]
        if not delayed:
            Fn.fp8_delayed_scaling = lambda *a, **k: orig_s(None)
        try:
            traj = [[float(x) for x in e.step(case.wsi.to(DEV).bfloat16(), case.rna.to(DEV), noise=noise)] for _ in range(5)]
        finally:
            Fn.K.quant_fp8_delayed, Fn.fp8_delayed_scaling, Fn.K.layernorm_fwd_q8 = orig_d, orig_s, orig_l
        return np.array(traj), len(launches), len(fused)
    t_del, n_del, n_ln = run(True)
    t_exact, n_exact, n_ln_exact = run(False)
    # steps 3..5: the three LayerNorms in front of to_qkv write the e4m3 copy themselves (and, where the fused attention kernels
    # run, attn1 does for to_out); the other activation sites and the weights take the one-pass quantisation launch
    assert n_exact == 0 and n_ln_exact == 0 and n_ln == 3 * 3 and n_del >= 3 * (2 + 8), (n_exact, n_del, n_ln)
    assert np.isfinite(t_del).all()
    # the first two steps ARE the exact path: step 1 agrees to rounding; step 2 sits behind one Adam update, whose first step
    # moves every weight by lr * sign(g), so parameters whose gradient is f32-atomics noise around zero make it run-to-run
    # bimodal (3e-3 here, with or without delayed scaling)
    np.testing.assert_allclose(t_del[0], t_exact[0], rtol=1e-5)
    np.testing.assert_allclose(t_del[1:], t_exact[1:], rtol=5e-2)
