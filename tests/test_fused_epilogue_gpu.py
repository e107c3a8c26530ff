"""GPU: the fused projection epilogues (mh_gemm_epi, include/mirror_hip.h) against the composed ops they replace.

Each fused launch must be BIT-IDENTICAL to the composed HIP path — the Linear's result is rounded to bf16 in the epilogue exactly
where the composed path stores it as bf16, and the dropout mask is the one mh_dropout draws for the same (seed, offset, element)
— and the composed path is what the oracle parity tests (test_model_gpu.py, test_bench_path_gpu.py) pin.  Shapes cover a ragged
last row tile, row windows that cross batch boundaries inside a tile, and both weight layouts the step uses.
Reference: [3P] to_out + Dropout + the residual add (models/mirror.py:312-313); retention_embed + random_masking + the
position embedding (:636-643, :690-693); retention_head + the masked MSE (losses/mirror_loss.py:98-103)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

bf16, f32 = torch.bfloat16, torch.float32


def _w(n, k, seed):
    g = torch.Generator().manual_seed(seed)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda().requires_grad_(True)
    b = (torch.randn(n, generator=g) * 0.1).cuda().requires_grad_(True)
    return w, b


@pytest.mark.parametrize("Bn,T,r0,R,Kd,N", [(3, 600, 88, 512, 512, 512), (2, 700, 1, 699, 256, 256), (4, 300, 0, 300, 512, 256)])
def test_to_out_dropout_residual_epilogue_equals_composed(Bn, T, r0, R, Kd, N):
    from mirror_amd import functional as Fn
    prec = Fn.POLICIES["bf16"]
    g = torch.Generator().manual_seed(1)
    core0 = torch.randn(Bn, T, Kd, generator=g).cuda().to(bf16)
    resid0 = torch.randn(Bn, R, N, generator=g).cuda()
    up = torch.randn(Bn, R, N, generator=g).cuda()
    w, b = _w(N, Kd, 2)
    res = []
    for fused in (True, False):
        Fn.manual_seed(77)
        Fn._dropout_state["offset"] = 4096          # a non-zero call offset
        core, resid = core0.clone().requires_grad_(True), resid0.clone().requires_grad_(True)
        w.grad = b.grad = None
        Fn._res_grads.clear()
        if fused:
            assert Fn.K.linear_fused_ok(core, Fn.shadow(w, prec), (r0, R))
            y = Fn.ToOutDropAddFn.apply(resid, core, w, b, r0, R, 0.1, prec)
        else:
            y = Fn.dropout_add(resid, Fn.LinearRowsFn.apply(core, w, b, r0, R, prec, bf16), 0.1, True, lite=True)
        y.backward(up.clone())      # the residual add hands its upstream gradient on in place: a fresh one per run
        res.append((y.detach().clone(), core.grad.clone(), resid.grad.clone(), w.grad.clone(), b.grad.clone(), Fn._dropout_state["offset"]))
    torch.cuda.synchronize()
    (y1, dc1, dr1, dw1, db1, o1), (y2, dc2, dr2, dw2, db2, o2) = res
    assert o1 == o2, "both forms consume the same dropout offsets"
    assert torch.equal(y1, y2), float((y1 - y2).abs().max())
    assert float((y1 == resid0).float().mean()) > 0.05, "dropout zeroes ~10 % of the projection"
    assert torch.equal(dc1, dc2) and torch.equal(dr1, dr2)
    assert float((db1 - db2).abs().max()) <= 1e-5 * float(db2.abs().max())        # column sums: f32 atomics, order may differ
    assert float((dw1 - dw2).abs().max()) <= 1e-5 * float(dw2.abs().max())        # split-K f32 sums: order may differ


@pytest.mark.parametrize("Bn,T,Kd,N", [(3, 257, 512, 512), (2, 1025, 256, 256)])
def test_retention_embed_mask_pos_epilogue_equals_composed(Bn, T, Kd, N):
    from mirror_amd import functional as Fn
    prec = Fn.POLICIES["bf16"]
    g = torch.Generator().manual_seed(3)
    h32 = torch.randn(Bn, T, Kd, generator=g).cuda()
    mask = (torch.rand(Bn, T - 1, generator=g) < 0.75).float().cuda()
    token = (torch.randn(1, 1, N, generator=g) * 0.02).cuda().requires_grad_(True)
    pos = (torch.randn(1, T, N, generator=g) * 0.02).cuda().requires_grad_(True)
    up = torch.randn(Bn, T, N, generator=g).cuda()
    w, b = _w(N, Kd, 4)
    res = []
    for fused in (True, False):
        h = h32.clone().requires_grad_(True)
        for p in (w, b, token, pos):
            p.grad = None
        if fused:
            h._bf16 = h32.to(bf16)
            y = Fn.embed_mask_pos(h, w, b, mask, token, pos, 1, prec)
            assert y.grad_fn.__class__.__name__.startswith("EmbedMaskPosFn")
        else:
            y = Fn.MaskApplyFn.apply(Fn.linear(h, w, b, prec=prec), mask, token, pos, 1, False)
        y.backward(up.clone())      # the residual add hands its upstream gradient on in place: a fresh one per run
        res.append([y.detach().clone()] + [t.grad.clone() for t in (h, w, b, token, pos)])
    torch.cuda.synchronize()
    for i, (a, c) in enumerate(zip(*res)):
        if i in (1, 2, 3, 4, 5):    # dW, db, dtoken, dpos: f32 sums whose order may differ (split-K, atomics); dh: the ragged rows go
            assert float((a - c).abs().max()) <= 1e-5 * float(c.abs().max())      # through the weight-streaming kernel (f32 FMA order)
        else:
            assert torch.equal(a, c), (i, float((a.float() - c.float()).abs().max()))


def test_layernorm_dual_writes_the_bf16_copy():
    from mirror_amd import functional as Fn
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 70, 512, generator=g).cuda().requires_grad_(True)
    gm, bt = torch.randn(512, generator=g).cuda().requires_grad_(True), torch.randn(512, generator=g).cuda().requires_grad_(True)
    y = Fn.layer_norm(x, gm, bt, 1e-5, rows=66, out_dtype=f32, bf16_copy=True)
    y0 = Fn.layer_norm(x, gm, bt, 1e-5, rows=66, out_dtype=f32)
    assert torch.equal(y, y0) and hasattr(y, "_bf16") and not hasattr(y0, "_bf16")
    assert torch.equal(y._bf16, y0.to(bf16)) and not y._bf16.requires_grad
    full, tgt, cls = Fn.enc_fanout(y)
    assert full._bf16 is y._bf16
    (full.sum() + 2 * tgt.sum() + 3 * cls.sum()).backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()


@pytest.mark.parametrize("Bn,T,Kd,N", [(3, 513, 512, 512), (2, 257, 256, 768)])
def test_retention_head_squared_error_epilogue_equals_mse_kernel(Bn, T, Kd, N):
    """HeadSqErrFn: the prediction is LinearRowsFn's bit for bit, the accumulator is mh_mse_masked_fwd's up to f32 summation order,
    and both loss Functions pick it up only for the very (target, mask) it was computed against."""
    from mirror_amd import functional as Fn
    prec = Fn.POLICIES["bf16"]
    g = torch.Generator().manual_seed(6)
    x = torch.randn(Bn, T, Kd, generator=g).cuda().to(bf16).requires_grad_(True)
    E = torch.randn(Bn, T, N, generator=g).cuda()
    tgt = E[:, 1:]
    mask = (torch.rand(Bn, T - 1, generator=g) < 0.75).float().cuda()
    w, b = _w(N, Kd, 7)
    y = Fn.head_sqerr(x, w, b, 1, T - 1, prec, tgt, mask)
    assert hasattr(y, "_sq"), "the fused path was not taken"
    y0 = Fn.LinearRowsFn.apply(x, w, b, 1, T - 1, prec, bf16)
    assert torch.equal(y, y0)
    acc0 = torch.zeros(2, device="cuda")
    Fn.K.mse_masked_fwd(y0.detach(), tgt, mask, acc0, Bn * (T - 1), N)
    acc = y._sq[0]
    torch.cuda.synchronize()
    assert abs(float(acc[1]) - float(acc0[1])) <= 1e-3 and float(acc0[1]) == float(mask.sum())
    assert abs(float(acc[0]) - float(acc0[0])) <= 2e-5 * float(acc0[0]), (float(acc[0]), float(acc0[0]))
    assert Fn._sq_of(y, tgt, mask) is acc
    assert Fn._sq_of(y, tgt.clone(), mask) is None and Fn._sq_of(y, tgt, mask.clone()) is None and Fn._sq_of(y0, tgt, mask) is None
    loss = Fn.masked_mse(y, tgt, mask, N)
    loss0 = Fn.masked_mse(y0, tgt, mask, N)
    assert abs(float(loss) - float(loss0)) <= 2e-5 * float(loss0)
    gx = torch.autograd.grad(loss, x, retain_graph=True)[0]
    gx0 = torch.autograd.grad(loss0, x)[0]
    assert float((gx.float() - gx0.float()).abs().max()) <= 1e-2 * float(gx0.float().abs().max())


def test_whole_model_train_step_fused_epilogues_equal_composed():
    """MIRROR forward + MIRRORLoss + backward in TRAIN mode (dropout on, bf16 policy) with the fused projection epilogues on and
    off (mirror_amd.kernels._EPI_ON): same dropout masks, same losses up to f32 summation order, same gradients up to the
    split-K / atomics order.  D = 512, 1024 tokens: every fused epilogue (DROPADD x 3 layers, MASKPOS, SQERR) is on its path."""
    import mirror_amd.models as M
    from mirror_amd import functional as Fn
    from mirror_amd import kernels as K
    from mirror_amd.losses import MIRRORLoss
    cfg = dict(wsi_embed_dim=128, rna_embed_dim=96, embed_dim=512, wsi_num_tokens=1024, rna_encoder_depth=1, rna_num_heads=8,
               rna_mlp_ratio=4.0, style_mlp_hidden_dim=128, style_mlp_out_dim=64, style_latent_dim=32, num_prototypes=300)
    g = torch.Generator().manual_seed(8)
    wsi = torch.randn(2, 1024, 128, generator=g).cuda().to(bf16)
    rna = torch.randn(2, 96, generator=g).cuda()
    noise = {"wsi_mask": torch.rand(2, 1024, generator=g).cuda(), "rna_mask": torch.rand(2, 512, generator=g).cuda(),
             "wsi_eps": torch.randn(2, 32, generator=g).cuda(), "rna_eps": torch.randn(2, 32, generator=g).cuda()}
    out = []
    was = K._EPI_ON
    try:
        for on in (True, False):
            K._EPI_ON = on
            torch.manual_seed(0)
            model = M.mirror(**cfg).cuda().train()
            model.precision = "bf16"
            Fn.manual_seed(99)
            launched = []
            orig = K.linear_fused
            K.linear_fused = lambda *a, **k: (launched.append(a[4].kind), orig(*a, **k))[1]
            try:
                losses = MIRRORLoss()(*model(wsi, rna, noise=noise))
                losses[0].backward()
            finally:
                K.linear_fused = orig
            torch.cuda.synchronize()
            assert sorted(launched) == ([1, 1, 1, 2, 3] if on else []), launched
            out.append(([float(x) for x in losses], {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
    finally:
        K._EPI_ON = was
    (l1, g1), (l2, g2) = out
    for a, c in zip(l1, l2):
        assert abs(a - c) <= 1e-4 * max(abs(c), 1e-3), (l1, l2)
    worst = 0.0
    for k in g1:
        d = float((g1[k] - g2[k]).norm()) / max(float(g2[k].norm()), 1e-12)
        worst = max(worst, d)
        assert d <= 2e-2, (k, d)          # bf16 activations turn a different f32 summation order into 1-ulp flips downstream
    print("fused vs composed: losses", l1, l2, "worst relative gradient difference", worst)


@pytest.mark.parametrize("Bn,n,D,m", [(3, 1025, 512, 256), (2, 300, 256, 128), (2, 4200, 512, 256), (4, 2300, 512, 256)])
def test_norm_qkv_with_landmark_rows_and_its_backward(Bn, n, D, m):
    """Fn.NormQkvLmFn (LayerNorm + to_qkv with the landmark means as extra rows of the same products: mh_layernorm_fwd_lm,
    the flat row-window GEMM with mh_gemm_desc.window_batches, mh_layernorm_bwd_lm with a bf16 addend) against the composed
    LayerNorm -> to_qkv -> explicit group means of q | k through torch autograd.  (3, 1025, 512, 256) runs the flat
    row-window kernels (3 ragged rows through the weight-streaming kernel), (2, 300, 256, 128) the generic products; (2, 4200, ..)
    and (4, 2300, ..) have few long landmark groups (l = 17 / 9): four / two waves share a group in mh_layernorm_fwd_lm."""
    import math
    from mirror_amd import functional as Fn
    prec = Fn.POLICIES["bf16"]
    g = torch.Generator().manual_seed(9)
    pad = (m - n % m) % m
    l = math.ceil(n / m)
    assert (pad + n) == m * l
    x0 = torch.randn(Bn, n, D, generator=g).cuda()
    gm0, bt0 = (1 + 0.1 * torch.randn(D, generator=g)).cuda(), (0.1 * torch.randn(D, generator=g)).cuda()
    w0 = (torch.randn(3 * D, D, generator=g) * D ** -0.5).cuda()
    up_q = torch.randn(Bn, pad + n, 3 * D, generator=g).cuda().to(bf16)
    up_l = torch.randn(Bn, m, 2 * D, generator=g).cuda().to(bf16)
    res = []
    for fused in (True, False):
        x, gm, bt, w = (t.clone().requires_grad_(True) for t in (x0, gm0, bt0, w0))
        if fused:
            qkv, lm = Fn.NormQkvLmFn.apply(x, gm, bt, 1e-5, n, pad, l, w, prec)
            assert lm.stride(1) == 3 * D and lm.data_ptr() == qkv.data_ptr() + qkv.numel() * 2      # rows of ONE buffer
            Fn.run_deferred(qkv)
            # the gradient arrives as NystromCoreFn hands it over: the two row ranges of one [B n_p + B m, 3D] buffer
            de, dq, dl = Fn.ext_rows_alloc(Bn, pad + n, m, 3 * D, 2 * D, x.device)
            dq.copy_(up_q)
            dl.copy_(up_l)
            de[Bn * (pad + n):, 2 * D:].zero_()
            torch.autograd.backward([qkv, lm], [dq, dl])
        else:
            y = Fn.layer_norm(x, gm, bt, 1e-5, pad=pad, out_dtype=bf16)
            qkv = Fn.linear(y, w, None, prec=prec)
            lm = qkv[..., :2 * D].float().reshape(Bn, m, l, 2 * D).mean(2)
            ((qkv.float() * up_q.float()).sum() + (lm * up_l.float()).sum()).backward()
        res.append((qkv.detach().float().clone(), lm.detach().float().clone(), x.grad.clone(), gm.grad.clone(), bt.grad.clone(), w.grad.clone()))
    torch.cuda.synchronize()
    (q1, m1, dx1, dg1, db1, dw1), (q2, m2, dx2, dg2, db2, dw2) = res
    assert torch.equal(q1, q2) and float(q1[:, :pad].abs().max() if pad else 0.0) == 0.0        # the same products on the same rows
    assert float((m1 - m2).abs().max()) <= 2 ** -6 * float(m2.abs().max())          # mean rounded to bf16 before the projection, not after
    for a, c, tol in ((dx1, dx2, 6e-3), (dg1, dg2, 6e-3), (db1, db2, 6e-3), (dw1, dw2, 6e-3)):
        assert float((a - c).norm()) <= tol * float(c.norm()), (float((a - c).norm()), float(c.norm()))


def test_whole_model_landmarks_from_layernorm_means_equal_the_landmark_kernels():
    """bf16 policy, train mode, D = 512: landmarks as to_qkv(group means of the LayerNorm output) (Fn.NormQkvLmFn) against
    the landmark kernels on q | k.  Algebraically identical ([3P] to_qkv is linear and bias-free); numerically the mean is
    rounded to bf16 before the projection instead of after it: losses within 2e-3, parameter gradients cosine >= 0.99 (the
    small bias vectors of the heads are the noisiest: 0.996 measured)."""
    import mirror_amd.models as M
    from mirror_amd import functional as Fn
    from mirror_amd.losses import MIRRORLoss
    cfg = dict(wsi_embed_dim=128, rna_embed_dim=96, embed_dim=512, wsi_num_tokens=1024, rna_encoder_depth=1, rna_num_heads=8,
               rna_mlp_ratio=4.0, style_mlp_hidden_dim=128, style_mlp_out_dim=64, style_latent_dim=32, num_prototypes=300)
    g = torch.Generator().manual_seed(10)
    wsi = torch.randn(2, 1024, 128, generator=g).cuda().to(bf16)
    rna = torch.randn(2, 96, generator=g).cuda()
    noise = {"wsi_mask": torch.rand(2, 1024, generator=g).cuda(), "rna_mask": torch.rand(2, 512, generator=g).cuda(),
             "wsi_eps": torch.randn(2, 32, generator=g).cuda(), "rna_eps": torch.randn(2, 32, generator=g).cuda()}
    out = []
    was = Fn._LM_ROWS
    try:
        for on in (True, False):
            Fn._LM_ROWS = on
            torch.manual_seed(0)
            model = M.mirror(**cfg).cuda().train()
            model.precision = "bf16"
            Fn.manual_seed(99)
            losses = MIRRORLoss()(*model(wsi, rna, noise=noise))
            losses[0].backward()
            torch.cuda.synchronize()
            out.append(([float(x) for x in losses], {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
    finally:
        Fn._LM_ROWS = was
    (l1, g1), (l2, g2) = out
    for a, c in zip(l1, l2):
        assert abs(a - c) <= 2e-3 * max(abs(c), 1e-3), (l1, l2)
    worst = 1.0
    for k in g1:
        a, c = g1[k].flatten().double(), g2[k].flatten().double()
        if float(c.norm()) < 1e-10:
            continue
        cos = float(a @ c / (a.norm() * c.norm()))
        worst = min(worst, cos)
        assert cos >= 0.99, (k, cos)
    print("landmark algebra vs kernels: losses", l1, l2, "worst gradient cosine", worst)


def test_lite_dropout_stream_statistics_and_backward_mask():
    """mh_dropout_lite: keep rate 1 - round(p * 65536) / 65536, unbiased scaling, the same mask for the same (seed, offset) — the
    backward regenerates it — and a different one for a different offset or seed."""
    from mirror_amd import kernels as K
    x = torch.ones(1 << 22, device="cuda")
    y = K.dropout_lite(x, 0.1, 1234, 64)
    y2 = K.dropout_lite(x.to(bf16), 0.1, 1234, 64)
    torch.cuda.synchronize()
    keep = float((y != 0).float().mean())
    assert abs(keep - (1 - 6554 / 65536)) < 1e-3, keep
    assert abs(float(y.mean()) - 1.0) < 2e-3
    assert torch.equal(y != 0, y2 != 0)
    assert not torch.equal(y != 0, K.dropout_lite(x, 0.1, 1234, 72) != 0)
    assert not torch.equal(y != 0, K.dropout_lite(x, 0.1, 1235, 64) != 0)
    a = torch.randn(1 << 22, device="cuda")
    z = K.dropout_lite(x, 0.1, 1234, 64, add_to=a)
    assert torch.equal(z, a + y)
    # neighbouring elements are independent: P(both kept) = keep^2
    both = float(((y[:-1] != 0) & (y[1:] != 0)).float().mean())
    assert abs(both - keep * keep) < 2e-3, (both, keep)


def test_skinny_wgrad_many_equals_single_launches():
    """mh_skinny_wgrad_many (one grid over the tiles of many [B, D]-row weight gradients) against mh_skinny_wgrad per item,
    bit for bit (same tile code, same accumulation order), ragged N / K included."""
    from mirror_amd import kernels as K
    g = torch.Generator().manual_seed(12)
    shapes = [(16, 512, 512), (16, 3000, 512), (7, 100, 72), (32, 256, 128), (16, 512, 2048), (1, 64, 64)]
    items, ref = [], []
    for M, N, Kd in shapes:
        dy = torch.randn(M, N, generator=g).cuda().to(bf16)
        x = torch.randn(M, Kd, generator=g).cuda().to(bf16)
        dw0 = torch.randn(N, Kd, generator=g).cuda()
        db0 = torch.randn(N, generator=g).cuda()
        dw1, db1 = dw0.clone(), db0.clone()
        K.skinny_wgrad(dy, x, dw1, accumulate=True, db=db1)
        dw2, db2 = dw0.clone(), db0.clone()
        items.append((dy, x, dw2, db2 if N != 100 else None))
        ref.append((dw1, db1 if N != 100 else None))
    K.skinny_wgrad_many(items)                # one launch (items must not share a destination: they run concurrently)
    K.skinny_wgrad_many(items)                # and once more: accumulation
    for (dy, x, dw2, db2), (dw1, db1) in zip(items, ref):
        K.skinny_wgrad(dy, x, dw1, accumulate=True, db=db1)
        torch.cuda.synchronize()
        assert torch.equal(dw2, dw1)
        if db2 is not None:
            assert torch.equal(db2, db1)
    many = []
    for i in range(40):                       # more than one launch's worth of items
        dy = torch.randn(16, 64, generator=g).cuda().to(bf16)
        x = torch.randn(16, 96, generator=g).cuda().to(bf16)
        many.append((dy, x, torch.zeros(64, 96, device="cuda"), None))
    K.skinny_wgrad_many(many)
    torch.cuda.synchronize()
    for dy, x, dw, _ in many:
        assert float((dw - dy.float().t() @ x.float()).abs().max()) < 1e-3


def test_nys_sim2_one_launch_equals_composed_chain_inputs():
    """mh_nys_sim2 (sim2 + softmax + tensor-wide abs-sum maxima + chain operand packing, one launch) against the composed
    mh_gemm / mh_softmax_fwd / mh_pinv_absmax / mh_pinv_chain_prep sequence, and the chain forward fed either way."""
    from mirror_amd import kernels as K
    from mirror_amd._lib import MH_BF16
    g = torch.Generator().manual_seed(13)
    Bn, h, m, dh = 3, 8, 256, 64
    D = h * dh
    lm = (torch.randn(Bn, m, 2 * D, generator=g) * 0.7).cuda().to(bf16)
    scale = dh ** -0.5
    a2, xt, z0f, st = K.nys_sim2(lm, h, scale, want_z0f=True)
    ql = lm.view(Bn, m, 2, h, dh)[:, :, 0].permute(0, 2, 1, 3)
    kl = lm.view(Bn, m, 2, h, dh)[:, :, 1].permute(0, 2, 1, 3)
    ref = K.gemm(ql, kl.transpose(-1, -2), alpha=scale, mma=MH_BF16, out_dtype=f32)
    K.softmax_fwd(ref, ref)
    st0 = K.pinv_absmax(ref)
    saved0 = K.pinv_chain_saved_alloc(6, Bn * h, m, "cuda")
    z0_ref, xt_ref = K.pinv_chain_prep(ref, st0, K.pinv_chain_z0_slot(saved0))
    torch.cuda.synchronize()
    assert float((a2 - ref).abs().max()) <= 2e-6, float((a2 - ref).abs().max())
    assert float((a2.sum(-1) - 1).abs().max()) < 1e-5
    v = lambda s: s.view(torch.int32).view(-1)[1::2].view(torch.float32)          # the float halves of the packed maxima  # noqa: E731
    assert torch.allclose(v(st)[:2], v(st0)[:2], rtol=1e-5), (v(st), v(st0))
    assert float((xt.float() - xt_ref.float()).abs().max()) <= 2 ** -8 * float(ref.max())       # bf16 ulp flips of ~1e-7 differences
    inv = 1.0 / float(v(st0)[0] * v(st0)[1])
    # z0f is panel native; scale it and compare through the chain's own saved[0] slot
    saved1 = K.pinv_chain_saved_alloc(6, Bn * h, m, "cuda")
    zf0 = torch.empty(Bn, h, m, m, device="cuda", dtype=bf16)
    zf1 = torch.empty_like(zf0)
    K.pinv_chain_fwd(xt_ref, saved0, zf0, 6)
    K.pinv_chain_fwd(xt, saved1, zf1, 6, z0f=z0f, stats=st)
    torch.cuda.synchronize()
    assert float((saved1[0, 0].float() - saved0[0, 0].float()).abs().max()) <= 2 ** -7 * float(saved0[0, 0].float().abs().max())
    assert abs(float(z0f.sum()) * inv - float(z0_ref.sum())) <= 1e-3 * abs(float(z0_ref.sum()))
    d = float((zf1.float() - zf0.float()).norm()) / float(zf0.float().norm())
    assert d < 2e-2, d
    # default path: no second pass in nys_sim2, the chain forward forms z_0 from the ROWS of attn2 — the same bf16 z_0, bit for bit
    a2b, xtb, none, stb = K.nys_sim2(lm, h, scale, torch.zeros(2, device="cuda", dtype=torch.int64))
    assert none is None and torch.equal(a2b, a2) and torch.equal(xtb, xt)
    saved2 = K.pinv_chain_saved_alloc(6, Bn * h, m, "cuda")
    zf2 = torch.empty_like(zf0)
    K.pinv_chain_fwd(xtb, saved2, zf2, 6, z0f=a2b, stats=stb, z0_rowmajor=True)
    torch.cuda.synchronize()
    # (the second pass recomputes the logits with the operands in the other MFMA roles and a differently contracted exp argument: an
    # f32 ulp now and then, i.e. a rare bf16 flip; the rows of attn2 ARE its transpose)
    assert float((saved2[0, 0].float() - saved1[0, 0].float()).abs().max()) <= 2 ** -7 * float(saved1[0, 0].float().abs().max())
    inv_t = 1.0 / float(v(stb)[0] * v(stb)[1])
    z0_exact = (a2b.transpose(-1, -2) * inv_t).to(bf16)
    it = torch.arange(m * m // 8, device="cuda")
    lane, T, jblk = it & 63, (it >> 6) & 15, it >> 10
    e = torch.arange(8, device="cuda")
    I = (16 * T + 4 * (lane >> 5))[:, None] + (e & 3) + 8 * (e >> 2)
    J = (32 * jblk + (lane & 31))[:, None].expand(-1, 8)
    pn = z0_exact.reshape(Bn * h, m, m)[:, I, J].reshape(Bn * h, m, m)
    assert float((saved2[0, 0].reshape(Bn * h, m, m).float() - pn.float()).abs().max()) <= 2 ** -8 * float(pn.float().abs().max())
    assert float((zf2.float() - zf1.float()).norm()) <= 1e-2 * float(zf1.float().norm())
    # the backward's z0 adjoint without a stored z0
    dz0 = torch.randn(Bn, h, m, m, generator=g).cuda()
    dx_a, dx_b = torch.zeros_like(ref), torch.zeros_like(ref)
    K.pinv_z0_bwd(ref, z0_ref, dz0, st0, dx_a)
    K.pinv_z0_bwd(ref, None, dz0, st0, dx_b)
    torch.cuda.synchronize()
    assert float((dx_a - dx_b).norm()) <= 1e-5 * float(dx_a.norm())


def test_whole_model_nys_sim2_on_off(monkeypatch):
    """Whole model, bf16, train mode: the one-launch sim2 path against the composed one (losses within 1e-3, gradient cosine >= 0.99)."""
    import mirror_amd.models as M
    from mirror_amd import functional as Fn
    from mirror_amd.losses import MIRRORLoss
    cfg = dict(wsi_embed_dim=128, rna_embed_dim=96, embed_dim=512, wsi_num_tokens=1024, rna_encoder_depth=1, rna_num_heads=8,
               rna_mlp_ratio=4.0, style_mlp_hidden_dim=128, style_mlp_out_dim=64, style_latent_dim=32, num_prototypes=300)
    g = torch.Generator().manual_seed(14)
    wsi = torch.randn(2, 1024, 128, generator=g).cuda().to(bf16)
    rna = torch.randn(2, 96, generator=g).cuda()
    noise = {"wsi_mask": torch.rand(2, 1024, generator=g).cuda(), "rna_mask": torch.rand(2, 512, generator=g).cuda(),
             "wsi_eps": torch.randn(2, 32, generator=g).cuda(), "rna_eps": torch.randn(2, 32, generator=g).cuda()}
    out = []
    from mirror_amd import kernels as K
    for on in (True, False):
        monkeypatch.setattr(K, "_NYS_SIM2", on)
        torch.manual_seed(0)
        model = M.mirror(**cfg).cuda().train()
        model.precision = "bf16"
        Fn.manual_seed(99)
        losses = MIRRORLoss()(*model(wsi, rna, noise=noise))
        losses[0].backward()
        torch.cuda.synchronize()
        out.append(([float(x) for x in losses], {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
    (l1, g1), (l2, g2) = out
    for a, c in zip(l1, l2):
        assert abs(a - c) <= 1e-3 * max(abs(c), 1e-3), (l1, l2)
    for k in g1:
        a, c = g1[k].flatten().double(), g2[k].flatten().double()
        if float(c.norm()) < 1e-10:
            continue
        assert float(a @ c / (a.norm() * c.norm())) >= 0.99, k


def test_whole_model_round5_fusions_on_off(monkeypatch):
    """Whole model, bf16, train mode, D = 512 / 1024 tokens (no square-pad rows: the ReLU fusion applies).  Round 5's fused forms against
    the launches they replace, one switch at a time: (a) _fc1's ReLU backward inside layer 1's LayerNorm backward — the same f32 value
    rounded once either way: _fc1's weight / bias gradients and the cls-token gradient at the run-to-run floor (f32 atomics above them);
    (b) res_conv inside attn3's forward + its two gradients in one pass — same MFMA products (losses identical, gradient cosine >= 0.9999);
    (c) attn3's backward as one kernel and (d) attn1's dq from the saved rows beside the chain / to_out's weight gradient in the
    window — other summation orders of the same bf16 products (losses identical: the forward does not change; cosine >= 0.999);
    (e) the three bias gradients left by the pass that wrote their dy (masked-MSE backward -> retention_head, mask/pos backward ->
    retention_embed, layer 1's LayerNorm backward -> _fc1) against mh_colsum over dy: the same stored values, another order;
    (f) the fan-out sum of the encoder output's three gradients inside its LayerNorm's backward against mh_fanout_bwd in front of it;
    (g) to_out's Dropout backward + bias gradient inside the LayerNorm backward that produces its dy (the final norms of the encoder and
    of the retention decoder) against mh_dropout_lite_colsum: bit-equal masked gradients, the bias gradients another summation order."""
    import mirror_amd.models as M
    from mirror_amd import functional as Fn
    from mirror_amd import kernels as K
    from mirror_amd.losses import MIRRORLoss
    cfg = dict(wsi_embed_dim=256, rna_embed_dim=96, embed_dim=512, wsi_num_tokens=1024, rna_encoder_depth=1, rna_num_heads=8,
               rna_mlp_ratio=4.0, style_mlp_hidden_dim=128, style_mlp_out_dim=64, style_latent_dim=32, num_prototypes=300)
    g = torch.Generator().manual_seed(15)
    wsi = torch.randn(2, 1024, 256, generator=g).cuda().to(bf16)
    rna = torch.randn(2, 96, generator=g).cuda()
    noise = {"wsi_mask": torch.rand(2, 1024, generator=g).cuda(), "rna_mask": torch.rand(2, 512, generator=g).cuda(),
             "wsi_eps": torch.randn(2, 32, generator=g).cuda(), "rna_eps": torch.randn(2, 32, generator=g).cuda()}

    def run():
        torch.manual_seed(0)
        model = M.mirror(**cfg).cuda().train()
        model.precision = "bf16"
        Fn.manual_seed(99)
        losses = MIRRORLoss()(*model(wsi, rna, noise=noise))
        losses[0].backward()
        torch.cuda.synchronize()
        return [float(x) for x in losses], {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    base_l, base_g = run()
    relu_keys = ("wsi_encoder._fc1.0.weight", "wsi_encoder._fc1.0.bias", "wsi_encoder.cls_token")
    bias_keys = tuple(k for k in base_g if k in ("wsi_encoder._fc1.0.bias", "wsi_encoder.retention_embed.bias",
                                                 "wsi_encoder.retention_head.bias"))
    assert len(bias_keys) == 3, [k for k in base_g if k.endswith("bias")][:40]
    seen_bias = set()
    for target, name, floor, exact_losses in ((Fn, "_RELU_IN_LN_BWD", 1.0, True), (Fn, "_RC_FUSED", 0.9999, True),
                                              (K, "NYS_A3_BWD_ONE_PASS", 0.999, True), (Fn, "_A1_DQ_IN_WINDOW", 0.999, True),
                                              (Fn, "_TO_OUT_WGRAD_IN_WINDOW", 0.999, True), (Fn, "_BIAS_IN_PRODUCER", 0.9999, True),
                                              (Fn, "_FAN_IN_LN_BWD", 0.9999, True), (Fn, "_DROP_IN_LN_BWD", 0.9999, True)):
        orig = getattr(target, name)          # (the hooks' defaults follow the measurements: most are on, some are off)
        monkeypatch.setattr(target, name, not orig)
        l2, g2 = run()
        monkeypatch.setattr(target, name, orig)
        if exact_losses:          # the forward is the same arithmetic either way: equal up to the f32-atomics order of the loss sums
            assert all(abs(x - y) <= 1e-5 * max(abs(y), 1e-3) for x, y in zip(l2, base_l)), (name, l2, base_l)
        for k in base_g:
            a, c = base_g[k].flatten().double(), g2[k].flatten().double()
            if float(c.norm()) < 1e-10:
                continue
            cos = float(a @ c / (a.norm() * c.norm()))
            if name == "_BIAS_IN_PRODUCER" and k in bias_keys:
                seen_bias.add(k)
                assert cos >= 0.99995 and abs(float(a.norm() / c.norm()) - 1.0) <= 1e-3, (name, k, cos)
            if name == "_RELU_IN_LN_BWD" and k in relu_keys:
                # the same f32 value rounded once to bf16 either way: only the run-to-run noise of the gradients above it is left
                assert cos >= 0.99995 and abs(float(a.norm() / c.norm()) - 1.0) <= 1e-3, (name, k, cos)
            assert cos >= min(floor, 0.9999), (name, k, cos)
    assert seen_bias == set(bias_keys)


def test_masked_landmark_rows_node_equals_the_composed_masked_path(monkeypatch):
    """BASELINE config 4 (key-padding mask), bf16, train mode: LayerNorm + to_qkv with the landmarks as extra rows UNDER A MASK
    (mh_layernorm_fwd_lm / _bwd_lm row_mask: masked rows leave as zero rows and stay out of the landmark sums) against the composed
    path it replaces (LayerNorm, mh_row_scale, to_qkv, mh_landmark_fwd): the same linear algebra in another rounding order — losses
    within bf16 noise, every gradient within cosine 0.999 (landmark means of the norm's output vs of its projection)."""
    import mirror_amd.models as M
    from mirror_amd import functional as Fn
    import importlib
    MM = importlib.import_module("mirror_amd.models.mirror")
    from mirror_amd.losses import MIRRORLoss
    cfg = dict(wsi_embed_dim=128, rna_embed_dim=96, embed_dim=512, wsi_num_tokens=1000, rna_encoder_depth=1, rna_num_heads=8,
               rna_mlp_ratio=4.0, style_mlp_hidden_dim=128, style_mlp_out_dim=64, style_latent_dim=32, num_prototypes=300)
    g = torch.Generator().manual_seed(16)
    n = 1000
    wsi = torch.randn(3, n, 128, generator=g).cuda().to(bf16)
    rna = torch.randn(3, 96, generator=g).cuda()
    lens = torch.tensor([n, 611, 257]).cuda()
    mask = torch.arange(n).cuda()[None, :] < lens[:, None]
    noise = {"wsi_mask": torch.rand(3, n, generator=g).cuda(), "rna_mask": torch.rand(3, 512, generator=g).cuda(),
             "wsi_eps": torch.randn(3, 32, generator=g).cuda(), "rna_eps": torch.randn(3, 32, generator=g).cuda()}

    def run():
        torch.manual_seed(0)
        model = M.mirror(**cfg).cuda().train()
        model.precision = "bf16"
        Fn.manual_seed(99)
        losses = MIRRORLoss()(*model(wsi, rna, noise=noise, wsi_key_padding_mask=mask))
        losses[0].backward()
        torch.cuda.synchronize()
        return [float(x) for x in losses], {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    l1, g1 = run()
    monkeypatch.setattr(MM, "_LM_MASKED", False)
    monkeypatch.setattr(Fn, "_SIM2_MASKED", False)      # and the one-launch masked sim2 (mlm of mh_nys_sim2) against gemm + mh_softmax_masked_fwd + ...
    l0, g0 = run()
    assert all(abs(x - y) <= 2e-3 * max(abs(y), 1e-2) for x, y in zip(l1, l0)), (l1, l0)
    for k in g0:
        a, c = g1[k].flatten().double(), g0[k].flatten().double()
        if float(c.norm()) < 1e-10:
            continue
        assert float(a @ c / (a.norm() * c.norm())) >= 0.999, k
    # 1000 tokens are no square (24 square-pad rows): _fc1's ReLU backward inside layer 1's LayerNorm backward with the pad rows' gradients
    # folded onto the rows they copy (bf16 after the gate) against the f32 fold + mh_relu_bwd of the composed path
    monkeypatch.setattr(MM, "_LM_MASKED", True)
    monkeypatch.setattr(Fn, "_SIM2_MASKED", True)
    monkeypatch.setattr(Fn, "_RELU_SQUARE_PAD", False)
    l2, g2 = run()
    assert all(abs(x - y) <= 1e-5 * max(abs(y), 1e-3) for x, y in zip(l2, l1)), (l2, l1)
    for k in g1:
        a, c = g1[k].flatten().double(), g2[k].flatten().double()
        if float(c.norm()) < 1e-10:
            continue
        assert float(a @ c / (a.norm() * c.norm())) >= 0.9999, k


def test_attn2_backward_tail_one_pass_equals_z0_bwd_plus_softmax_bwd():
    """mh_pinv_s2_bwd (z_0 backward + the two max() sub-gradients + attn2's softmax backward in one pass, the column maximum as a
    rank-one correction) against the composed mh_pinv_z0_bwd + mh_softmax_bwd on the same inputs ([3P] moore_penrose_iter_pinv's
    start, called at models/mirror.py:312)."""
    from mirror_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    BH, m = 6, 256
    p = torch.softmax(torch.randn(BH, m, m, generator=g) * 2.0, dim=-1).cuda().contiguous()
    dz0 = torch.randn(BH, m, m, generator=g).cuda()
    dx0 = torch.randn(BH, m, m, generator=g).cuda()
    st = K.pinv_absmax(p, torch.zeros(4, device="cuda").view(torch.int64))
    ref = dx0.clone()
    K.pinv_z0_bwd(p, None, dz0, st, ref)
    K.softmax_bwd(p, ref)
    got = dx0.clone()
    K.pinv_s2_bwd(p, dz0, st, got)
    got_z = dx0.clone()
    K.pinv_s2_bwd(p, dz0, st, got_z, zeroed_scratch=torch.zeros(1, device="cuda"))     # the caller's pre-zeroed float: no memset node
    ref_z = dx0.clone()
    K.pinv_z0_bwd(p, None, dz0, st, ref_z, zeroed_scratch=torch.zeros(1, device="cuda"))
    K.softmax_bwd(p, ref_z)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    # (the dot product behind the column term is a float-atomic sum: run-to-run rounding, not bit equality)
    assert float((got_z - got).abs().max()) <= 2e-6 * scale and float((ref_z - ref).abs().max()) <= 2e-6 * scale
    assert float((got - ref).abs().max()) <= 2e-6 * scale, (float((got - ref).abs().max()), scale)
    # the matrix that holds the column maximum really is corrected (the test would pass trivially if the term were negligible)
    ci = int(st.view(torch.int64)[1].item()) & 0xffffffff
    raw = dx0 + dz0.transpose(-1, -2) / (float(p.sum(-1).max()) * float(p.sum(-2).max()))
    nofix = p * (raw - (p * raw).sum(-1, keepdim=True))
    assert float((nofix[ci // m] - ref[ci // m]).abs().max()) > 1e-4 * scale


def test_attn2_backward_tail_under_a_key_padding_mask_equals_the_composed_masked_path():
    """mh_pinv_s2_bwd(mlm) (round 5, BASELINE config 4) against mh_pinv_z0_bwd + mh_softmax_masked_bwd: p comes from masked_fill + softmax
    (mh_nys_sim2(mlm), checked here against mh_gemm + mh_softmax_masked_fwd too): filled entries get no gradient — also in a fully masked,
    uniform row — and the column maximum's rank-one term stays out of them."""
    from mirror_amd import kernels as K
    from mirror_amd._lib import MH_BF16
    g = torch.Generator().manual_seed(6)
    B, h, m, dh = 3, 2, 256, 64
    lm = (torch.randn(B, m, 2 * h * dh, generator=g) * 0.5).cuda().to(bf16)
    mlm = torch.ones(B, m)
    mlm[1, 200:] = 0
    mlm[2, :17] = 0
    mlm[2, 100] = 0
    mlm = mlm.cuda().contiguous()
    scale = dh ** -0.5
    st = torch.zeros(4, device="cuda").view(torch.int64)
    a2, xt, z0f, st = K.nys_sim2(lm, h, scale, st, mlm=mlm)
    ql = lm[..., :h * dh].reshape(B, m, h, dh).permute(0, 2, 1, 3)
    kl = lm[..., h * dh:].reshape(B, m, h, dh).permute(0, 2, 1, 3)
    ref_p = K.gemm(ql, kl.transpose(-1, -2), alpha=scale, mma=MH_BF16, out_dtype=torch.float32)
    K.softmax_masked_fwd(ref_p, mlm, mlm, ref_p)
    assert float((a2 - ref_p).abs().max()) <= 2e-6
    assert abs(float(a2[1, 0, 250].sum()) - 1.0) < 1e-5 and float((a2[1, 0, 250] - 1.0 / m).abs().max()) < 1e-7      # a fully masked row: uniform
    assert float(a2[1, 0, 3, 200:].abs().max()) == 0.0                                                           # masked columns of a valid row
    st_ref = K.pinv_absmax(ref_p, torch.zeros(4, device="cuda").view(torch.int64))
    # the same maxima: values of both, position of the column maximum (every row's abs sum is 1 up to rounding: its argmax is a tie-break)
    sv, sr = st.view(torch.int64)[:2], st_ref.view(torch.int64)[:2]
    assert int(sv[1]) == int(sr[1]) and abs((int(sv[0]) >> 32) - (int(sr[0]) >> 32)) <= 8
    p = a2.reshape(B * h, m, m).contiguous()
    dz0 = torch.randn(B * h, m, m, generator=g).cuda()
    dx0 = torch.randn(B * h, m, m, generator=g).cuda()
    ref = dx0.clone()
    K.pinv_z0_bwd(p, None, dz0, st, ref)
    K.softmax_masked_bwd(p.view(B, h, m, m), ref.view(B, h, m, m), mlm, mlm)
    got = dx0.clone()
    K.pinv_s2_bwd(p, dz0, st, got, mlm=mlm, heads=h)
    torch.cuda.synchronize()
    sc = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 3e-6 * sc, (float((got - ref).abs().max()), sc)
    assert float(got.view(B, h, m, m)[1, :, 200:].abs().max()) == 0.0 and float(got.view(B, h, m, m)[1, :, :, 200:].abs().max()) == 0.0


def test_dz_dav_one_launch_equals_gemm_pack_gemm():
    """mh_nys_dz_dav (dZ = dW2 av^T packed panel native for the chain + dAV = Z^T dW2) against mh_gemm + mh_pinv_chain_pack + mh_gemm:
    the products that open [3P] NystromAttention's backward around moore_penrose_iter_pinv (models/mirror.py:312)."""
    from mirror_amd import kernels as K
    from mirror_amd._lib import MH_BF16
    g = torch.Generator().manual_seed(9)
    B, h, m, dh = 2, 4, 256, 64
    dw2 = torch.randn(B, h, m, dh, generator=g).cuda()
    av = torch.randn(B, h, m, dh, generator=g).cuda()
    zfT = (torch.randn(B, h, m, m, generator=g) * 0.1).cuda().to(bf16)
    dz = K.gemm(dw2, av.transpose(-1, -2), mma=MH_BF16, out_dtype=torch.float32)
    up_ref = K.pinv_chain_pack(dz)
    dav_ref = K.gemm(zfT, dw2, mma=MH_BF16, out_dtype=bf16)             # Z^T dW2 with Z^T = zfT
    up, dav = K.nys_dz_dav(dw2, av, zfT)
    torch.cuda.synchronize()
    assert up.shape == up_ref.shape and dav.shape == dav_ref.shape
    # same bf16-rounded operands, f32 accumulation: results agree to the rounding of the bf16 outputs (summation order may differ)
    du = (up.float() - up_ref.float()).abs().max() / up_ref.float().abs().max()
    dd = (dav.float() - dav_ref.float()).abs().max() / dav_ref.float().abs().max()
    assert float(du) <= 8e-3 and float(dd) <= 8e-3, (float(du), float(dd))
    assert float((up.float() - up_ref.float()).abs().mean() / up_ref.float().abs().mean()) <= 5e-4
    # the third result: attn3's delta = sum_d dAV av of the dAV this launch stored, exactly what mh_nys_attn3_bwd's first launch computes
    up3, dav3, delta3 = K.nys_dz_dav(dw2, av, zfT, want_delta3=True)
    torch.cuda.synchronize()
    assert torch.equal(up3, up) and torch.equal(dav3, dav) and delta3.shape == (B, h, m)
    dref = (dav.float() * av).sum(-1)
    assert float((delta3 - dref).abs().max()) <= 1e-4 * float(dref.abs().max())


@pytest.mark.parametrize("B,T,R,Kd,N,kc", [(2, 1280, 1025, 512, 1024, 1), (2, 1280, 1025, 1536, 512, 0), (3, 768, 513, 256, 256, 1)])
@pytest.mark.parametrize("beside_chain", [False, True])
def test_rows_window_product_skips_the_front_pad_rows(B, T, R, Kd, N, kc, beside_chain):
    """to_qkv / its data gradient over the B x n real rows of [3P] NystromAttention's front-padded buffers as one flat problem
    (mh_gemm_desc.a_rows_per_batch + c_rows_per_batch, the ragged rows through the weight-streaming kernel): equals the matmul of the
    window, rows outside the window and columns outside the slice are untouched (models/mirror.py:312)."""
    from mirror_amd import functional as Fn, kernels as K
    from mirror_amd._lib import MH_BF16
    g = torch.Generator().manual_seed(1)
    r0 = T - R
    a = torch.randn(B, T, Kd, generator=g).cuda().to(bf16)
    if kc:
        w = (torch.randn(N, Kd, generator=g) * 0.05).cuda().to(bf16)
        b2, wt = w.t(), None
    else:
        w = (torch.randn(Kd, N, generator=g) * 0.05).cuda().to(bf16)
        b2, wt = w, w.t().contiguous()
    out = torch.full((B, T, N + 256), 7.0, device="cuda", dtype=bf16)
    o3 = out[..., :N]
    K.shared_chip = beside_chain            # the one-workgroup-per-tile kernel instead of the persistent one
    try:
        Fn._rows_window(a, b2, o3, r0, R, mma=MH_BF16, wt=wt)
    finally:
        K.shared_chip = False
    torch.cuda.synchronize()
    ref = (a[:, r0:].float() @ b2.float())
    assert float((o3[:, r0:].float() - ref).abs().max()) <= 1e-2 * float(ref.abs().max())
    assert bool((o3[:, :r0] == 7).all()) and bool((out[..., N:] == 7).all())


def test_template_geometry_pinv_on_the_side_stream_equals_the_serial_order():
    """D = 768 (dh = 96, m = 384: the reference template, configs/pretrain/mirror.template.yaml:27-31): the tile-kernel Moore-Penrose
    iteration beside the attention sides' products on the side stream (Fn._TILE_SIDE) and the f32 partial sums of its backward as
    addends (Fn._PINV_R32) against the serial order / read-modify-write sums — the same launches in another order: output and gradients
    agree to the run-to-run noise of mh_pinv_z0_bwd's atomics."""
    import importlib
    from mirror_amd import functional as Fn
    MM = importlib.import_module("mirror_amd.models.mirror")
    prec = Fn.POLICIES["bf16"]
    g = torch.Generator().manual_seed(21)
    x0 = torch.randn(2, 900, 768, generator=g).cuda()
    up = torch.randn(2, 900, 768, generator=g).cuda()

    def run():
        torch.manual_seed(3)
        layer = MM.TransLayer(768).cuda().eval()
        x = x0.clone().requires_grad_(True)
        y = layer(x, prec)
        y.backward(up.clone())      # the residual add hands its upstream gradient on in place: a fresh one per run
        torch.cuda.synchronize()
        return y.detach().float(), x.grad.float(), {k: p.grad.float() for k, p in layer.named_parameters()}

    y1, dx1, g1 = run()
    Fn._TILE_SIDE, Fn._PINV_R32 = False, False
    try:
        y2, dx2, g2 = run()
    finally:
        Fn._TILE_SIDE, Fn._PINV_R32 = True, True
    assert torch.equal(y1, y2)
    for name, a, b in [("dx", dx1, dx2)] + [(k, g1[k], g2[k]) for k in g1]:
        d = float((a - b).norm()) / max(float(b.norm()), 1e-12)
        assert d <= 1e-3, (name, d)
