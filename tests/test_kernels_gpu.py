"""GPU: every C-ABI kernel against a plain PyTorch fp32/fp64 CPU reference of the same op."""
import itertools
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mirror_amd import kernels as K  # noqa: E402
from mirror_amd._lib import ACT_GELU, ACT_NONE, ACT_RELU, MH_BF16, MH_F32  # noqa: E402

DEV = "cuda"


def g(seed=0):
    return torch.Generator().manual_seed(seed)


def ints(shape, gen, lo=-3, hi=4):
    return torch.randint(lo, hi, shape, generator=gen).float()


def close(got, ref, rtol, atol, msg=""):
    got, ref = got.detach().float().cpu().double(), ref.detach().double()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bad.any(), f"{msg}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.3e} at {np.unravel_index(int(err.argmax()), err.shape) if err.numel() else ()}"


# --------------------------------------------------------------------------------------- GEMM
LAYOUTS = list(itertools.product([True, False], [True, False]))  # (a row-major?, b "KC" i.e. given as [N,K]^T)


def _mk(a_rm, shape, gen, dtype, integer):
    """Return (device view with math shape `shape`, cpu math tensor)."""
    r, c = shape[-2], shape[-1]
    base = ints(shape, gen) if integer else torch.randn(shape, generator=gen)
    if a_rm:
        dev = base.to(DEV, dtype)
    else:
        dev = base.transpose(-1, -2).contiguous().to(DEV, dtype).transpose(-1, -2)
    return dev, base


@pytest.mark.parametrize("mma,dtype,out_dtype", [
    (MH_F32, torch.float32, torch.float32),
    (MH_BF16, torch.bfloat16, torch.bfloat16),
    (MH_BF16, torch.bfloat16, torch.float32),
    (MH_BF16, torch.float32, torch.float32),
    (MH_BF16, torch.float32, torch.bfloat16),
])
@pytest.mark.parametrize("a_rm,b_t", LAYOUTS)
@pytest.mark.parametrize("M,N,Kd", [(128, 128, 64), (200, 72, 100), (37, 3000, 17), (256, 64, 4352), (5, 5, 4), (130, 257, 33)])
def test_gemm_layouts_dtypes_tails(mma, dtype, out_dtype, a_rm, b_t, M, N, Kd):
    gen = g(M * 7 + N * 3 + Kd)
    integer = mma == MH_BF16  # small integers are exact in bf16 -> bit-exact check of the MFMA data path
    a_dev, a = _mk(a_rm, (M, Kd), gen, dtype, integer)
    b_dev, b = _mk(not b_t, (Kd, N), gen, dtype, integer)
    out = K.gemm(a_dev, b_dev, mma=mma, out_dtype=out_dtype)
    ref = a.double() @ b.double()
    if integer:
        if out_dtype == torch.bfloat16:
            ref = ref.float().bfloat16().double()
        close(out, ref, 0, 0, "integer gemm must be exact")
    else:
        close(out, ref, 2e-5, 2e-5 * math.sqrt(Kd), "f32 gemm")


@pytest.mark.parametrize("a_rm,b_t", LAYOUTS)
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_large_tile_kernel(a_rm, b_t, out_dtype):
    """256 x 256 tiles (gemm_big.hip) take bf16 problems with M, N % 256 == 0: all layouts, batch, bias + ReLU,
    read-modify-write accumulate and split-K atomics, bit-exact on small integers."""
    gen = g(77 + a_rm + 2 * b_t)
    Bt, M, N, Kd = 3, 512, 768, 320
    bf = torch.bfloat16
    a_dev, a = _mk(a_rm, (Bt, M, Kd), gen, bf, True)
    b_dev, b = _mk(not b_t, (Bt, Kd, N), gen, bf, True)
    bias = ints((N,), gen)
    ref = a.double() @ b.double()
    out = K.gemm(a_dev, b_dev, mma=MH_BF16, out_dtype=out_dtype)
    close(out, ref.float().to(out_dtype).double(), 0, 0, "large tile plain")
    out = K.gemm(a_dev, b_dev, bias=bias.to(DEV), act=ACT_RELU, alpha=0.5, mma=MH_BF16, out_dtype=out_dtype)
    close(out, torch.relu(0.5 * ref + bias.double()).float().to(out_dtype).double(), 0, 0, "large tile bias+relu")
    base = ints((Bt, M, N), gen)
    acc = base.to(DEV, out_dtype)
    K.gemm(a_dev, b_dev, out=acc, accumulate=True, mma=MH_BF16)
    close(acc, (ref + base.double()).float().to(out_dtype).double(), 0, 0, "large tile accumulate")
    if out_dtype == torch.float32:
        dw = base[0].to(DEV).contiguous()
        K.gemm(a_dev, b_dev, out=dw.expand(Bt, M, N), accumulate=True, split_k=3, mma=MH_BF16)   # batch broadcast + split-K
        # 3 batches x 3 K-slices = 9 partial tiles, each f32 accumulator rounded ONCE to bf16 on its way to the fold pass (round 4):
        # bit-exact against the same sums with that rounding (slices of 128, 128, 64 of K = 320)
        want = base[0].double().clone()
        for z in range(Bt):
            for k0, k1 in ((0, 128), (128, 256), (256, 320)):
                want += (a[z][:, k0:k1].double() @ b[z][k0:k1].double()).float().bfloat16().double()
        close(dw, want, 0, 0, "large tile split-K partial tiles (bf16) + fold")
        assert float((dw.cpu().double() - ref.sum(0) - base[0].double()).abs().max()) <= 2 ** -8 * 9 * float(ref.abs().max())
    if a_rm:        # ragged M (K-contiguous A rows): the last row tile clamps its loads and guards its stores
        Mr = 4 * 256 + 37
        ar_dev, ar = _mk(True, (Bt, Mr, Kd), gen, bf, True)
        outr = torch.full((Bt, Mr + 3, N), 7.0, device=DEV, dtype=out_dtype)
        K.gemm(ar_dev, b_dev, out=outr[:, :Mr], mma=MH_BF16)
        # the library names the instance it launched (mh_gemm_variant_name)
        from mirror_amd import _lib
        name = _lib.load().mh_gemm_variant_name().decode()
        assert name.startswith(("gemm_pq_kernel<", "gemm_pp_kernel<", "gemm_big_kernel<", "gemm_kernel<1,bf16,bf16,")), name
        assert ("float" if out_dtype == torch.float32 else "bf16") in name.split(",", 3)[-1] or name.startswith("gemm_p"), name
        close(outr[:, :Mr], (ar.double() @ b.double()).float().to(out_dtype).double(), 0, 0, "large tile ragged M")
        assert bool((outr[:, Mr:] == 7.0).all()), "rows past M were written"


def test_split_k_bf16_partials_error_bound_under_cancellation(monkeypatch):
    """the split-K partial tiles of the f32 weight-gradient products are bf16 (kernels.SPLITK_PARTIALS), so the error of
    an element is bounded relative to the LARGEST partial, not to the final sum.  A weight-gradient-shaped product (K = 64 slices of
    1024 rows) whose slices cancel pairwise down to a small signal: the bf16-partial result against the f32-atomics result
    (MIRROR_SPLITK_PARTIALS=f32 / kernels.SPLITK_PARTIALS) and against f64, per element, with the bound the header states
    (parts * 2^-9 * max |partial|) — and the f32 policy itself at f32-summation accuracy."""
    gen = g(4242)
    bf = torch.bfloat16
    M, N, parts, ks = 512, 512, 64, 1024
    Kd = parts * ks
    # slice 2i and 2i + 1 hold (almost) opposite contributions: x_{2i+1} = -x_{2i} + small signal
    xa = (torch.randn(parts // 2, ks, M, generator=gen) * 1.0).to(bf)
    sig = (torch.randn(parts // 2, ks, M, generator=gen) * 2.0 ** -8).to(bf)
    xb = (-xa.float() + sig.float()).to(bf)
    x = torch.stack([xa, xb], 1).reshape(Kd, M)                     # [K, M]: dy^T rows
    y = (torch.randn(ks, N, generator=gen)).to(bf).repeat(parts, 1)  # the same activations under every slice: partials cancel pairwise
    xd, yd = x.to(DEV), y.to(DEV)
    ref = x.double().t() @ y.double()
    partial_max = max(float((x[i * ks:(i + 1) * ks].double().t() @ y[i * ks:(i + 1) * ks].double()).abs().max()) for i in range(0, parts, 8))
    assert partial_max > 20 * float(ref.abs().max()), "the construction must cancel"
    from mirror_amd import _lib
    outs = {}
    for mode in ("bf16", "f32"):
        monkeypatch.setattr(K, "SPLITK_PARTIALS", mode)
        dw = torch.zeros(M, N, device=DEV)
        K.gemm(xd.t(), yd, out=dw, accumulate=True, split_k=parts, mma=MH_BF16)
        name = _lib.load().mh_gemm_variant_name().decode()
        assert ("part" in name) == (mode == "bf16"), (mode, name)
        outs[mode] = dw.cpu().double()
    err_f32 = float((outs["f32"] - ref).abs().max())
    err_bf = float((outs["bf16"] - ref).abs().max())
    assert err_f32 <= 1e-5 * partial_max * parts ** 0.5, (err_f32, partial_max)          # f32 atomics: exact partial sums, f32 adds
    assert err_bf <= parts * 2.0 ** -9 * partial_max, (err_bf, partial_max)               # the documented bound
    assert float((outs["bf16"] - outs["f32"]).abs().max()) <= parts * 2.0 ** -9 * partial_max
    # and it IS relative to the partials: on this input the bf16 form is far from the f32 form relative to the final sum
    assert err_bf > 10 * err_f32


@pytest.mark.parametrize("a_rm,b_t", LAYOUTS)
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_persistent_kernel_epilogue_variants(a_rm, b_t, out_dtype):
    """gemm_pq_kernel (>= 128 tiles of 256 x 256: one workgroup per CU walks the units) carries ONE epilogue per instance, chosen by
    the host: bf16 tiles plain / alpha + bias / + ReLU, f32 C store / ReLU / accumulate, bf16 split-K partial tiles (plain / alpha).
    Every variant bit-exact on small integers, and the library says which kernel it launched."""
    from mirror_amd import _lib
    lib = _lib.load()
    gen = g(311 + a_rm + 2 * b_t)
    M, N, Kd = 4096, 2048, 320              # 16 x 8 = 128 units, 5 K-tiles each
    bf = torch.bfloat16
    a_dev, a = _mk(a_rm, (M, Kd), gen, bf, True)
    b_dev, b = _mk(not b_t, (Kd, N), gen, bf, True)
    bias = ints((N,), gen)
    ref = a.double() @ b.double()
    cast = lambda t: t.float().to(out_dtype).double()  # noqa: E731

    def launched(prefix):
        name = lib.mh_gemm_variant_name().decode()
        assert name.startswith(prefix), f"{name} (expected {prefix}...)"

    out = K.gemm(a_dev, b_dev, mma=MH_BF16, out_dtype=out_dtype)
    launched("gemm_pq_kernel<")
    close(out, cast(ref), 0, 0, "persistent plain")
    out = K.gemm(a_dev, b_dev, bias=bias.to(DEV), alpha=0.5, mma=MH_BF16, out_dtype=out_dtype)
    launched("gemm_pq_kernel<")
    close(out, cast(0.5 * ref + bias.double()), 0, 0, "persistent alpha + bias")
    out = K.gemm(a_dev, b_dev, bias=bias.to(DEV), act=ACT_RELU, alpha=0.5, mma=MH_BF16, out_dtype=out_dtype)
    launched("gemm_pq_kernel<")
    close(out, cast(torch.relu(0.5 * ref + bias.double())), 0, 0, "persistent alpha + bias + ReLU")
    base = ints((M, N), gen)
    acc = base.to(DEV, out_dtype)
    K.gemm(a_dev, b_dev, out=acc, accumulate=True, mma=MH_BF16)
    launched("gemm_pq_kernel<" if out_dtype == torch.float32 else "gemm_kernel<1,")      # accumulating bf16 C: the 128 x 128 kernel (f32 add, ONE rounding)
    close(acc, cast(ref + base.double()), 0, 0, "accumulate")
    if a_rm:        # ragged M: the last row tile clamps its loads and guards its stores
        Mr = M - 256 + 37
        outr = torch.full((Mr + 3, N), 7.0, device=DEV, dtype=out_dtype)
        K.gemm(a_dev[:Mr], b_dev, out=outr[:Mr], mma=MH_BF16)
        launched("gemm_pq_kernel<")
        close(outr[:Mr], cast(ref[:Mr]), 0, 0, "persistent ragged M")
        assert bool((outr[Mr:] == 7.0).all()), "rows past M were written"
    if out_dtype == torch.float32:
        # split-K into bf16 partial tiles + fold: few output tiles, a long contraction (the weight-gradient shape)
        Mw, Nw, Kw = 512, 512, 16384
        aw_dev, aw = _mk(a_rm, (Mw, Kw), gen, bf, True)
        bw_dev, bw = _mk(not b_t, (Kw, Nw), gen, bf, True)
        for alpha in (1.0, 0.5):
            dw = torch.zeros((Mw, Nw), device=DEV, dtype=torch.float32)
            K.gemm(aw_dev, bw_dev, out=dw, accumulate=True, split_k=32, alpha=alpha, mma=MH_BF16)
            launched("gemm_pq_kernel<")
            want = torch.zeros((Mw, Nw), dtype=torch.float64)
            for k0 in range(0, Kw, Kw // 32):
                want += (alpha * (aw[:, k0:k0 + Kw // 32].double() @ bw[k0:k0 + Kw // 32].double())).float().bfloat16().double()
            close(dw, want, 0, 0, f"persistent split-K partial tiles, alpha = {alpha}")


@pytest.mark.parametrize("a_rm,b_t", [(True, False), (False, False), (True, True)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_tile384_kernel(a_rm, b_t, out_dtype):
    """192 x 384 tiles (gemm_tile.hip: the batched 384-cubed products of the template's Moore-Penrose iteration): the three
    operand layouts the iteration uses, batch, alpha / diag / R addend (bf16 R beside an f32 C), read-modify-write
    accumulate and the bf16 copy of the final C — bit-exact on small integers."""
    gen = g(91 + a_rm + 2 * b_t)
    Bt, M, N, Kd = 5, 384, 768, 192
    bf = torch.bfloat16
    a_dev, a = _mk(a_rm, (Bt, M, Kd), gen, bf, True)
    b_dev, b = _mk(not b_t, (Bt, Kd, N), gen, bf, True)
    ref = a.double() @ b.double()
    out = K.gemm(a_dev, b_dev, mma=MH_BF16, out_dtype=out_dtype)
    close(out, ref.float().to(out_dtype).double(), 0, 0, "tile plain")
    assert K.gemm_tile_ok(M, N, Kd)
    Rm = ints((Bt, M, N), gen)
    eye = torch.zeros(M, N, dtype=torch.float64)
    eye[torch.arange(M), torch.arange(M)] = 1.0
    out = K.gemm(a_dev, b_dev, alpha=-1.0, diag=13.0, R=Rm.to(DEV, out_dtype), rcoef=2.0, mma=MH_BF16, out_dtype=out_dtype)
    close(out, (-ref + 13.0 * eye + 2.0 * Rm.double()).float().to(out_dtype).double(), 0, 0, "tile alpha + diag + R")
    base = ints((Bt, M, N), gen)
    acc = base.to(DEV, out_dtype)
    c2 = torch.zeros((Bt, M, N), device=DEV, dtype=bf)
    K.gemm(a_dev, b_dev, out=acc, accumulate=True, R=Rm.to(DEV, bf), rcoef=-7.0, mma=MH_BF16, c2=c2)   # bf16 R beside any C
    want = (ref + base.double() - 7.0 * Rm.double()).float()
    close(acc, want.to(out_dtype).double(), 0, 0, "tile accumulate + bf16 R")
    close(c2, want.to(bf).double(), 0, 0, "tile bf16 copy")
    if out_dtype == bf:      # an f32 R beside a bf16 C (mh_gemm_desc.r_bf16 = 2): an f32 partial sum joins the product and leaves as bf16
        R32 = (ints((Bt, M, N), gen) * 257.0 + 0.5).to(DEV)          # values bf16 cannot hold: read as f32 or the sum is wrong
        outb = torch.empty((Bt, M, N), device=DEV, dtype=bf)
        K.gemm(a_dev, b_dev, out=outb, R=R32, rcoef=1.0, mma=MH_BF16)
        close(outb, (ref + R32.double().cpu()).float().to(bf).double(), 0, 0, "tile bf16 C + f32 R")
    # K = 96 (the 96-wide heads of the template geometry): the second K-tile is half empty and reads as zeros
    a_dev, a = _mk(a_rm, (Bt, M, 96), gen, bf, True)
    b_dev, b = _mk(not b_t, (Bt, 96, N), gen, bf, True)
    out = K.gemm(a_dev, b_dev, alpha=0.5, mma=MH_BF16, out_dtype=out_dtype)
    close(out, (0.5 * (a.double() @ b.double())).float().to(out_dtype).double(), 0, 0, "tile ragged K")


@pytest.mark.parametrize("Kd", [96, 128])
def test_gemm_row_softmax_epilogue(Kd):
    """softmax=True on the 192 x 384 tile kernel (sim1 of the template geometry: rows of length m = 384): equals the f32 logits
    GEMM followed by mh_softmax_fwd -> bf16 to one bf16 ulp, and torch.softmax of the f64 product to bf16 rounding."""
    gen = g(211 + Kd)
    bf = torch.bfloat16
    Bt, M, N = 3, 576, 384
    a = (torch.randn(Bt, M, Kd, generator=gen)).to(bf)
    b = (torch.randn(Bt, N, Kd, generator=gen)).to(bf)
    a_dev, bT = a.to(DEV), b.to(DEV).transpose(-1, -2)
    scale = Kd ** -0.5
    assert K.gemm_softmax_ok(M, N, Kd)
    got = K.gemm(a_dev, bT, alpha=scale, mma=MH_BF16, out_dtype=bf, softmax=True)
    ref = torch.softmax(scale * (a.double() @ b.double().transpose(-1, -2)), dim=-1)
    two = K.softmax_fwd(K.gemm(a_dev, bT, alpha=scale, mma=MH_BF16, out_dtype=torch.float32), None, out_dtype=bf)
    assert float((got.float() - two.float()).abs().max()) <= 2.0 ** -8 * float(two.float().abs().max())
    err = (got.float().cpu().double() - ref).abs().max() / ref.abs().max()
    assert float(err) < 2.0 ** -7, float(err)
    assert float((got.float().sum(-1) - 1).abs().max()) < 2e-2
    # backward form: dS = P o (dP - rowsum(P o dP)), dP = alpha a b^T, P given in bf16
    P = got
    want = K.softmax_bwd(P, K.gemm(a_dev, bT, alpha=scale, mma=MH_BF16, out_dtype=bf))       # the two-launch path, in place on dP
    dS = K.gemm(a_dev, bT, alpha=scale, mma=MH_BF16, out_dtype=bf, softmax_bwd_of=P)
    ref_d = P.float().cpu().double() * ((scale * (a.double() @ b.double().transpose(-1, -2)))
                                        - (P.float().cpu().double() * (scale * (a.double() @ b.double().transpose(-1, -2)))).sum(-1, keepdim=True))
    assert float((dS.float().cpu().double() - ref_d).abs().max()) < 2.0 ** -6 * float(ref_d.abs().max())
    assert float((dS.float() - want.float()).abs().max()) < 2.0 ** -5 * float(want.float().abs().max())


@pytest.mark.parametrize("a_rm,b_t", [(True, False), (False, False), (True, True)])
def test_gemm_ragged_n_instances(a_rm, b_t):
    """N = 96 (one column tile that ends inside: the attn·v / dS·k products of the template's 96-wide heads) with M and K whole
    tiles: FULL == 3 instances of the 128 x 128 kernel (only B's loads and the stores are guarded) — bit-exact on small integers,
    bf16 and f32 C, accumulate."""
    gen = g(171 + a_rm + 2 * b_t)
    bf = torch.bfloat16
    Bt, M, N, Kd = 3, 384, 96, 192
    a_dev, a = _mk(a_rm, (Bt, M, Kd), gen, bf, True)
    b_dev, b = _mk(not b_t, (Bt, Kd, N), gen, bf, True)
    ref = a.double() @ b.double()
    for od in (bf, torch.float32):
        out = torch.full((Bt, M, N + 8), 5.0, device=DEV, dtype=od)        # columns behind N must stay untouched
        K.gemm(a_dev, b_dev, out=out[..., :N], alpha=0.5, mma=MH_BF16)
        close(out[..., :N], (0.5 * ref).float().to(od).double(), 0, 0, "ragged N")
        assert bool((out[..., N:] == 5.0).all())
    base = ints((Bt, M, N), gen)
    acc = base.to(DEV)
    K.gemm(a_dev, b_dev, out=acc, accumulate=True, mma=MH_BF16)
    close(acc, ref + base.double(), 0, 0, "ragged N accumulate")


@pytest.mark.parametrize("a_rm,b_t", [(True, True), (True, False), (False, False)])
def test_gemm_ksum_sums_operand_pairs_in_one_launch(a_rm, b_t):
    """K.gemm_ksum (k_segments of mh_gemm_desc): sum_s A_s B_s over operand pairs a constant stride apart, one launch, one
    accumulator — bit-exact on small integers, with an R addend, the bf16 copy and a strided stack."""
    gen = g(131 + a_rm + 2 * b_t)
    S, Bt, M, N, Kd = 3, 4, 384, 384, 128
    bf = torch.bfloat16
    a = ints((S, Bt, M, Kd), gen)
    b = ints((S, Bt, Kd, N), gen)
    a_dev = a.to(DEV, bf) if a_rm else a.transpose(-1, -2).contiguous().to(DEV, bf).transpose(-1, -2)
    b_dev = b.to(DEV, bf) if not b_t else b.transpose(-1, -2).contiguous().to(DEV, bf).transpose(-1, -2)
    ref = (a.double() @ b.double()).sum(0)
    out = K.gemm_ksum(a_dev, b_dev, mma=MH_BF16, out_dtype=torch.float32)
    close(out, ref, 0, 0, "ksum plain")
    Rm = ints((Bt, M, N), gen)
    c2 = torch.zeros((Bt, M, N), device=DEV, dtype=bf)
    out = K.gemm_ksum(a_dev, b_dev, alpha=-1.0, R=Rm.to(DEV, bf), rcoef=-7.0, mma=MH_BF16, out_dtype=torch.float32, c2=c2)
    want = (-ref - 7.0 * Rm.double()).float()
    close(out, want.double(), 0, 0, "ksum alpha + bf16 R")
    close(c2, want.to(bf).double(), 0, 0, "ksum bf16 copy")
    # every second pair of a longer stack (pair stride = two matrices)
    out = K.gemm_ksum(a_dev[::2], b_dev[::2], mma=MH_BF16, out_dtype=torch.float32)
    close(out, (a[::2].double() @ b[::2].double()).sum(0), 0, 0, "ksum strided stack")


def test_pinv_tile_path_matches_generic_path():
    """m = 384 (the template's landmark count): pinv_forward_tile / pinv_backward_tile (one 192 x 384-tile launch per product)
    against the generic bf16 path they replace and, loosely, against f64 autograd through the same iteration."""
    from mirror_amd import functional as Fn
    gen = g(5)
    m, BH, iters = 384, 6, 6
    logits = torch.randn(BH, m, m, generator=gen) + 4.0 * torch.eye(m)
    a2 = torch.softmax(logits, dim=-1).to(DEV).contiguous()
    dZ = (torch.randn(BH, m, m, generator=gen) * 0.1).to(DEV)
    z_t, saved_t, st_t = Fn.pinv_forward_tile(a2, iters)
    z_g, saved_g, st_g = Fn.pinv_forward(a2, iters, MH_BF16, torch.bfloat16)
    assert torch.equal(st_t, st_g)
    close(z_t.float(), z_g.double().cpu(), 3e-2, 3e-2 * float(z_g.float().abs().max()), "pinv tile forward")
    dX_t = Fn.pinv_backward_tile(a2, saved_t, st_t, dZ)
    Fn._PINV_R32 = False       # the f32 sums as read-modify-write destinations (the form before round 5): the same numbers up to the
    try:                       # order of mh_pinv_z0_bwd's atomics (two runs of either form differ by ~1e-5 at a scale of ~160)
        close(Fn.pinv_backward_tile(a2, saved_t, st_t, dZ), dX_t.double().cpu(), 0, 2e-4, "pinv tile backward: f32 sums as addends vs destinations")
    finally:
        Fn._PINV_R32 = True
    dX_g = Fn.pinv_backward(a2, saved_g, st_g, dZ, MH_BF16, torch.bfloat16)
    x64 = a2.double().cpu().requires_grad_(True)
    ax = x64.abs()
    z = x64.transpose(-1, -2) / (ax.sum(-1).max() * ax.sum(-2).max())
    eye = torch.eye(m, dtype=torch.float64)
    for _ in range(iters):
        xz = x64 @ z
        z = 0.25 * z @ (13 * eye - xz @ (15 * eye - xz @ (7 * eye - xz)))
    z.backward(dZ.double().cpu())
    ref = x64.grad

    def cos(a, b):
        a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
        return float(torch.dot(a, b) / (a.norm() * b.norm())), float(a.norm() / b.norm())
    c_t, r_t = cos(dX_t, ref)
    c_g, r_g = cos(dX_g, ref)
    assert c_t > 0.995 and 0.97 < r_t < 1.03, (c_t, r_t)
    assert c_t > c_g - 2e-3, (c_t, c_g)          # no worse than the path it replaces
    close(z_t.float(), z.detach(), 3e-2, 3e-2 * float(z.detach().abs().max()), "pinv tile forward vs f64")


@pytest.mark.parametrize("mma,dtype", [(MH_F32, torch.float32), (MH_BF16, torch.bfloat16)])
def test_gemm_batched_strided_views(mma, dtype):
    """The Nystrom use: heads are column slices of a [B, n, 3D] buffer; output written into a [B, n, D] view."""
    gen = g(5)
    B, n, h, dh, m = 2, 96, 4, 16, 24
    D = h * dh
    qkv = ints((B, n, 3 * D), gen)
    lm = ints((B, m, 2 * D), gen)
    qkv_d, lm_d = qkv.to(DEV, dtype), lm.to(DEV, dtype)
    q = qkv_d.view(B, n, 3, h, dh)[:, :, 0].permute(0, 2, 1, 3)        # [B,h,n,dh]
    kl = lm_d.view(B, m, 2, h, dh)[:, :, 1].permute(0, 2, 1, 3)        # [B,h,m,dh]
    sim = K.gemm(q, kl.transpose(-1, -2), alpha=0.5, mma=mma, out_dtype=torch.float32)   # [B,h,n,m]
    qc = qkv.view(B, n, 3, h, dh)[:, :, 0].permute(0, 2, 1, 3).double()
    klc = lm.view(B, m, 2, h, dh)[:, :, 1].permute(0, 2, 1, 3).double()
    close(sim, 0.5 * qc @ klc.transpose(-1, -2), 0, 0, "sim1")
    # attn-like [B,h,n,m] @ [B,h,m,dh] -> columns of a [B,n,D] buffer
    w2 = ints((B, h, m, dh), gen).to(DEV, dtype)
    a1 = ints((B, h, n, m), gen, 0, 2).to(DEV, dtype)
    out = torch.zeros((B, n, D), device=DEV, dtype=dtype)
    K.gemm(a1, w2, out=out.view(B, n, h, dh).permute(0, 2, 1, 3), mma=mma)
    ref = (a1.cpu().double() @ w2.cpu().double()).permute(0, 2, 1, 3).reshape(B, n, D)
    close(out, ref, 0, 0, "attn@w2 into head columns")
    # K-strided both: a1^T @ dout  (TN, the weight-gradient shape)
    dout = ints((B, h, n, dh), gen).to(DEV, dtype)
    dw2 = K.gemm(a1.transpose(-1, -2), dout, mma=mma, out_dtype=torch.float32)
    close(dw2, a1.cpu().double().transpose(-1, -2) @ dout.cpu().double(), 0, 0, "TN batched")
    # broadcast weight across batch
    W = ints((D, D), gen).to(DEV, dtype)
    y = K.gemm(out, W.t(), mma=mma, out_dtype=torch.float32)
    close(y, out.cpu().double() @ W.cpu().double().t(), 0, 0, "broadcast weight")


@pytest.mark.parametrize("mma,dtype", [(MH_F32, torch.float32), (MH_BF16, torch.bfloat16)])
def test_gemm_epilogues_and_split_k(mma, dtype):
    gen = g(9)
    M, N, Kd = 150, 130, 96
    a, b = ints((M, Kd), gen), ints((N, Kd), gen)
    bias = ints((N,), gen)
    a_d, b_d, bias_d = a.to(DEV, dtype), b.to(DEV, dtype), bias.to(DEV)
    base = a.double() @ b.double().t()
    y = K.gemm(a_d, b_d.t(), bias=bias_d, act=ACT_RELU, mma=mma, out_dtype=torch.float32)
    close(y, F.relu(base + bias.double()), 0, 0, "bias+relu")
    y = K.gemm(a_d, b_d.t(), bias=bias_d, alpha=0.125, mma=mma, out_dtype=torch.float32)
    close(y, 0.125 * base + bias.double(), 0, 0, "alpha+bias")
    from mirror_amd import MirrorHipError
    with pytest.raises(MirrorHipError):  # GELU is not fused into the GEMM epilogue (mh_gelu_fwd)
        K.gemm(a_d, b_d.t(), act=ACT_GELU, mma=mma)
    # a FULL-tile problem exercises the LDS-staged wide-store epilogue, incl. read-modify-write accumulate
    af, bf_ = ints((256, 128), gen).to(DEV, dtype), ints((128, 128), gen).to(DEV, dtype)
    cf = ints((256, 128), gen).to(DEV, dtype)
    ref_f = af.cpu().double() @ bf_.cpu().double().t() + ints((128,), g(1)).double()
    y = K.gemm(af, bf_.t(), bias=ints((128,), g(1)).to(DEV), act=ACT_RELU, mma=mma)
    close(y, F.relu(ref_f), 0, 0, "full tile bias+relu")
    acc_f = cf.clone()
    K.gemm(af, bf_.t(), out=acc_f, accumulate=True, mma=mma)
    exp = cf.cpu().double() + af.cpu().double() @ bf_.cpu().double().t()
    if dtype == torch.bfloat16:
        exp = exp.float().bfloat16().double()
    close(acc_f, exp, 0, 0, "full tile accumulate")
    sq = ints((64, 64), gen).to(DEV, dtype)
    y = K.gemm(sq, sq, alpha=-1.0, diag=15.0, mma=mma, out_dtype=torch.float32)
    close(y, 15 * torch.eye(64, dtype=torch.float64) - sq.cpu().double() @ sq.cpu().double(), 0, 0, "diag")
    # accumulate and split-K (weight gradient: dW[N,K] += dY^T X over many rows)
    rows = 3000
    dy, x = ints((rows, 70), gen), ints((rows, 50), gen)
    dy_d, x_d = dy.to(DEV, dtype), x.to(DEV, dtype)
    acc = torch.ones((70, 50), device=DEV)
    K.gemm(dy_d.t(), x_d, out=acc, accumulate=True, split_k=7, mma=mma)
    close(acc, 1 + dy.double().t() @ x.double(), 0, 0, "split-k atomics")
    acc2 = torch.ones((70, 50), device=DEV)
    K.gemm(dy_d.t(), x_d, out=acc2, accumulate=True, mma=mma)
    close(acc2, 1 + dy.double().t() @ x.double(), 0, 0, "accumulate")


def test_gemm_f32_accuracy_vs_fp64():
    gen = g(11)
    a, b = torch.randn(300, 1024, generator=gen), torch.randn(1024, 200, generator=gen)
    y = K.gemm(a.to(DEV), b.to(DEV), mma=MH_F32)
    # exact-f32 fma chain: error ~1e-7 * sum|a*b| (~650 here), cf. cdna_hip_programming.md §3
    close(y, a.double() @ b.double(), 1e-5, 3e-4, "f32 MFMA K=1024")
    y = K.gemm(a.to(DEV).bfloat16(), b.to(DEV).bfloat16(), mma=MH_BF16, out_dtype=torch.float32)
    ref = a.bfloat16().double() @ b.bfloat16().double()
    close(y, ref, 1e-5, 1e-3, "bf16 MFMA on bf16-rounded data, f32 accumulate")


# --------------------------------------------------------------------------------------- row kernels
@pytest.mark.parametrize("D", [32, 96, 512, 1536])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_fwd_bwd(D, dtype):
    gen = g(D)
    B, rpb, pad = 3, 7, 2
    x = torch.randn(B, rpb, D, generator=gen) * 2 + 0.5
    gam, bet = torch.randn(D, generator=gen), torch.randn(D, generator=gen)
    xd = x.to(DEV)  # residual stream is f32 in both modes
    y = torch.zeros((B, rpb + pad, D), device=DEV, dtype=dtype)
    mean = torch.empty(B * rpb, device=DEV)
    rstd = torch.empty(B * rpb, device=DEV)
    K.layernorm_fwd(xd, gam.to(DEV), bet.to(DEV), y[:, pad:], mean, rstd, B, rpb, D, rpb * D, (rpb + pad) * D, 1e-5)
    xr = x.clone().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (D,), gr, br, 1e-5)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    close(y[:, pad:], ref, tol, tol, "ln fwd")
    assert float(y[:, :pad].abs().max()) == 0.0
    dy = torch.randn(B, rpb + pad, D, generator=gen)
    ref.backward(dy[:, pad:])
    dyd = dy.to(DEV, dtype)
    dx = torch.empty_like(xd)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    K.layernorm_bwd(dyd[:, pad:], xd, gam.to(DEV), mean, rstd, dx, dg, db, B, rpb, D, rpb * D, (rpb + pad) * D)
    close(dx, xr.grad, 10 * tol, 10 * tol, "ln dx")
    close(dg, gr.grad, 10 * tol, 20 * tol, "ln dgamma")
    close(db, br.grad, 10 * tol, 20 * tol, "ln dbeta")


@pytest.mark.parametrize("D,with_cls,acc", [(512, True, False), (1024, False, False), (1536, True, True)])
def test_layernorm_bwd_with_the_fanout_sum_inside(D, with_cls, acc):
    """mh_layernorm_bwd_fan against mh_fanout_bwd + mh_layernorm_bwd: the encoder output's three gradients (decoder input, retention
    target = rows 1.. as alpha * bf16 source, cls row; models/mirror.py:684-700) summed while the LayerNorm backward reads dy, versus
    the [B, T, D] sum tensor written first.  The sums are the same f32 additions in the same order per element."""
    gen = g(D + 7)
    B, T = 3, 67
    x = (torch.randn(B, T, D, generator=gen) * 2 + 0.5).to(DEV)
    gam, bet = torch.randn(D, generator=gen).to(DEV), torch.randn(D, generator=gen).to(DEV)
    y = torch.empty((B, T, D), device=DEV)
    mean, rstd = torch.empty(B * T, device=DEV), torch.empty(B * T, device=DEV)
    K.layernorm_fwd(x, gam, bet, y, mean, rstd, B, T, D, T * D, T * D, 1e-5)
    gf = torch.randn(B, T, D, generator=gen).to(DEV)
    src = torch.randn(B, T - 1, D, generator=gen).to(DEV, torch.bfloat16)
    cls = torch.randn(B, D, generator=gen).to(DEV) if with_cls else None
    alpha = -1.0
    base = torch.randn(B, T, D, generator=gen).to(DEV)
    dE = K.fanout_bwd(gf, src, alpha, cls, B, T, D)
    dx0 = base.clone() if acc else torch.empty_like(x)
    dg0, db0 = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    K.layernorm_bwd(dE, x, gam, mean, rstd, dx0, dg0, db0, B, T, D, T * D, T * D, accumulate_dx=acc)
    dx1 = base.clone() if acc else torch.empty_like(x)
    dg1, db1 = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    assert K.layernorm_bwd_fan_ok(gf, x, dx1, src, cls, B, T, D)
    K.layernorm_bwd(gf, x, gam, mean, rstd, dx1, dg1, db1, B, T, D, T * D, T * D, accumulate_dx=acc, fan=(src, alpha, cls))
    close(dx1, dx0.cpu(), 1e-6, 1e-6, "dx with the fan-out inside")
    close(dg1, dg0.cpu(), 1e-6, 1e-5, "dgamma")
    close(db1, db0.cpu(), 1e-6, 1e-5, "dbeta")
    # the decoder's data gradient arrives in bf16 (round 5): the same launch reads bf16 dy, against the f32 copy of those bf16 values
    gfb = gf.to(torch.bfloat16)
    dx2 = base.clone() if acc else torch.empty_like(x)
    dg2, db2 = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    K.layernorm_bwd(gfb, x, gam, mean, rstd, dx2, dg2, db2, B, T, D, T * D, T * D, accumulate_dx=acc, fan=(src, alpha, cls))
    dx3 = base.clone() if acc else torch.empty_like(x)
    dg3, db3 = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    K.layernorm_bwd(gfb.float(), x, gam, mean, rstd, dx3, dg3, db3, B, T, D, T * D, T * D, accumulate_dx=acc, fan=(src, alpha, cls))
    close(dx2, dx3.cpu(), 1e-6, 1e-6, "dx, bf16 dy")
    close(dg2, dg3.cpu(), 1e-6, 1e-5, "dgamma, bf16 dy")
    with pytest.raises(K.MirrorHipError):       # off the form: loud
        K.layernorm_bwd(gf, x, gam, mean, rstd, dx1, dg1, db1, B, T, D, T * D, T * D, fan=(src.float(), alpha, cls))


@pytest.mark.parametrize("D,dy_dtype,with_fan", [(512, torch.float32, True), (512, torch.bfloat16, False), (1024, torch.float32, False)])
def test_layernorm_bwd_with_to_outs_dropout_backward_inside(D, dy_dtype, with_fan):
    """mh_layernorm_bwd_drop against mh_layernorm_bwd(_fan) + mh_dropout_lite_colsum: x is the output of resid + Dropout(to_out(.))
    ([3P] to_out + TransLayer's residual, models/mirror.py:312), so the LayerNorm backward's dx is that Dropout's upstream gradient:
    the masked bf16 gradient must be BIT-equal to the two-launch path's (same Philox block per element, same rounding), its column
    sums equal up to summation order, dx / dgamma / dbeta untouched; the device base of a graphed step shifts the masks the same way."""
    gen = g(D + 11)
    B, T = 3, 67
    x = (torch.randn(B, T, D, generator=gen) * 2 + 0.5).to(DEV)
    gam, bet = torch.randn(D, generator=gen).to(DEV), torch.randn(D, generator=gen).to(DEV)
    y = torch.empty((B, T, D), device=DEV)
    mean, rstd = torch.empty(B * T, device=DEV), torch.empty(B * T, device=DEV)
    K.layernorm_fwd(x, gam, bet, y, mean, rstd, B, T, D, T * D, T * D, 1e-5)
    dy = torch.randn(B, T, D, generator=gen).to(DEV, dy_dtype)
    fan = (torch.randn(B, T - 1, D, generator=gen).to(DEV, torch.bfloat16), -1.0, torch.randn(B, D, generator=gen).to(DEV)) if with_fan else None
    p, seed, off = 0.1, 4242, 4096
    for base in (None, torch.tensor([1000 * 8], device=DEV, dtype=torch.int64)):
        dx0 = torch.empty_like(x)
        dg0, db0 = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
        K.layernorm_bwd(dy, x, gam, mean, rstd, dx0, dg0, db0, B, T, D, T * D, T * D, fan=fan)
        gb0, bias0 = torch.empty(B, T, D, device=DEV, dtype=torch.bfloat16), torch.full((D,), 0.5, device=DEV)
        K.dropout_lite_colsum(dx0, p, seed, off, base, gb0, bias0)
        dx1 = torch.empty_like(x)
        dg1, db1 = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
        gb1, bias1 = torch.full((B, T, D), float("nan"), device=DEV, dtype=torch.bfloat16), torch.full((D,), 0.5, device=DEV)
        assert K.layernorm_bwd_drop_ok(dy, x, dx1, gb1, bias1, B, T, D, off)
        K.layernorm_bwd(dy, x, gam, mean, rstd, dx1, dg1, db1, B, T, D, T * D, T * D, fan=fan, drop=(gb1, p, seed, off, base, bias1))
        assert torch.equal(dx1, dx0) and torch.equal(gb1, gb0)
        assert 0.05 < float((gb1 == 0).float().mean()) < 0.15
        close(dg1, dg0.cpu(), 1e-6, 1e-5, "dgamma")
        close(db1, db0.cpu(), 1e-6, 1e-5, "dbeta")
        close(bias1, bias0.cpu(), 1e-5, 1e-4, "to_out bias gradient")
    with pytest.raises(K.MirrorHipError):
        K.layernorm_bwd(dy, x, gam, mean, rstd, dx1, dg1, db1, B, T, D, T * D, T * D, drop=(gb1, p, seed, off + 4, None, bias1))
    if not with_fan:
        # a norm over the FIRST rows of a longer (square-padded) sequence: the Dropout's tensor has Tx > T rows per batch, its element
        # indices (and so its masks) follow Tx; the rows behind the norm's are the caller's zeros
        Tx = T + 5
        xx = torch.zeros(B, Tx, D, device=DEV)
        xx[:, :T] = x
        dxa = torch.zeros_like(xx)
        K.layernorm_bwd(dy, xx, gam, mean, rstd, dxa, torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), B, T, D, Tx * D, T * D)
        gba, biasa = torch.empty(B, Tx, D, device=DEV, dtype=torch.bfloat16), torch.zeros(D, device=DEV)
        K.dropout_lite_colsum(dxa, p, seed, off, None, gba, biasa)
        dxb = torch.zeros_like(xx)
        gbb, biasb = torch.zeros(B, Tx, D, device=DEV, dtype=torch.bfloat16), torch.zeros(D, device=DEV)
        K.layernorm_bwd(dy, xx, gam, mean, rstd, dxb, torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), B, T, D, Tx * D, T * D,
                        drop=(gbb, p, seed, off, None, biasb))
        assert torch.equal(dxb, dxa) and torch.equal(gbb, gba)
        close(biasb, biasa.cpu(), 1e-5, 1e-4, "to_out bias gradient, padded rows")


def test_layernorm_bwd_lm_relu_rows_and_their_column_sums():
    """mh_layernorm_bwd_lm(relu_out, relu_db): rows 1 .. R of x are a ReLU's output — their gradient leaves as bf16 (x > 0 ? dx : 0) and
    relu_db receives the column sums of exactly those stored values (_fc1's bias gradient, models/mirror.py:346), the other outputs are
    those of the launch without relu_db."""
    gen = g(77)
    B, T, D, l = 2, 129, 512, 2
    pad = (l - T % l) % l
    m = (pad + T) // l
    x = torch.randn(B, T, D, generator=gen).to(DEV)
    x[:, 1:] = x[:, 1:].clamp_min(0)
    gam = torch.randn(D, generator=gen).to(DEV)
    mean, rstd = x.mean(-1).reshape(-1).contiguous(), (x.var(-1, unbiased=False) + 1e-5).rsqrt().reshape(-1).contiguous()
    dy = torch.randn(B, pad + T, D, generator=gen).to(DEV, torch.bfloat16)
    gadd = torch.randn(B, m, D, generator=gen).to(DEV, torch.bfloat16)
    outs = []
    for with_db in (False, True):
        G = torch.ones(B, T, D, device=DEV)
        dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
        dh = torch.empty(B, T - 1, D, device=DEV, dtype=torch.bfloat16)
        rdb = torch.full((D,), 2.0, device=DEV) if with_db else None
        K.layernorm_bwd(dy[:, pad:], x, gam, mean, rstd, G, dg, db, B, T, D, T * D, (pad + T) * D, accumulate_dx=True, gadd=gadd, pad=pad, l=l,
                        relu_out=dh, relu_first=1, relu_db=rdb)
        outs.append((G[:, 0].clone(), dg, db, dh, rdb))
    (c0, g0, b0, h0, _), (c1, g1, b1, h1, rdb) = outs
    assert torch.equal(h0, h1) and torch.equal(c0, c1)
    close(g1, g0.cpu(), 1e-6, 1e-5, "dgamma")
    close(b1, b0.cpu(), 1e-6, 1e-5, "dbeta")
    ref = 2.0 + h1.double().sum((0, 1))
    close(rdb, ref.float().cpu(), 1e-5, 1e-4, "relu_db = colsum(relu_out)")
    assert float((h1.float() * (x[:, 1:] <= 0)).abs().max()) == 0.0


@pytest.mark.parametrize("cols", [16, 256, 1000, 4352])
@pytest.mark.parametrize("din,dout", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16)])
def test_softmax_fwd_bwd(cols, din, dout):
    gen = g(cols)
    x = torch.randn(2, 3, 5, cols, generator=gen) * 3
    y = K.softmax_fwd(x.to(DEV, din), out_dtype=dout)
    ref = x.softmax(-1)
    tol = 1e-6 if dout == torch.float32 else 1e-2
    close(y, ref, tol * 10, tol * 0.1, "softmax fwd")
    dy = torch.randn(x.shape, generator=gen)
    yc = y.float().cpu()
    dx = K.softmax_bwd(y, dy.to(DEV, dout).clone())
    dyc = dy.to(dout).float()
    close(dx, yc * (dyc - (dyc * yc).sum(-1, keepdim=True)), 2e-2 if dout == torch.bfloat16 else 1e-5, 1e-5 if dout == torch.float32 else 2e-3, "softmax bwd")


def test_l2norm():
    gen = g(3)
    x = torch.randn(4, 6, 40, generator=gen)
    xd = x.to(DEV)
    y, nrm = K.l2norm_fwd(xd, 4, 40, 6 * 40, 1e-12, torch.float32)  # row 0 of each batch
    xr = x.clone().requires_grad_(True)
    ref = F.normalize(xr, dim=-1)[:, 0]
    close(y, ref, 1e-6, 1e-6, "l2 fwd")
    dy = torch.randn(4, 40, generator=gen)
    ref.backward(dy)
    dx = torch.zeros_like(xd)
    K.l2norm_bwd(y, nrm, dy.to(DEV), dx, 4, 40, 6 * 40, accumulate=True)
    close(dx, xr.grad, 1e-5, 1e-6, "l2 bwd")
    # L2NormRowFn on the cls rows of a [B, T, D] buffer handed over as a strided [B, D] view: read in place, gradient written whole
    from mirror_amd import functional as Fn
    xs = xd.clone().requires_grad_(True)
    rows = xs[:, 0]
    assert not rows.is_contiguous()
    ys = Fn.L2NormRowFn.apply(rows, 1e-12, torch.float32)
    close(ys, ref.detach(), 1e-6, 1e-6, "L2NormRowFn strided rows")
    ys.backward(dy.to(DEV))
    close(xs.grad, xr.grad, 1e-5, 1e-6, "L2NormRowFn strided rows bwd")
    x3 = xd.clone().requires_grad_(True)
    y3 = Fn.L2NormRowFn.apply(x3, 1e-12, torch.float32)          # the 3-D form: the other rows' gradient is zero
    y3.backward(dy.to(DEV))
    close(x3.grad, xr.grad, 1e-5, 1e-6, "L2NormRowFn 3-D bwd")


# --------------------------------------------------------------------------------------- Nystrom pieces
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_landmarks(dtype):
    gen = g(1)
    B, n_p, D, l = 2, 24, 16, 3
    qkv = ints((B, n_p, 3 * D), gen)
    lm = K.landmark_fwd(qkv.to(DEV, dtype), l)
    ref = qkv[..., : 2 * D].reshape(B, n_p // l, l, 2 * D).mean(2)
    close(lm, ref, 1e-2 if dtype == torch.bfloat16 else 1e-6, 1e-6, "landmark fwd")
    dlm = ints((B, n_p // l, 2 * D), gen, -6, 7) * 3
    dq = ints((B, n_p, 3 * D), gen)
    dq_d = dq.to(DEV, dtype)
    K.landmark_bwd(dlm.to(DEV, dtype), dq_d, l)
    exp = dq.clone()
    exp[..., : 2 * D] += dlm.repeat_interleave(l, dim=1) / l
    close(dq_d, exp, 1e-2 if dtype == torch.bfloat16 else 1e-6, 1e-6, "landmark bwd")


@pytest.mark.parametrize("dh,n_p", [(4, 32), (64, 100), (32, 384)])
def test_resconv_fwd_adjoint_wgrad(dh, n_p):
    gen = g(dh)
    B, h, taps = 2, 8, 33
    D = h * dh
    qkv = torch.randn(B, n_p, 3 * D, generator=gen)
    w = torch.randn(h, 1, taps, 1, generator=gen) * 0.2
    v = qkv[..., 2 * D:].reshape(B, n_p, h, dh).permute(0, 2, 1, 3).contiguous().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(v, wr, padding=(taps // 2, 0), groups=h)          # [B,h,n_p,dh]
    qkv_d = qkv.to(DEV)
    out = torch.ones((B, n_p, D), device=DEV)
    K.resconv(qkv_d[..., 2 * D:], w.to(DEV), out, h, transpose=False, accumulate=True)
    close(out, 1 + ref.permute(0, 2, 1, 3).reshape(B, n_p, D), 1e-5, 1e-5, "resconv fwd")
    dout = torch.randn(B, n_p, D, generator=gen)
    ref.backward(dout.reshape(B, n_p, h, dh).permute(0, 2, 1, 3))
    dqkv = torch.zeros((B, n_p, 3 * D), device=DEV)
    K.resconv(dout.to(DEV), w.to(DEV), dqkv[..., 2 * D:], h, transpose=True, accumulate=True)
    close(dqkv[..., 2 * D:], v.grad.permute(0, 2, 1, 3).reshape(B, n_p, D), 1e-5, 1e-5, "resconv adjoint")
    dw = torch.zeros((h, taps), device=DEV)
    K.resconv_wgrad(qkv_d[..., 2 * D:], dout.to(DEV), dw, h)
    close(dw, wr.grad.reshape(h, taps), 1e-4, 1e-3, "resconv wgrad")


@pytest.mark.parametrize("n_p", [256, 100, 4352])
def test_resconv_mfma_path_bf16(n_p):
    """bf16 / dh = 64 / 33 taps runs as a banded Toeplitz product on the matrix cores (resconv_mfma.hip): forward,
    transposed (data-gradient) form and the weight gradient against conv2d autograd on the bf16-rounded operands."""
    gen = g(n_p)
    B, h, dh, taps = 2, 8, 64, 33
    D = h * dh
    bf = torch.bfloat16
    qkv = torch.randn(B, n_p, 3 * D, generator=gen).to(bf)
    w = (torch.randn(h, 1, taps, 1, generator=gen) * 0.2)
    wq = w.to(bf).float()                                               # the kernel feeds bf16 weights to the MFMA
    v = qkv[..., 2 * D:].float().reshape(B, n_p, h, dh).permute(0, 2, 1, 3).contiguous().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    from oracle import mirror_oracle as O
    with O.exact_cpu_convs():     # torch's oneDNN CPU conv2d weight gradient is wrong at n_p = 256
        ref = F.conv2d(v, wr, padding=(taps // 2, 0), groups=h)
    qkv_d = qkv.to(DEV)
    base = torch.randn(B, n_p, D, generator=gen).to(bf)
    out = base.to(DEV).clone()
    K.resconv(qkv_d[..., 2 * D:], w.to(DEV), out, h, transpose=False, accumulate=True)
    want = base.float() + ref.detach().permute(0, 2, 1, 3).reshape(B, n_p, D)
    close(out, want, 1e-2, 2e-2, "resconv fwd (mfma, accumulate)")
    out2 = torch.full((B, n_p, D), float("nan"), device=DEV, dtype=bf)
    K.resconv(qkv_d[..., 2 * D:], w.to(DEV), out2, h, transpose=False, accumulate=False)
    close(out2, ref.detach().permute(0, 2, 1, 3).reshape(B, n_p, D), 1e-2, 2e-2, "resconv fwd (mfma, store)")
    dout = torch.randn(B, n_p, D, generator=gen).to(bf)
    with O.exact_cpu_convs():
        ref.backward(dout.float().reshape(B, n_p, h, dh).permute(0, 2, 1, 3))
    dqkv = torch.zeros((B, n_p, 3 * D), device=DEV, dtype=bf)
    K.resconv(dout.to(DEV), w.to(DEV), dqkv[..., 2 * D:], h, transpose=True, accumulate=True)
    close(dqkv[..., 2 * D:], v.grad.permute(0, 2, 1, 3).reshape(B, n_p, D), 1e-2, 2e-2, "resconv adjoint (mfma)")
    dw = torch.zeros((h, taps), device=DEV)
    K.resconv_wgrad(qkv_d[..., 2 * D:], dout.to(DEV), dw, h)
    gref = wr.grad.reshape(h, taps)
    close(dw, gref, 2e-3, 2e-3 * float(gref.abs().max()), "resconv wgrad (mfma)")


@pytest.mark.parametrize("n_p", [256, 100, 1152, 4352])
def test_resconv_bwd_one_pass_equals_the_two_passes(n_p):
    """mh_resconv_bwd (round 5): the adjoint conv accumulated into the v columns of d qkv AND the tap gradient in one pass over dout.
    The adjoint part is the same MFMA product as mh_resconv_fwd(transpose=1, accumulate=1): bit-identical; the tap gradient against
    conv2d autograd (f32 on the bf16-rounded operands) and against mh_resconv_wgrad; dw is ACCUMULATED (a non-zero start)."""
    gen = g(7 * n_p + 1)
    B, h, dh, taps = 2, 8, 64, 33
    D = h * dh
    bf = torch.bfloat16
    qkv = torch.randn(B, n_p, 3 * D, generator=gen).to(bf)
    w = (torch.randn(h, 1, taps, 1, generator=gen) * 0.2)
    dout = torch.randn(B, n_p, D, generator=gen).to(bf)
    base = torch.randn(B, n_p, 3 * D, generator=gen).to(bf)             # what the attention kernels left in d qkv
    dw0 = torch.randn(h, taps, generator=gen)
    qkv_d, dout_d, w_d = qkv.to(DEV), dout.to(DEV), w.to(DEV)
    # the two passes
    dq2 = base.to(DEV).clone()
    K.resconv(dout_d, w_d, dq2[..., 2 * D:], h, transpose=True, accumulate=True)
    dw2 = dw0.to(DEV).clone()
    K.resconv_wgrad(qkv_d[..., 2 * D:], dout_d, dw2, h)
    # one pass
    dq1 = base.to(DEV).clone()
    dw1 = dw0.to(DEV).clone()
    K.resconv_bwd(dout_d, qkv_d[..., 2 * D:], w_d, dq1[..., 2 * D:], dw1, h)
    assert torch.equal(dq1[..., :2 * D], base.to(DEV)[..., :2 * D]), "columns outside the v block were touched"
    assert torch.equal(dq1[..., 2 * D:], dq2[..., 2 * D:]), "adjoint conv differs from mh_resconv_fwd(transpose=1, accumulate=1)"
    # the tap gradient against autograd
    v = qkv[..., 2 * D:].float().reshape(B, n_p, h, dh).permute(0, 2, 1, 3).contiguous()
    wr = w.to(bf).float().clone().requires_grad_(True)
    from oracle import mirror_oracle as O
    with O.exact_cpu_convs():
        F.conv2d(v, wr, padding=(taps // 2, 0), groups=h).backward(dout.float().reshape(B, n_p, h, dh).permute(0, 2, 1, 3))
    gref = wr.grad.reshape(h, taps)
    close(dw1 - dw0.to(DEV), gref, 2e-3, 2e-3 * float(gref.abs().max()), "tap gradient (one pass)")
    close(dw1, dw2.cpu(), 1e-4, 1e-4 * float(gref.abs().max()), "tap gradient vs mh_resconv_wgrad")


def test_resconv_bwd_other_dtypes_fall_back_to_the_two_passes():
    """f32 tensors (the parity policy) take the composed passes behind the same entry point."""
    gen = g(11)
    B, h, dh, taps, n_p = 2, 4, 32, 33, 70
    D = h * dh
    qkv = torch.randn(B, n_p, 3 * D, generator=gen)
    w = torch.randn(h, 1, taps, 1, generator=gen) * 0.2
    dout = torch.randn(B, n_p, D, generator=gen)
    v = qkv[..., 2 * D:].reshape(B, n_p, h, dh).permute(0, 2, 1, 3).contiguous().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    from oracle import mirror_oracle as O
    with O.exact_cpu_convs():
        F.conv2d(v, wr, padding=(taps // 2, 0), groups=h).backward(dout.reshape(B, n_p, h, dh).permute(0, 2, 1, 3))
    dq = torch.zeros(B, n_p, 3 * D, device=DEV)
    dw = torch.zeros(h, taps, device=DEV)
    K.resconv_bwd(dout.to(DEV), qkv.to(DEV)[..., 2 * D:], w.to(DEV), dq[..., 2 * D:], dw, h)
    close(dq[..., 2 * D:], v.grad.permute(0, 2, 1, 3).reshape(B, n_p, D), 1e-5, 1e-5, "adjoint (f32)")
    close(dw, wr.grad.reshape(h, taps), 1e-4, 1e-3, "tap gradient (f32)")


def test_pinv_init_and_adjoint():
    gen = g(2)
    BH, m = 6, 40
    x = (torch.randn(BH, m, m, generator=gen)).softmax(-1)
    xd = x.to(DEV)
    st = K.pinv_absmax(xd)
    z0 = K.pinv_z0(xd, st)
    xr = x.clone().requires_grad_(True)
    ax = xr.abs()
    ref = xr.transpose(-1, -2) / (ax.sum(-1).max() * ax.sum(-2).max())
    close(z0, ref, 1e-6, 1e-7, "z0")
    dz0 = torch.randn(BH, m, m, generator=gen)
    ref.backward(dz0)
    dx = torch.zeros_like(xd)
    K.pinv_z0_bwd(xd, z0, dz0.to(DEV), st, dx)
    # the row-sum max of a softmax matrix is 1 up to rounding: its arg-max row is arbitrary, and the
    # gradient through it is annihilated by the softmax backward.  Compare after projecting both.
    def proj(gx):
        return x * (gx - (gx * x).sum(-1, keepdim=True))
    close(proj(dx.cpu()), proj(xr.grad), 1e-4, 1e-6, "z0 adjoint (after softmax bwd)")
    T = K.eye_minus(xd, 7.0)
    close(T, 7 * torch.eye(m) - x, 0, 1e-7, "eye_minus")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,add", [(20, 5), (16, 0)])
def test_seq_finish(dtype, N, add):
    gen = g(N)
    B, D = 2, 24
    n = 1 + N + add
    h = ints((B, N, D), gen)
    cls = ints((D,), gen)
    seq = torch.zeros((B, n, D), device=DEV, dtype=dtype)
    seq[:, 1:1 + N] = h.to(DEV, dtype)
    K.seq_finish(seq, cls.to(DEV), N, add)
    ref = torch.cat([cls.expand(B, 1, D), h, h[:, :add]], dim=1)
    close(seq, ref, 0, 0, "seq finish")
    d = ints((B, n, D), gen)
    dd = d.to(DEV, dtype)
    dcls = torch.zeros(D, device=DEV)
    K.seq_finish_bwd(dd, dcls, N, add)
    exp = d[:, 1:1 + N].clone()
    exp[:, :add] += d[:, 1 + N:]
    close(dd[:, 1:1 + N], exp, 0, 0, "seq finish bwd rows")
    close(dcls, d[:, 0].sum(0), 0, 0, "dcls")


@pytest.mark.parametrize("S,D,B", [(5, 32, 2), (16, 256, 2), (9, 300, 2), (40, 64, 8), (33, 32, 4), (91, 32, 2), (70, 64, 8)])
def test_ppeg(S, D, B):
    """(40, 64, 8) and (33, 32, 4): several strips per image and (batch, strip) groups a multiple of 8 — the XCD-aware workgroup
    order of ppeg_rows2_kernel (the x-tiles of a group on one XCD); the others take the plain order.  (91, 32, 2) (config 4's grid) and
    (70, 64, 8): more than 64 rows — the weight gradient walks two EQUAL strips (46 + 45, 35 + 35), the forward's rows per strip come
    from mh_ppeg_fwd's round rule."""
    gen = g(S)
    x = torch.randn(B, 1 + S * S, D, generator=gen)
    ws = [torch.randn(D, 1, k, k, generator=gen) * 0.2 for k in (7, 5, 3)]
    bs = [torch.randn(D, generator=gen) * 0.1 for _ in range(3)]
    xr = x.clone().requires_grad_(True)
    wr = [w.clone().requires_grad_(True) for w in ws]
    br = [b.clone().requires_grad_(True) for b in bs]
    grid = xr[:, 1:].transpose(1, 2).reshape(B, D, S, S)
    y = sum(F.conv2d(grid, wr[i], br[i], padding=(7 - 2 * i) // 2, groups=D) for i in range(3)) + grid
    ref = torch.cat([xr[:, :1], y.flatten(2).transpose(1, 2)], dim=1)
    merged, bsum = K.ppeg_merge(*[w.to(DEV) for w in ws], *[b.to(DEV) for b in bs])
    out = K.ppeg(x.to(DEV), merged, bsum, S, flip=False)
    close(out, ref, 1e-5, 1e-5, "ppeg fwd")
    dy = torch.randn(ref.shape, generator=gen)
    ref.backward(dy)
    dx = K.ppeg(dy.to(DEV), merged, bsum, S, flip=True)
    close(dx, xr.grad, 1e-5, 1e-5, "ppeg adjoint")
    dm, dbs = torch.zeros_like(merged), torch.zeros_like(bsum)
    K.ppeg_wgrad(x.to(DEV), dy.to(DEV), dm, dbs, S)
    dm = dm.cpu().t().reshape(D, 7, 7)
    atol = 1e-3 * max(1.0, (B * S * S) ** 0.5 / 100)      # f32 sums of B S^2 unit-variance terms: the absolute error grows like their norm
    close(dm, wr[0].grad[:, 0], 1e-4, atol, "ppeg dw7")
    close(dm[:, 1:6, 1:6], wr[1].grad[:, 0], 1e-4, atol, "ppeg dw5")
    close(dm[:, 2:5, 2:5], wr[2].grad[:, 0], 1e-4, atol, "ppeg dw3")
    close(dbs, br[0].grad, 1e-4, atol, "ppeg db")


# --------------------------------------------------------------------------------------- masking / RNA
def test_rank_mask_matches_double_argsort():
    gen = g(4)
    noise = torch.rand(3, 200, generator=gen)
    for keep in (0, 50, 199, 200):
        mask = K.rank_mask(noise.to(DEV), keep)
        rank = torch.argsort(torch.argsort(noise, dim=1), dim=1)
        assert torch.equal(mask.cpu(), (rank >= keep).float())
    # the sizes of the hot path and the largest one, with ties (stable order = lower index first) and values of both signs
    for B, N, keep in ((16, 4096, 1024), (2, 512, 384), (2, 16384, 5000), (3, 1000, 1), (1, 1, 0), (2, 3, 2)):
        noise = (torch.randint(-50, 50, (B, N), generator=gen).float() / 64 if N >= 1000 else torch.rand(B, N, generator=gen))
        mask = K.rank_mask(noise.to(DEV), keep)
        rank = torch.argsort(torch.argsort(noise, dim=1, stable=True), dim=1, stable=True)
        assert torch.equal(mask.cpu(), (rank >= keep).float()), (B, N, keep)
        assert int(mask.sum()) == B * (N - keep)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mask_apply(dtype):
    gen = g(6)
    B, T, D = 3, 9, 20
    x = ints((B, T, D), gen)
    mask = (torch.rand(B, T - 1, generator=gen) > 0.5).float()
    tok, pos = ints((D,), gen), ints((T, D), gen)
    xd = x.to(DEV, dtype)
    K.mask_apply_fwd(xd, mask.to(DEV), tok.to(DEV), pos.to(DEV), B, T, D, 1, False)
    ref = x.clone()
    ref[:, 1:] = torch.where(mask[..., None] > 0, tok.expand(B, T - 1, D), x[:, 1:])
    close(xd, ref + pos, 0, 0, "mask fwd")
    dy = ints((B, T, D), gen)
    dyd = dy.to(DEV, dtype)
    dtok, dpos = torch.zeros(D, device=DEV), torch.zeros((T, D), device=DEV)
    K.mask_apply_bwd(dyd, mask.to(DEV), dtok, dpos, B, T, D, 1, False)
    keep = torch.cat([torch.ones(B, 1), 1 - mask], dim=1)[..., None]
    close(dyd, dy * keep, 0, 0, "mask dx")
    close(dtok, (dy[:, 1:] * mask[..., None]).sum((0, 1)), 0, 0, "dtoken")
    close(dpos, dy.sum(0), 0, 0, "dpos")
    if dtype == torch.float32:      # 16-byte path (D % 4 == 0, D >= 256), ragged band and a batch that is not a multiple of 8
        Bv, Tv, Dv = 11, 37, 512
        dyv = ints((Bv, Tv, Dv), gen)
        mv = (torch.rand(Bv, Tv - 1, generator=gen) > 0.5).float()
        dyvd = dyv.to(DEV)
        dtv, dpv = torch.zeros(Dv, device=DEV), torch.zeros((Tv, Dv), device=DEV)
        K.mask_apply_bwd(dyvd, mv.to(DEV), dtv, dpv, Bv, Tv, Dv, 1, False)
        keepv = torch.cat([torch.ones(Bv, 1), 1 - mv], dim=1)[..., None]
        close(dyvd, dyv * keepv, 0, 0, "mask dx (vec)")
        close(dtv, (dyv[:, 1:] * mv[..., None]).sum((0, 1)), 0, 0, "dtoken (vec)")
        close(dpv, dyv.sum(0), 0, 0, "dpos (vec)")
    # RNA form: channels masked with a scalar token -> B x (T=D) x 1
    xr = ints((B, D), gen)
    mr = (torch.rand(B, D, generator=gen) > 0.4).float()
    t1, p1 = torch.tensor([2.0]), ints((D,), gen)
    xrd = xr.to(DEV, dtype)
    K.mask_apply_fwd(xrd, mr.to(DEV), t1.to(DEV), p1.to(DEV), B, D, 1, 0, True)
    close(xrd, torch.where(mr > 0, t1.expand(B, D), xr) + p1, 0, 0, "rna mask fwd")
    dy = ints((B, D), gen)
    dyd = dy.to(DEV, dtype)
    dt1, dp1 = torch.zeros(1, device=DEV), torch.zeros(D, device=DEV)
    K.mask_apply_bwd(dyd, mr.to(DEV), dt1, dp1, B, D, 1, 0, True)
    close(dyd, dy * (1 - mr), 0, 0, "rna mask dx")
    close(dt1, (dy * mr).sum().reshape(1), 0, 0, "rna dtoken")
    close(dp1, dy.sum(0), 0, 0, "rna dpos")


def test_mask_apply_mixed_dtypes():
    """bf16 activations in -> f32 residual stream out, and an f32 gradient back to bf16 (the WSI retention decoder)."""
    gen = g(61)
    B, T, D = 5, 19, 512
    x = ints((B, T, D), gen)
    mask = (torch.rand(B, T - 1, generator=gen) > 0.5).float()
    tok, pos = ints((D,), gen), ints((T, D), gen)
    y = torch.empty(B, T, D, device=DEV)
    K.mask_apply_fwd(x.to(DEV, torch.bfloat16), mask.to(DEV), tok.to(DEV), pos.to(DEV), B, T, D, 1, False, out=y)
    ref = x.clone()
    ref[:, 1:] = torch.where(mask[..., None] > 0, tok.expand(B, T - 1, D), x[:, 1:])
    close(y, ref + pos, 0, 0, "mask fwd bf16 -> f32")
    dy = ints((B, T, D), gen)
    dx = torch.empty(B, T, D, device=DEV, dtype=torch.bfloat16)
    dtok, dpos = torch.zeros(D, device=DEV), torch.zeros((T, D), device=DEV)
    K.mask_apply_bwd(dy.to(DEV), mask.to(DEV), dtok, dpos, B, T, D, 1, False, out=dx)
    keep = torch.cat([torch.ones(B, 1), 1 - mask], dim=1)[..., None]
    close(dx, dy * keep, 0, 0, "mask dx f32 -> bf16")
    close(dtok, (dy[:, 1:] * mask[..., None]).sum((0, 1)), 0, 0, "dtoken")
    close(dpos, dy.sum(0), 0, 0, "dpos")
    # round 5: the same pass leaves the bias gradient of the projection in front (column sums of dx), accumulating into dbias
    dx2 = torch.empty(B, T, D, device=DEV, dtype=torch.bfloat16)
    dtok2, dpos2 = torch.zeros(D, device=DEV), torch.zeros((T, D), device=DEV)
    dbias = torch.full((D,), 3.0, device=DEV)
    assert K.mask_apply_bwd_dbias_ok(dy.to(DEV), dx2, dpos2, D)
    K.mask_apply_bwd(dy.to(DEV), mask.to(DEV), dtok2, dpos2, B, T, D, 1, False, out=dx2, dbias=dbias)
    close(dx2, dy * keep, 0, 0, "mask dx with dbias")
    close(dtok2, dtok.cpu(), 0, 0, "dtoken with dbias")
    close(dpos2, dpos.cpu(), 0, 0, "dpos with dbias")
    close(dbias, 3.0 + (dy * keep).sum((0, 1)), 0, 0, "dbias = colsum(dx)")      # small integers: exact in any order
    with pytest.raises(K.MirrorHipError):       # the forms without quads cannot carry it: loud, not silently dropped
        K.mask_apply_bwd(dy[:, :, :20].contiguous().to(DEV), mask.to(DEV), dtok2[:20].contiguous(), torch.zeros((T, 20), device=DEV), B, T, 20, 1, False,
                         dbias=torch.zeros(20, device=DEV))


@pytest.mark.parametrize("D,rows_b", [(1024, 300), (512, 37), (768, 129), (256, 64)])
def test_masked_mse_bwd_leaves_column_sums(D, rows_b):
    """mh_mse_masked_bwd(colsum_ws): the per-block column sums of dpred fold to exactly what mh_colsum over the stored bf16 dpred
    gives (the bias gradient of retention_head, models/mirror.py:698-699), and dpred / dtgt are those of the plain launch."""
    gen = g(63)
    B, N = 3, rows_b
    pred = torch.randn(B, N, D, generator=gen).to(DEV, torch.bfloat16)
    E = torch.randn(B, N + 1, D, generator=gen).to(DEV)
    mask = (torch.rand(B, N, generator=gen) > 0.4).float().to(DEV)
    acc = torch.zeros(2, device=DEV)
    K.mse_masked_fwd(pred, E[:, 1:], mask, acc, B * N, D)
    gup = torch.full((1,), 0.7, device=DEV)
    dp0, dt0 = torch.empty_like(pred), torch.empty(B, N, D, device=DEV)
    K.mse_masked_bwd(pred, E[:, 1:], mask, acc, gup, dp0, dt0, B * N, D, gmul=0.15)
    dp1, dt1 = torch.empty_like(pred), torch.empty(B, N, D, device=DEV)
    assert K.mse_masked_bwd_colsum_ok(pred, E[:, 1:], dp1, D)
    ws = torch.full((K.MSE_CS_BLOCKS, D), float("nan"), device=DEV)       # every row must be written
    K.mse_masked_bwd(pred, E[:, 1:], mask, acc, gup, dp1, dt1, B * N, D, gmul=0.15, colsum_ws=ws)
    assert torch.equal(dp0, dp1)
    close(dt1, -dp1.float().cpu(), 0, 0, "dtgt = -dpred (as stored)")
    db = torch.zeros(D, device=DEV)
    K.colsum(ws, db)
    ref = dp1.double().sum((0, 1))
    close(db, ref.float().cpu(), 1e-5, 1e-6 * float(dp1.float().abs().max()) * B * N, "folded column sums")
    with pytest.raises(K.MirrorHipError):
        K.mse_masked_bwd(pred.float(), E[:, 1:], mask, acc, gup, dp0.float(), None, B * N, D, colsum_ws=ws)


@pytest.mark.parametrize("D", [512, 24])
def test_masked_mse_row_window_target(D):
    """The WSI retention target is encoder_output[:, 1:], read in place; dtgt may be left unmaterialised."""
    gen = g(62)
    B, N = 3, 37
    pred = torch.randn(B, N, D, generator=gen)
    E = torch.randn(B, N + 1, D, generator=gen)
    mask = (torch.rand(B, N, generator=gen) > 0.4).float()
    pr, er = pred.clone().requires_grad_(True), E.clone().requires_grad_(True)
    ref = (((pr - er[:, 1:]) ** 2).mean(-1) * mask).sum() / mask.sum()
    ref.backward()
    Ed = E.to(DEV)
    acc = torch.zeros(2, device=DEV)
    K.mse_masked_fwd(pred.to(DEV), Ed[:, 1:], mask.to(DEV), acc, B * N, D)
    close(acc[0] / acc[1], ref, 1e-5, 1e-6, "masked mse (window)")
    dp, dtg = torch.empty(B, N, D, device=DEV), torch.empty(B, N, D, device=DEV)
    K.mse_masked_bwd(pred.to(DEV), Ed[:, 1:], mask.to(DEV), acc, torch.ones(1, device=DEV), dp, dtg, B * N, D)
    close(dp, pr.grad, 1e-5, 1e-7, "mse dpred (window)")
    close(dtg, er.grad[:, 1:], 1e-5, 1e-7, "mse dtgt (window)")
    # bf16 prediction / bf16 dpred, no dtgt
    pb = pred.to(DEV, torch.bfloat16)
    acc2 = torch.zeros(2, device=DEV)
    K.mse_masked_fwd(pb, Ed[:, 1:], mask.to(DEV), acc2, B * N, D)
    prb = pb.float().cpu().requires_grad_(True)
    refb = (((prb - E[:, 1:]) ** 2).mean(-1) * mask).sum() / mask.sum()
    refb.backward()
    close(acc2[0] / acc2[1], refb, 1e-5, 1e-6, "masked mse (bf16 pred)")
    dpb = torch.empty(B, N, D, device=DEV, dtype=torch.bfloat16)
    K.mse_masked_bwd(pb, Ed[:, 1:], mask.to(DEV), acc2, torch.ones(1, device=DEV), dpb, None, B * N, D)
    close(dpb, prb.grad.bfloat16().float(), 1e-2, 1e-6, "mse dpred (bf16)")


@pytest.mark.parametrize("D", [512, 20])
def test_fanout_bwd(D):
    gen = g(63)
    B, T = 3, 11
    gf, c = torch.randn(B, T, D, generator=gen), torch.randn(B, D, generator=gen)
    x = torch.randn(B, T - 1, D, generator=gen).bfloat16()
    ref = gf.clone()
    ref[:, 1:] += -1.0 * x.float()
    ref[:, 0] += c
    close(K.fanout_bwd(gf.to(DEV), x.to(DEV), -1.0, c.to(DEV), B, T, D), ref, 1e-6, 1e-6, "fanout all")
    ref2 = torch.zeros(B, T, D)
    ref2[:, 1:] = 0.5 * x.float()
    close(K.fanout_bwd(None, x.to(DEV), 0.5, None, B, T, D), ref2, 1e-6, 1e-6, "fanout x only")
    close(K.fanout_bwd(gf.to(DEV), None, 0.0, None, B, T, D), gf, 0, 0, "fanout gfull only")


def test_enc_fanout_autograd():
    """EncFanoutFn + masked_mse hand-over == plain autograd over the three consumers of the encoder output."""
    from mirror_amd import functional as Fn
    gen = g(64)
    B, T, D = 2, 9, 64
    E = torch.randn(B, T, D, generator=gen)
    pred = torch.randn(B, T - 1, D, generator=gen)
    mask = (torch.rand(B, T - 1, generator=gen) > 0.4).float()
    wf, wc = torch.randn(B, T, D, generator=gen), torch.randn(B, D, generator=gen)
    er, pr = E.clone().requires_grad_(True), pred.clone().requires_grad_(True)
    ref = (((pr - er[:, 1:]) ** 2).mean(-1) * mask).sum() / mask.sum() + (er * wf).sum() + (er[:, 0] * wc).sum()
    ref.backward()
    Ed, pd = E.to(DEV).requires_grad_(True), pred.to(DEV).requires_grad_(True)
    full, tgt, cls = Fn.enc_fanout(Ed * 1.0)
    loss = Fn.masked_mse(pd, tgt, mask.to(DEV), D) + (full * wf.to(DEV)).sum() + (cls * wc.to(DEV)).sum()
    loss.backward()
    close(loss, ref, 1e-5, 1e-6, "fan-out loss")
    close(Ed.grad, er.grad, 1e-5, 1e-6, "fan-out dE")
    close(pd.grad, pr.grad, 1e-5, 1e-7, "fan-out dpred")
    # a consumer that does not know the protocol still gets summed correctly
    Ed2 = E.to(DEV).requires_grad_(True)
    full, tgt, cls = Fn.enc_fanout(Ed2 * 1.0)
    ((tgt * 2.0).sum() + full.sum()).backward()
    ref2 = torch.ones(B, T, D)
    ref2[:, 1:] += 2.0
    close(Ed2.grad, ref2, 0, 0, "fan-out generic consumer")


@pytest.mark.parametrize("H,hd", [(8, 4), (12, 8), (8, 64)])
def test_headattn(H, hd):
    gen = g(H * hd)
    B, D = 5, H * hd
    qkv = torch.randn(B, 3 * D, generator=gen)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = qr.reshape(B, 3, H, hd).unbind(1)
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, D)
    out, attn = K.headattn_fwd(qkv.to(DEV), H)
    close(out, ref, 1e-5, 1e-5, "headattn fwd")
    dy = torch.randn(B, D, generator=gen)
    ref.backward(dy)
    dq = K.headattn_bwd(qkv.to(DEV), attn, dy.to(DEV), H)
    close(dq, qr.grad, 1e-4, 1e-5, "headattn bwd")


# --------------------------------------------------------------------------------------- elementwise
def test_elementwise_family():
    gen = g(8)
    x = torch.randn(1000, generator=gen) * 2
    y = torch.randn(1000, generator=gen)
    close(K.add(x.to(DEV), y.to(DEV).bfloat16(), out_dtype=torch.float32), x + y.bfloat16().float(), 0, 1e-7, "add")
    close(K.cast(x.to(DEV), torch.bfloat16), x.bfloat16().float(), 0, 0, "cast")
    close(K.gelu_fwd(x.to(DEV)), F.gelu(x), 1e-6, 1e-6, "gelu")
    xr = x.clone().requires_grad_(True)
    F.gelu(xr).backward(y)
    close(K.gelu_bwd(x.to(DEV), y.to(DEV)), xr.grad, 1e-5, 1e-6, "gelu bwd")
    r = F.relu(x)
    close(K.relu_bwd(r.to(DEV), y.to(DEV)), y * (r > 0), 0, 0, "relu bwd")
    r3 = F.relu(torch.randn(3, 7, 8, generator=gen))
    big = torch.zeros(3, 9, 8)
    big[:, 1:8] = r3
    d3 = torch.randn(3, 7, 8, generator=gen)
    close(K.relu_bwd(big.to(DEV)[:, 1:8], d3.to(DEV), out_dtype=torch.bfloat16), (d3 * (r3 > 0)).bfloat16().float(), 0, 0, "relu bwd strided")
    m = torch.randn(300, 70, generator=gen)
    out = torch.ones(70, device=DEV)
    K.colsum(m.to(DEV)[:, :], out)
    close(out, 1 + m.sum(0), 1e-5, 1e-4, "colsum")
    mu, ls, eps = (torch.randn(4, 16, generator=gen) for _ in range(3))
    mr, lr = mu.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    zr = mr + eps * torch.exp(0.5 * lr)
    close(K.reparam_fwd(mu.to(DEV), ls.to(DEV), eps.to(DEV)), zr, 1e-6, 1e-6, "reparam")
    dz = torch.randn(4, 16, generator=gen)
    zr.backward(dz)
    dmu, dls = K.reparam_bwd(ls.to(DEV), eps.to(DEV), dz.to(DEV))
    close(dmu, mr.grad, 0, 0, "reparam dmu")
    close(dls, lr.grad, 1e-6, 1e-6, "reparam dls")
    # the gradients that reach mu / logstd from their other consumer (the KL term) are summed inside the same launch
    am, al = torch.randn(4, 16, generator=gen), torch.randn(4, 16, generator=gen)
    dmu2, dls2 = K.reparam_bwd(ls.to(DEV), eps.to(DEV), dz.to(DEV), am.to(DEV), al.to(DEV))
    close(dmu2, mr.grad + am, 1e-6, 1e-6, "reparam dmu + addend")
    close(dls2, lr.grad + al, 1e-6, 1e-6, "reparam dls + addend")
    from mirror_amd import functional as Fn
    mg, lg = mu.to(DEV).requires_grad_(True), ls.to(DEV).requires_grad_(True)
    z, mo, lo = Fn.ReparamFn.apply(mg, lg, eps.to(DEV))
    ((z * dz.to(DEV)).sum() + (mo * am.to(DEV)).sum() + (lo * al.to(DEV)).sum()).backward()
    close(mg.grad, mr.grad + am, 1e-6, 1e-6, "ReparamFn: mu through both outputs")
    close(lg.grad, lr.grad + al, 1e-6, 1e-6, "ReparamFn: logstd through both outputs")
    mg2 = mu.to(DEV).requires_grad_(True)
    _, mo2, _ = Fn.ReparamFn.apply(mg2, ls.to(DEV), eps.to(DEV))
    (mo2 * am.to(DEV)).sum().backward()          # z unused: the pass-through gradient alone
    close(mg2.grad, am, 0, 0, "ReparamFn: pass-through only")


def test_exp_fn_and_linear_pair_fn_match_torch():
    """ExpFn (`logit_scale.exp()`) and LinearPairFn (style_mu / style_logstd on one hidden vector, models/mirror.py:845-857) against
    torch autograd on the same numbers."""
    from mirror_amd import functional as Fn
    from mirror_amd.functional import POLICIES
    gen = g(29)
    s0 = torch.tensor(2.6593)
    sd = s0.to(DEV).requires_grad_(True)
    y = Fn.exp(sd)
    close(y, s0.exp(), 1e-6, 0, "exp fwd")
    (y * 3.0).backward()
    close(sd.grad, 3.0 * s0.exp(), 1e-6, 0, "exp bwd")
    prec = POLICIES["bf16"]
    M, Kd, N = 16, 256, 128
    x = ints((M, Kd), gen)
    w1, w2 = ints((N, Kd), gen), ints((N, Kd), gen)
    b1, b2 = ints((N,), gen), ints((N,), gen)
    g1, g2 = ints((M, N), gen), ints((M, N), gen)
    leaves = [t.to(DEV).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    xd = leaves[0].bfloat16()
    y1, y2 = Fn.linear_pair(xd, leaves[1], leaves[2], leaves[3], leaves[4], prec=prec, out_dtype=torch.float32)
    assert y1.grad_fn is y2.grad_fn and "LinearPair" in type(y1.grad_fn).__name__      # ONE node: x has a single consumer
    close(y1, x.double() @ w1.double().t() + b1.double(), 0, 0, "pair y1")
    close(y2, x.double() @ w2.double().t() + b2.double(), 0, 0, "pair y2")
    ((y1 * g1.to(DEV)).sum() + (y2 * g2.to(DEV)).sum()).backward()
    close(leaves[0].grad, (g1.double() @ w1.double() + g2.double() @ w2.double()).float().bfloat16().double(), 0, 0, "pair dx")
    close(leaves[1].grad, g1.double().t() @ x.double(), 0, 0, "pair dw1")
    close(leaves[3].grad, g2.double().t() @ x.double(), 0, 0, "pair dw2")
    close(leaves[2].grad, g1.double().sum(0), 0, 0, "pair db1")
    close(leaves[4].grad, g2.double().sum(0), 0, 0, "pair db2")
    # only one of the two outputs is used: the other's gradients stay undefined, x gets the used one's alone
    leaves = [t.to(DEV).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    y1, y2 = Fn.linear_pair(leaves[0].bfloat16(), leaves[1], leaves[2], leaves[3], leaves[4], prec=prec, out_dtype=torch.float32)
    (y2 * g2.to(DEV)).sum().backward()
    close(leaves[0].grad, (g2.double() @ w2.double()).float().bfloat16().double(), 0, 0, "pair dx (second only)")
    assert leaves[1].grad is None and leaves[2].grad is None
    close(leaves[3].grad, g2.double().t() @ x.double(), 0, 0, "pair dw2 (second only)")


def test_dropout_is_deterministic_and_unbiased():
    x = torch.ones(1 << 20, device=DEV)
    a = K.dropout(x, 0.1, seed=123, offset=0)
    b = K.dropout(x, 0.1, seed=123, offset=0)
    c = K.dropout(x, 0.1, seed=124, offset=0)
    assert torch.equal(a, b) and not torch.equal(a, c)
    keep = float((a > 0).float().mean())
    assert abs(keep - 0.9) < 2e-3
    assert abs(float(a.mean()) - 1.0) < 3e-3
    assert float(a.max()) == pytest.approx(1 / 0.9, rel=1e-6)
    # offset shifts the stream: elements [4:] of offset 0 == elements [:-4] of offset 4
    d = K.dropout(x, 0.1, seed=123, offset=4)
    assert torch.equal(a[4:], d[:-4])
    assert torch.equal(K.dropout(x, 0.0, seed=1, offset=0), x)


# --------------------------------------------------------------------------------------- losses / step glue
def test_ce_rows_clip_loss():
    gen = g(10)
    B, D = 8, 32
    w, r = torch.randn(B, D, generator=gen), torch.randn(B, D, generator=gen)
    s = torch.tensor(14.28)
    wr, rr, sr = (t.clone().requires_grad_(True) for t in (w, r, s))
    lab = torch.arange(B)
    ref = 0.5 * (F.cross_entropy(sr * wr @ rr.T, lab) + F.cross_entropy(sr * rr @ wr.T, lab))
    ref.backward()
    G = K.gemm(w.to(DEV), r.to(DEV).t(), mma=MH_F32)
    out = torch.zeros(1, device=DEV)
    sd = s.to(DEV).reshape(1)
    lse1 = K.ce_rows_fwd(G, sd, 1.0, 0, 0.5 / B, out)
    lse2 = K.ce_rows_fwd(G.t().contiguous(), sd, 1.0, 0, 0.5 / B, out)
    close(out, ref.reshape(1), 1e-5, 1e-5, "clip loss")
    gone = torch.ones(1, device=DEV)
    ds = torch.zeros(1, device=DEV)
    dG1 = K.ce_rows_bwd(G, sd, 1.0, lse1, gone, False, 0.5 / B, ds, 0)
    dG2 = K.ce_rows_bwd(G.t().contiguous(), sd, 1.0, lse2, gone, False, 0.5 / B, ds, 0)
    dG = dG1 + dG2.t()
    close(dG.cpu() @ r, wr.grad, 1e-4, 1e-5, "dW")
    close(dG.cpu().t() @ w, rr.grad, 1e-4, 1e-5, "dR")
    close(ds, sr.grad.reshape(1), 1e-4, 1e-5, "dscale")


def test_retention_style_cluster_losses():
    gen = g(12)
    B, N, D, P = 4, 10, 24, 300
    pred, tgt = torch.randn(B, N, D, generator=gen), torch.randn(B, N, D, generator=gen)
    mask = (torch.rand(B, N, generator=gen) > 0.4).float()
    pr, tr = pred.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
    ref = (((pr - tr) ** 2).mean(-1) * mask).sum() / mask.sum()
    ref.backward()
    acc = torch.zeros(2, device=DEV)
    K.mse_masked_fwd(pred.to(DEV), tgt.to(DEV), mask.to(DEV), acc, B * N, D)
    close(acc[0] / acc[1], ref, 1e-5, 1e-6, "masked mse")
    dp, dtg = torch.empty(B, N, D, device=DEV), torch.empty(B, N, D, device=DEV)
    K.mse_masked_bwd(pred.to(DEV), tgt.to(DEV), mask.to(DEV), acc, torch.ones(1, device=DEV), dp, dtg, B * N, D)
    close(dp, pr.grad, 1e-5, 1e-7, "mse dpred")
    close(dtg, tr.grad, 1e-5, 1e-7, "mse dtgt")
    mu, ls = torch.randn(B, 16, generator=gen), torch.randn(B, 16, generator=gen) * 0.5
    mr, lr = mu.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    ref = 0.5 * (lr.exp() + mr ** 2 - 1 - lr).sum(1).mean()
    ref.backward()
    out = torch.zeros(1, device=DEV)
    K.kl_fwd(mu.to(DEV), ls.to(DEV), out, 0.5 / B)
    close(out, ref.reshape(1), 1e-5, 1e-6, "style kl")
    dmu, dls = K.kl_bwd(mu.to(DEV), ls.to(DEV), torch.ones(1, device=DEV), 0.5 / B)
    close(dmu, mr.grad, 1e-5, 1e-7, "kl dmu")
    close(dls, lr.grad, 1e-5, 1e-7, "kl dls")
    w, r = torch.randn(B, P, generator=gen) * 2, torch.randn(B, P, generator=gen) * 2
    wr, rr = w.clone().requires_grad_(True), r.clone().requires_grad_(True)
    pw, prr = wr.softmax(-1), rr.softmax(-1)
    ref = 0.5 * (F.kl_div(pw.log(), prr, reduction="batchmean") + F.kl_div(prr.log(), pw, reduction="batchmean"))
    ref.backward()
    out = torch.zeros(1, device=DEV)
    K.symkl_fwd(w.to(DEV), r.to(DEV), out, 0.5 / B)
    close(out, ref.reshape(1), 1e-5, 1e-6, "cluster loss")
    dw, dr = K.symkl_bwd(w.to(DEV), r.to(DEV), torch.ones(1, device=DEV), 0.5 / B)
    close(dw, wr.grad, 1e-4, 1e-7, "symkl dw")
    close(dr, rr.grad, 1e-4, 1e-7, "symkl dr")


def test_step_glue_rownorm_clamp_adam():
    gen = g(13)
    w = torch.randn(300, 40, generator=gen)
    wd = w.to(DEV)
    K.rownorm_(wd)
    close(wd, F.normalize(w, dim=1), 1e-6, 1e-7, "rownorm")
    wd2, sh2 = w.to(DEV), torch.zeros(300, 40, device=DEV, dtype=torch.bfloat16)
    K.rownorm_(wd2, shadow=sh2)            # the bf16 copy the GEMMs read, written by the same launch
    assert torch.equal(wd2, wd) and torch.equal(sh2, wd.bfloat16())
    s = torch.tensor([5.3], device=DEV)
    K.clamp_(s, 0.0, math.log(100))
    assert float(s) == pytest.approx(math.log(100))
    n = 10007
    p, gr = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=2e-5)
    pd, m, v = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sh = torch.zeros(n, device=DEV, dtype=torch.bfloat16)
    for step in range(1, 4):
        gstep = gr * step
        pr.grad = gstep.clone()
        opt.step()
        K.adam(pd, gstep.to(DEV), m, v, sh, 2e-5, 0.9, 0.999, 1e-8, 1 - 0.9 ** step, 1 - 0.999 ** step)
    close(pd, pr.detach(), 1e-6, 1e-7, "adam")
    close(sh, pr.detach().bfloat16().float(), 1e-2, 1e-6, "adam bf16 shadow")
    # step glue riding on the Adam launches: one parameter clamped behind its update (logit_scale), a device counter advanced
    for idx in (5, n - 2):                 # inside the 4-wide body / in the scalar tail (n % 4 == 3)
        p2, m2, v2 = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        p3, m3, v3 = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        sh3 = torch.zeros(n, device=DEV, dtype=torch.bfloat16)
        cnt = torch.tensor([40], device=DEV, dtype=torch.int64)
        K.adam(p2, gr.to(DEV), m2, v2, None, 2e-5, 0.9, 0.999, 1e-8, 0.1, 0.001)
        lo, hi = float(p2[idx]) + 0.25, float(p2[idx]) + 0.5
        K.adam(p3, gr.to(DEV), m3, v3, sh3, 2e-5, 0.9, 0.999, 1e-8, 0.1, 0.001, clamp=(idx, lo, hi), counter=cnt, counter_add=7)
        want = p2.clone()
        want[idx] = lo
        assert torch.equal(p3, want) and torch.equal(sh3, want.bfloat16()) and int(cnt) == 47
    from mirror_amd import MirrorHipError
    with pytest.raises(MirrorHipError):
        K.adam(p2, gr.to(DEV), m2, v2, None, 2e-5, 0.9, 0.999, 1e-8, 0.1, 0.001, clamp=(n, 0.0, 1.0))
    # one optimizer step as two launches (round 5): a range updated first (and the device step state advanced by that launch), the
    # rest later with tick=False and a hole — bit-identical to the single launch, the counter advanced once, the clamp honoured
    for lo_, hi_ in ((1024, 7000), (0, 4096), (4096, 10004)):
        outs = []
        for split in (False, True):
            pp, mm, vv = p.to(DEV), torch.full((n,), 0.01, device=DEV), torch.full((n,), 0.02, device=DEV)
            shh = torch.zeros(n, device=DEV, dtype=torch.bfloat16)
            st = torch.tensor([2.0, 0.0, 0.0, 3e-4, 1.0, 0.0], device=DEV)
            cnt = torch.tensor([8], device=DEV, dtype=torch.int64)
            gd = gr.to(DEV)
            kw = dict(dev_state=st, clamp=(5 if lo_ else n - 2, -0.1, 0.1), counter=cnt, counter_add=16)
            if split:
                K.adam(pp[lo_:hi_], gd[lo_:hi_], mm[lo_:hi_], vv[lo_:hi_], shh[lo_:hi_], 0.0, 0.9, 0.999, 1e-8, 1.0, 1.0, dev_state=st, tick=True)
                assert float(st[0]) == 3.0 and int(cnt) == 8
                K.adam(pp, gd, mm, vv, shh, 0.0, 0.9, 0.999, 1e-8, 1.0, 1.0, tick=False, hole=(lo_, hi_), **kw)
            else:
                K.adam(pp, gd, mm, vv, shh, 0.0, 0.9, 0.999, 1e-8, 1.0, 1.0, **kw)
            assert float(st[0]) == 3.0 and int(cnt) == 24
            outs.append((pp, mm, vv, shh))
        for a, b in zip(*outs):
            assert torch.equal(a, b), (lo_, hi_)
    with pytest.raises(MirrorHipError):        # the hole must be quad-aligned and must not hold the clamped parameter
        K.adam(p2, gr.to(DEV), m2, v2, None, 2e-5, 0.9, 0.999, 1e-8, 0.1, 0.001, hole=(2, 8))
    with pytest.raises(MirrorHipError):
        K.adam(p2, gr.to(DEV), m2, v2, None, 2e-5, 0.9, 0.999, 1e-8, 0.1, 0.001, clamp=(5, 0.0, 1.0), hole=(4, 8))


@pytest.mark.parametrize("M,N,Kd", [(512, 256, 64), (1024, 512, 512), (768, 1024, 1536)])
def test_gemm_four_wave_experiment_is_bit_equal_to_mh_gemm(M, N, Kd):
    """mh_gemm_w4 (round 5 experiment: the 256 x 256 x 64 tile on four waves with 128 x 128 wave tiles) computes the same bf16 products in
    the same k order as the 8-wave kernels behind mh_gemm: the results must be bit-identical, with and without a bias."""
    gen = g(M + Kd)
    a = torch.randn(M, Kd, generator=gen).to(DEV, torch.bfloat16)
    w = (torch.randn(N, Kd, generator=gen) * 0.05).to(DEV, torch.bfloat16)
    b = torch.randn(N, generator=gen).to(DEV)
    for bias in (None, b):
        ref = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        K.gemm(a, w.t(), out=ref, bias=bias, mma=MH_BF16)
        got = K.gemm_w4(a, w, bias)
        assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
    with pytest.raises(K.MirrorHipError):
        K.gemm_w4(a[:, :Kd - 32].contiguous(), w[:, :Kd - 32].contiguous())


def test_noise_draws_one_launch_uniform_and_normal():
    """mh_noise_draws: uniform [0, 1) on 24 bits and Box-Muller normals from the Philox dropout stream — a pure function of
    (seed, offset + device base, element), the moments of the distributions they stand for (torch.rand / torch.randn at
    models/mirror.py:630, :516, :832-833), and what Fn.noise_draws hands the model: four views, the host offset advanced past them."""
    from mirror_amd import functional as Fn
    nu, nn_ = 16 * 4096 + 16 * 512, 2 * 16 * 256
    a = K.noise_draws(nu, nn_, 1234, 64, None, DEV)
    b = K.noise_draws(nu, nn_, 1234, 64, None, DEV)
    assert torch.equal(a, b)
    base = torch.tensor([128], device=DEV, dtype=torch.int64)
    c = K.noise_draws(nu, nn_, 1234, 64, base, DEV)
    d = K.noise_draws(nu, nn_, 1234, 192, None, DEV)
    assert torch.equal(c, d) and not torch.equal(a, c)
    assert not torch.equal(a, K.noise_draws(nu, nn_, 1235, 64, None, DEV))
    u, z = a[:nu].double().cpu(), a[nu:].double().cpu()
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 5e-3 and abs(float(u.var()) - 1 / 12) < 2e-3
    assert bool(torch.isfinite(z).all()) and abs(float(z.mean())) < 0.05 and abs(float(z.var()) - 1.0) < 0.05
    assert abs(float((z ** 4).mean()) - 3.0) < 0.3 and 3.0 < float(z.abs().max()) < 6.5
    assert len(torch.unique(u[:4096])) > 4000          # no stuck words
    Fn.manual_seed(7)
    Fn._dropout_state["offset"] = 6
    m0, m1, e0, e1 = Fn.noise_draws(3, 10, 7, 5, DEV)
    assert m0.shape == (3, 10) and m1.shape == (3, 7) and e0.shape == (3, 5) and e1.shape == (3, 5)
    assert Fn._dropout_state["offset"] == 6            # a range of their own: the RNA branch's graph replay has its offsets baked from 0
    assert not torch.equal(e0, e1)
    Fn.noise_draws_advance()
    assert Fn._dropout_state["offset"] == 16
    Fn._dropout_state["offset"] = 0


# --------------------------------------------------------------------------------------- skinny-M linears
@pytest.mark.parametrize("M,N,Kd", [(16, 1024, 2048), (3, 3000, 512), (32, 96, 64), (20, 16, 32),
                                    (16, 1536, 10234), (16, 768, 1975), (16, 1975, 768), (5, 37, 13), (32, 50, 1001)])
def test_skinny_fwd_wgrad_transpose(M, N, Kd):
    """(.., 10234), (.., 1975), (1975, ..) — the reference template's gene count and MLP width (configs/pretrain/mirror.template.yaml:27-46)
    — and the small odd shapes run the element-wise instance of mh_skinny_fwd (K or N no multiple of 32, rows only 2-byte aligned);
    integer operands: exact."""
    gen = g(M + N)
    x, w, dy = ints((M, Kd), gen), ints((N, Kd), gen), ints((M, N), gen)
    bias = ints((N,), gen)
    xd, wd, dyd = x.to(DEV).bfloat16(), w.to(DEV).bfloat16(), dy.to(DEV).bfloat16()
    assert K.skinny_ok(xd, wd)
    y = K.skinny_fwd(xd, wd, bias.to(DEV), ACT_RELU, torch.float32)
    close(y, F.relu(x.double() @ w.double().t() + bias.double()), 0, 0, "skinny fwd")
    yb = K.skinny_fwd(xd, wd, None, ACT_NONE, torch.bfloat16)
    close(yb, (x.double() @ w.double().t()).float().bfloat16().double(), 0, 0, "skinny fwd bf16")
    wt = K.transpose_bf16(wd)
    close(wt, w.t(), 0, 0, "transpose")
    assert K.skinny_vec_ok(xd, wd) == (Kd % 32 == 0)
    if Kd % 32:      # f32 x straight into the element-wise instance, from a row-strided buffer
        xs = torch.zeros(M, Kd + 3, device=DEV)
        xs[:, :Kd] = x.to(DEV)
        close(K.skinny_fwd(xs[:, :Kd], wd, None, ACT_NONE, torch.float32), x.double() @ w.double().t(), 0, 0, "skinny fwd, f32 x, odd row stride")
    if True:
        dx = K.skinny_fwd(dyd, wt, None, ACT_NONE, torch.float32)
        close(dx, dy.double() @ w.double(), 0, 0, "skinny dgrad via W^T")
        # an f32 addend (another consumer's data gradient of the same x) is summed in front of the activation, any row stride
        add = ints((M, Kd + 8), gen).to(DEV)[:, :Kd]
        dx2 = K.skinny_fwd(dyd, wt, None, ACT_NONE, torch.float32, addend=add)
        close(dx2, dy.double() @ w.double() + add.double().cpu(), 0, 0, "skinny dgrad + addend")
        dx3 = K.skinny_fwd(dyd, wt, None, ACT_RELU, torch.bfloat16, addend=add)
        close(dx3, F.relu(dy.double() @ w.double() + add.double().cpu()).float().bfloat16().double(), 0, 0, "skinny relu(dgrad + addend) bf16")
    dw = torch.full((N, Kd), 2.0, device=DEV)
    K.skinny_wgrad(dyd, xd, dw, accumulate=True)
    close(dw, 2 + dy.double().t() @ x.double(), 0, 0, "skinny wgrad accumulate")
    K.skinny_wgrad(dyd, xd, dw, accumulate=False)
    close(dw, dy.double().t() @ x.double(), 0, 0, "skinny wgrad")
    dbv = torch.full((N,), 1.0, device=DEV)
    K.skinny_wgrad(dyd, xd, dw, accumulate=False, db=dbv)                  # bias gradient in the same launch
    close(dbv, 1 + dy.double().sum(0), 0, 0, "skinny wgrad db")
    close(dw, dy.double().t() @ x.double(), 0, 0, "skinny wgrad (with db)")
    # batched transpose table
    src = torch.cat([wd.reshape(-1), xd.reshape(-1)])
    dst = torch.zeros_like(src)
    tab = torch.tensor([0, 0, N, Kd, N * Kd, N * Kd, M, Kd], device=DEV, dtype=torch.int64)
    K.transpose_bf16_many(src, dst, tab, 2, max(N, M), Kd)
    close(dst[:N * Kd].view(Kd, N), w.t(), 0, 0, "batched transpose 0")
    close(dst[N * Kd:].view(Kd, M), x.t(), 0, 0, "batched transpose 1")


def test_transpose_many_vec():
    """16-byte path of the batched transpose (every W^T shadow after the optimizer step): ragged 64-tiles, two shapes."""
    gen = g(71)
    a = torch.randint(-100, 100, (96, 200), generator=gen).float().bfloat16()
    b = torch.randint(-100, 100, (1536, 512), generator=gen).float().bfloat16()
    src = torch.cat([a.reshape(-1), b.reshape(-1)]).to(DEV)
    dst = torch.zeros_like(src)
    tab = torch.tensor([0, 0, 96, 200, a.numel(), a.numel(), 1536, 512], device=DEV, dtype=torch.int64)
    K.transpose_bf16_many(src, dst, tab, 2, 1536, 512, vec_ok=True)
    close(dst[:a.numel()].view(200, 96), a.float().t(), 0, 0, "vec transpose 0")
    close(dst[a.numel():].view(512, 1536), b.float().t(), 0, 0, "vec transpose 1")


# --------------------------------------------------------------------------------------- fused pinv chain
@pytest.mark.parametrize("BH", [1, 3])
def test_pinv_chain_matches_reference_iteration(BH):
    """One-launch Moore-Penrose chain (bf16 operands, f32 accumulate) vs the f64 iteration and its autograd."""
    from oracle import mirror_oracle as O
    m = K.PINV_CHAIN_M
    gen = g(m + BH)
    iters = 6
    x = (torch.randn(1, BH, m, m, generator=gen) * 2).softmax(-1)
    xd = x.to(DEV)
    st = K.pinv_absmax(xd)
    saved = torch.zeros((iters, 4, BH, m, m), device=DEV, dtype=torch.bfloat16)
    z0, xb = K.pinv_chain_prep(xd, st, saved[0, 0])
    close(z0, K.pinv_z0(xd, st).cpu(), 0, 1e-12, "prep z0")

    def pn(M):      # panel-native layout of [BH, m, m] (include/mirror_hip.h)
        it = torch.arange(m * m // 8)
        lane, T, jblk = it & 63, (it >> 6) & 15, it >> 10
        e = torch.arange(8)
        i = (16 * T + 4 * (lane >> 5))[:, None] + (e & 3) + 8 * (e >> 2)
        j = (32 * jblk + (lane & 31))[:, None].expand(-1, 8)
        return M[:, i, j].reshape(M.shape[0], m, m)

    close(xb.reshape(BH, m, m), pn(x[0].to(torch.bfloat16).float()), 0, 0, "prep PN(x)")
    close(saved[0, 0], pn(z0.reshape(BH, m, m).cpu().to(torch.bfloat16).float()), 0, 0, "prep PN(z0)")
    zfT = torch.empty((BH, m, m), device=DEV, dtype=torch.bfloat16)
    K.pinv_chain_fwd(xb, saved, zfT, iters)
    zf = zfT.transpose(-1, -2)
    xr = x.double().requires_grad_(True)
    ref = O.pinv_iter(xr, iters)
    scale = float(ref.abs().max())
    err = float((zf.float().cpu().double() - ref[0].detach()).abs().max()) / scale
    assert err < 3e-2, f"chain fwd rel-to-max err {err}"
    # backward: d/dX of sum(G * z_final), G random; compare direction and size with f64 autograd
    G = torch.randn(1, BH, m, m, generator=gen).double()
    (ref * G).sum().backward()
    work = torch.empty_like(saved)
    dX = torch.empty((BH, m, m), device=DEV)
    dz0 = torch.empty((BH, m, m), device=DEV)
    K.pinv_chain_bwd(xb, saved, K.pinv_chain_pack(G[0].float().to(DEV)), work, dX, dz0, iters)
    assert bool(torch.isfinite(dX).all()) and bool(torch.isfinite(dz0).all())
    K.pinv_z0_bwd(xd.reshape(BH, m, m), z0.reshape(BH, m, m), dz0, st, dX)
    got = dX.cpu().double()
    want = xr.grad[0]
    # project both through the softmax backward (the row-sum max sub-gradient is arbitrary, see test_pinv_init_and_adjoint)
    xs = x[0].double()
    proj = lambda gx: xs * (gx - (gx * xs).sum(-1, keepdim=True))  # noqa: E731
    a, b = proj(got).flatten(), proj(want).flatten()
    cos = float((a @ b) / (a.norm() * b.norm()))
    ratio = float(a.norm() / b.norm())
    assert cos > 0.995 and 0.97 < ratio < 1.03, (cos, ratio)


# --------------------------------------------------------------------------------------- fused Nystrom attention sides
@pytest.mark.parametrize("B,h,l", [(1, 1, 1), (2, 2, 3), (1, 8, 17)])
def test_nys_fused_attention_sides(B, h, l):
    """mh_nys_attn{1,3}_{fwd,bwd} against autograd through the plain softmax(q k^T) products (f64 on CPU)."""
    gen = g(100 * B + 10 * h + l)
    m, dh = 256, 64
    D, n_p, scale = h * dh, m * l, dh ** -0.5
    bf = torch.bfloat16
    qkv = (torch.randn((B, n_p, 3 * D), generator=gen) * 1.5).to(bf)
    lm = (torch.randn((B, m, 2 * D), generator=gen) * 1.5).to(bf)
    w2 = torch.randn((B, h, m, dh), generator=gen).to(bf)
    dout = torch.randn((B, n_p, D), generator=gen).to(bf)
    dav = torch.randn((B, h, m, dh), generator=gen).to(bf)

    def heads(t, which, parts):
        return t.double().view(B, t.shape[1], parts, h, dh)[:, :, which].permute(0, 2, 1, 3)

    qkv_r, lm_r, w2_r = (t.double().requires_grad_() for t in (qkv, lm, w2))
    q, k, v = (heads(qkv_r, i, 3) for i in range(3))
    ql, kl = heads(lm_r, 0, 2), heads(lm_r, 1, 2)
    s1 = scale * q @ kl.transpose(-1, -2)
    out_ref = (torch.softmax(s1, -1) @ w2_r).permute(0, 2, 1, 3).reshape(B, n_p, D)
    s3 = scale * ql @ k.transpose(-1, -2)
    av_ref = torch.softmax(s3, -1) @ v
    ((out_ref * dout.double()).sum() + (av_ref * dav.double()).sum()).backward()

    qkv_d, lm_d, w2_d, dout_d, dav_d = (t.to(DEV) for t in (qkv, lm, w2, dout, dav))
    out = torch.full((B, n_p, D), float("nan"), device=DEV, dtype=bf)
    lse1 = K.nys_attn1_fwd(qkv_d, lm_d, w2_d, out, h, scale)
    out_acc = torch.ones((B, n_p, D), device=DEV, dtype=bf)
    o1 = torch.full((B, n_p, D), float("nan"), device=DEV, dtype=bf)
    K.nys_attn1_fwd(qkv_d, lm_d, w2_d, out_acc, h, scale, accumulate=True, o1=o1)
    close(out_acc, 1.0 + out.float().cpu(), 0.0, 2e-2 * float(out.float().abs().max()), "attn1 accumulate")
    assert torch.equal(o1, out), "o1 = attn1's own rows, without the addend"
    av, lse3 = K.nys_attn3_fwd(qkv_d, lm_d, h, scale)
    out_ref, av_ref, s1, s3 = (t.detach() for t in (out_ref, av_ref, s1, s3))
    close(out, out_ref, 0.0, 2e-2 * float(out_ref.abs().max()), "attn1 out")
    close(lse1, torch.logsumexp(s1, -1), 1e-5, 1e-4, "lse1")
    close(av, av_ref, 0.0, 1e-2 * float(av_ref.abs().max()), "attn3 av")
    close(lse3, torch.logsumexp(s3, -1), 1e-5, 1e-4, "lse3")

    dqkv = torch.full_like(qkv_d, float("nan"))
    dw2 = torch.zeros((B, h, m, dh), device=DEV)
    dlm = torch.zeros((B, m, 2 * D), device=DEV)
    # round 5: part 1 (dw2, dk_l, delta1 = sum_d dO o1) first, then part 2 (dq from delta1) — as NystromCoreFn.backward issues them
    delta1 = torch.full((B, h, n_p), float("nan"), device=DEV)
    K.nys_attn1_bwd(qkv_d, lm_d, w2_d, dout_d, lse1, o1, delta1, dqkv, dw2, dlm, h, scale, which=1)
    p1 = torch.softmax(s1, -1)
    dp1 = heads(dout, 0, 1) @ w2.double().transpose(-1, -2)
    dref = (p1 * dp1).sum(-1)
    close(delta1, dref, 0.0, 2e-2 * float(dref.abs().max()), "delta1 = sum_l P dP from the saved rows")
    K.nys_attn1_bwd(qkv_d, lm_d, w2_d, dout_d, lse1, o1, delta1, dqkv, dw2, dlm, h, scale, which=2)
    K.nys_attn3_bwd(qkv_d, lm_d, av, dav_d, lse3, dqkv, dlm, h, scale)
    for name, got, ref in (("dqkv", dqkv, qkv_r.grad), ("dw2", dw2, w2_r.grad), ("dlm", dlm, lm_r.grad)):
        close(got, ref, 0.0, 2e-2 * float(ref.abs().max()), name)
        rel = float((got.float().cpu().double() - ref).norm() / ref.norm())
        assert rel < 1e-2, (name, rel)
    # both forms of attn3's backward (round 5: one pass by default; the dk / dv + dq_l pair stays behind the switch)
    for one in (True, False):
        dq3 = torch.full_like(qkv_d, float("nan"))
        dl3 = torch.zeros_like(dlm)
        K.nys_attn3_bwd(qkv_d, lm_d, av, dav_d, lse3, dq3, dl3, h, scale, one_pass=one)
        ref_kv = qkv_r.grad[..., D:]
        close(dq3[..., D:], ref_kv, 0.0, 2e-2 * float(ref_kv.abs().max()), f"dk | dv (one_pass={one})")
        ref_ql = lm_r.grad[..., :D] - 0.0      # attn3's share of d q_l: attn1 adds nothing to the q_l half
        close(dl3[..., :D], ref_ql, 0.0, 2e-2 * float(ref_ql.abs().max()), f"dq_l (one_pass={one})")
        assert float(dl3[..., D:].abs().max()) == 0.0
    # delta3 handed in (what mh_nys_dz_dav leaves): the call skips its first launch and writes the same dk / dv
    dqkv2 = torch.full_like(qkv_d, float("nan"))
    K.nys_attn3_bwd(qkv_d, lm_d, av, dav_d, lse3, dqkv2, torch.zeros_like(dlm), h, scale, delta3=(dav_d.float().view(B, h, m, dh) * av).sum(-1))
    assert float((dqkv2[..., D:].float() - dqkv[..., D:].float()).abs().max()) <= 1e-2 * float(dqkv[..., D:].float().abs().max())


@pytest.mark.parametrize("B,h,l", [(1, 1, 1), (2, 2, 3), (1, 8, 17)])
def test_nys_attn3_fwd_with_res_conv_inside(B, h, l):
    """mh_nys_attn3_fwd(rc_w, rc_out) (round 5): the launch that stages the v tiles also writes res_conv(v) — the same Toeplitz MFMA
    product as mh_resconv_fwd on the same operands, so the rows are bit-identical; av / lse3 do not change; every row of rc_out is
    written (NaN-filled on entry), including the first / last 16 rows whose halo lies outside the sequence."""
    gen = g(1000 * B + 10 * h + l)
    m, dh, taps = 256, 64, 33
    D, n_p, scale = h * dh, m * l, dh ** -0.5
    bf = torch.bfloat16
    qkv = (torch.randn((B, n_p, 3 * D), generator=gen) * 1.5).to(bf).to(DEV)
    lm = (torch.randn((B, m, 2 * D), generator=gen) * 1.5).to(bf).to(DEV)
    w = (torch.randn(h, 1, taps, 1, generator=gen) * 0.2).to(DEV)
    av0, lse0 = K.nys_attn3_fwd(qkv, lm, h, scale)
    want = torch.empty((B, n_p, D), device=DEV, dtype=bf)
    K.resconv(qkv[..., 2 * D:], w, want, h, transpose=False, accumulate=False)
    out = torch.full((B, n_p, D), float("nan"), device=DEV, dtype=bf)
    av1, lse1 = K.nys_attn3_fwd(qkv, lm, h, scale, rc=(w.reshape(-1).contiguous(), out))
    assert torch.equal(av0, av1) and torch.equal(lse0, lse1)
    assert bool(torch.isfinite(out.float()).all()), "rows of rc_out left unwritten"
    assert torch.equal(out, want), float((out.float() - want.float()).abs().max())


def test_nys_fused_rejects_other_geometry():
    from mirror_amd._lib import MirrorHipError
    bf = torch.bfloat16
    qkv = torch.zeros((1, 128, 3 * 32), device=DEV, dtype=bf)
    assert not K.nys_fused_ok(qkv, 1, 128)
    with pytest.raises(MirrorHipError):
        K.nys_attn3_fwd(qkv, torch.zeros((1, 128, 64), device=DEV, dtype=bf), 1, 1.0)


@pytest.mark.parametrize("dtype,Fd", [(torch.bfloat16, 1024), (torch.float32, 8), (torch.bfloat16, 7)])
def test_gather_rows_and_device_feed(dtype, Fd):
    """batched `wsi_feature[sampled_indices]` (datasets/dataset_pretrain.py:162) — bit-exact copies."""
    gen = g(81)
    bank = torch.randn(300, Fd, generator=gen).to(dtype)
    rows = torch.randint(0, 300, (3, 41), generator=gen)
    out = K.gather_rows(bank.to(DEV), rows.to(DEV))
    assert out.shape == (3, 41, Fd) and torch.equal(out.cpu(), bank[rows])


def test_device_slide_bank_matches_reference_items():
    """The device feed reproduces the reference dataset's items when it is given the reference's own index draws
    (tests/golden/golden_datafeed.npz), and draws by the same replace rule on its own."""
    import numpy as np
    import os
    from mirror_amd.data import DeviceSlideBank
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_datafeed.npz"))
    N = int(z["num_tokens"])
    slides = [torch.from_numpy(z[f"slide/{k}"]) for k in range(3)]
    bank = DeviceSlideBank(slides, torch.from_numpy(z["rna"]), N, device=DEV)
    ids = [int(k) for k in z["order"]]
    rows = torch.stack([torch.from_numpy(z[f"out/{j}/idx"]).long() + int(bank.offsets[k]) for j, k in enumerate(ids)])
    wsi, rna = bank.batch(ids, rows=rows)
    for j in range(len(ids)):
        assert torch.equal(wsi[j].cpu(), torch.from_numpy(z[f"out/{j}/wsi"]))
        assert torch.equal(rna[j].cpu(), torch.from_numpy(z[f"out/{j}/rna"]))
    gen = torch.Generator(device=DEV).manual_seed(1)
    wsi2, _ = bank.batch([0, 1, 2], generator=gen)
    drawn = bank.draw([0, 1, 2], generator=gen).cpu()
    assert wsi2.shape == (3, N, slides[0].shape[1])
    for j, k in enumerate([0, 1, 2]):
        local = drawn[j] - int(bank.offsets[k])
        assert int(local.min()) >= 0 and int(local.max()) < slides[k].shape[0]
        if slides[k].shape[0] >= N:
            assert len(set(local.tolist())) == N


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_softmax_masked_and_row_scale(dtype):
    """Key-padding-mask pieces ([3P] masked_fill(-finfo.max) + softmax; zeroed rows / masked-mean scale)."""
    gen = g(91)
    B, h, R, Cc = 2, 3, 5, 300
    x = torch.randn(B, h, R, Cc, generator=gen)
    rm = (torch.rand(B, R, generator=gen) > 0.3).float()
    cm = (torch.rand(B, Cc, generator=gen) > 0.4).float()
    rm[0, 0] = 0.0                                           # a fully masked row -> uniform
    xr = x.to(dtype).float().requires_grad_(True)
    keep = (rm[:, None, :, None] > 0) & (cm[:, None, None, :] > 0)
    ref = xr.masked_fill(~keep, -torch.finfo(torch.float32).max).softmax(-1)
    dy = torch.randn(B, h, R, Cc, generator=gen)
    ref.backward(dy)
    y = K.softmax_masked_fwd(x.to(DEV, dtype), rm.to(DEV), cm.to(DEV), out_dtype=torch.float32)
    close(y, ref.detach(), 1e-5, 1e-7, "masked softmax")
    assert abs(float(y[0, 0, 0].sum()) - 1.0) < 1e-5 and float(y[0, 0, 0].max() - y[0, 0, 0].min()) < 1e-9
    dx = K.softmax_masked_bwd(y, dy.to(DEV), rm.to(DEV), cm.to(DEV))
    close(dx, xr.grad, 1e-4, 1e-6, "masked softmax bwd")
    z = torch.randn(7, 9, 64, generator=gen).to(dtype)
    sc = torch.rand(7, 9, generator=gen)
    close(K.row_scale(z.to(DEV), sc.to(DEV)), (z.float() * sc[..., None]).to(dtype).float(), 0, 0, "row scale")


def test_dropout_add_equals_dropout_then_add():
    """Fused a + dropout(b) and the fused cast + dropout backward draw the very masks of the two-op form."""
    from mirror_amd import functional as Fn
    gen = g(95)
    a = torch.randn(3, 50, 64, generator=gen).to(DEV)
    b = torch.randn(3, 50, 64, generator=gen).to(DEV, torch.bfloat16)
    dy = torch.randn(3, 50, 64, generator=gen).to(DEV)
    Fn.manual_seed(77)
    a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y1 = Fn.add(a1, Fn.dropout(b1, 0.25, True), torch.float32)
    y1.backward(dy)
    Fn.manual_seed(77)
    a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y2 = Fn.dropout_add(a2, b2, 0.25, True)
    y2.backward(dy.clone())
    # same masks; the fused form skips two bf16 roundings (the dropped copy of b, the cast of dy), so values agree to bf16 eps
    assert torch.equal(y1 == a, y2 == a) and torch.equal(b1.grad == 0, b2.grad == 0)
    close(y2, y1.cpu(), 1e-2, 1e-2, "dropout_add forward")
    close(b2.grad, b1.grad.float().cpu(), 2e-2, 1e-3, "dropout_add db")
    assert torch.equal(a1.grad, a2.grad)
    ref = a.cpu() + torch.where(y2.cpu() != a.cpu(), b.float().cpu() / 0.75, torch.zeros(()))
    close(y2, ref, 1e-6, 1e-6, "dropout_add exact")
    kept = float((y2 != a).float().mean())
    assert 0.70 < kept < 0.80
    Fn._res_grads.clear()


@pytest.mark.parametrize("M,N,Kd,batch", [(300, 256, 128, 1), (1100, 128, 64, 1), (137, 384, 192, 3), (4100, 512, 256, 4)])
def test_fp8_quant_and_gemm(M, N, Kd, batch):
    """BASELINE config 5 pieces: per-tensor e4m3 quantisation (bit-exact against torch.float8_e4m3fn) and the fp8 MFMA product
    (f32 accumulate) with bias / ReLU, ragged M, a batch of row windows of a larger buffer.  The last shape is large enough for
    the 256-tile pipeline with the 32x32x64 f8f6f4 MFMA; the others run on the 128-tile kernel."""
    gen = g(101)
    e4 = torch.float8_e4m3fn
    x = (torch.randn(batch, M + 5, Kd, generator=gen) * 3).bfloat16()
    w = torch.randn(N, Kd, generator=gen).bfloat16()
    bias = torch.randn(N, generator=gen)
    xq, sx = K.quant_fp8(x.to(DEV))
    wq, sw = K.quant_fp8(w.to(DEV))
    for t, q, sc in ((x, xq, sx), (w, wq, sw)):
        amax = t.float().abs().max()
        qr = (t.float() * (448.0 / amax)).clamp(-448, 448).to(e4)
        assert torch.equal(q.cpu().view(e4).float(), qr.float()), "quantised bytes differ"
        close(sc, (amax / 448.0).reshape(1), 1e-7, 0, "scale")
    xr = xq.cpu().view(e4).float()[:, 2:2 + M]            # a row window, like to_out(x)[:, -n:]
    ref = torch.relu(xr.double() @ wq.cpu().view(e4).float().double().t() * float(sx) * float(sw) + bias.double())
    out = torch.full((batch, M + 1, N), -7.0, device=DEV)
    K.gemm_fp8(xq[:, 2:2 + M], sx, wq, sw, out[:, 1:], bias=bias.to(DEV), act=1)
    # products of e4m3 values are exact in f32, but the MFMA's 16-term dot product is not accumulated at full f32 precision
    # (measured: ~1e-4 relative to the row / column norms), hence the tolerance
    close(out[:, 1:], ref, 1e-3, 2e-2, "fp8 gemm")
    assert float(out[:, 0].max()) == -7.0                 # nothing written outside the window
    ob = torch.empty((batch, M, N), device=DEV, dtype=torch.bfloat16)
    K.gemm_fp8(xq[:, 2:2 + M], sx, wq, sw, ob)
    ref2 = xr.double() @ wq.cpu().view(e4).float().double().t() * float(sx) * float(sw)
    close(ob, ref2.float().bfloat16().double(), 2e-2, 2e-2, "fp8 gemm bf16 out")      # within one bf16 ulp


@pytest.mark.parametrize("B,h,l", [(2, 2, 2), (1, 8, 3)])
def test_nys_fused_attention_sides_with_key_padding_mask(B, h, l):
    """The fused attention kernels with the package's key-padding mask (BASELINE config 4: 'fused attention mask path'):
    masked_fill(-finfo.max) on sim1 / sim3, uniform rows for fully masked queries / landmarks, zero gradient at filled logits —
    against f64 autograd through the plain masked products."""
    gen = g(300 * B + 10 * h + l)
    m, dh = 256, 64
    D, n_p, scale = h * dh, m * l, dh ** -0.5
    bf = torch.bfloat16
    qkv = (torch.randn((B, n_p, 3 * D), generator=gen) * 1.5).to(bf)
    lm = (torch.randn((B, m, 2 * D), generator=gen) * 1.5).to(bf)
    w2 = torch.randn((B, h, m, dh), generator=gen).to(bf)
    dout = torch.randn((B, n_p, D), generator=gen).to(bf)
    dav = torch.randn((B, h, m, dh), generator=gen).to(bf)
    # front padding + a ragged valid length per sample; landmark group j covers rows [j l, (j + 1) l)
    mrow = torch.zeros(B, n_p)
    for b in range(B):
        lo = 5 + 40 * b
        hi = n_p - (17 + 100 * b)
        mrow[b, lo:hi] = 1.0
    mlm = (mrow.reshape(B, m, l).sum(-1) > 0).float()
    assert float(mlm.min()) == 0.0 and float(mrow.min()) == 0.0

    def heads(t, which, parts):
        return t.double().view(B, t.shape[1], parts, h, dh)[:, :, which].permute(0, 2, 1, 3)

    qkv_r, lm_r, w2_r = (t.double().requires_grad_() for t in (qkv, lm, w2))
    q, k, v = (heads(qkv_r, i, 3) for i in range(3))
    ql, kl = heads(lm_r, 0, 2), heads(lm_r, 1, 2)
    neg = -torch.finfo(torch.float32).max
    mb, ml = mrow.bool()[:, None, :], mlm.bool()[:, None, :]
    s1 = (scale * q @ kl.transpose(-1, -2)).masked_fill(~(mb[..., None] & ml[..., None, :]), neg)
    out_ref = (torch.softmax(s1, -1) @ w2_r).permute(0, 2, 1, 3).reshape(B, n_p, D)
    s3 = (scale * ql @ k.transpose(-1, -2)).masked_fill(~(ml[..., None] & mb[..., None, :]), neg)
    av_ref = torch.softmax(s3, -1) @ v
    ((out_ref * dout.double()).sum() + (av_ref * dav.double()).sum()).backward()

    kmask = (mrow.to(DEV), mlm.to(DEV))
    qkv_d, lm_d, w2_d, dout_d, dav_d = (t.to(DEV) for t in (qkv, lm, w2, dout, dav))
    out = torch.full((B, n_p, D), float("nan"), device=DEV, dtype=bf)
    o1 = torch.empty_like(out)
    lse1 = K.nys_attn1_fwd(qkv_d, lm_d, w2_d, out, h, scale, kmask=kmask, o1=o1)
    av, lse3 = K.nys_attn3_fwd(qkv_d, lm_d, h, scale, kmask=kmask)
    out_ref, av_ref = out_ref.detach(), av_ref.detach()
    close(out, out_ref, 0.0, 2e-2 * float(out_ref.abs().max()), "masked attn1 out")
    close(av, av_ref, 0.0, 1e-2 * float(av_ref.abs().max()), "masked attn3 av")
    # a fully masked query row attends uniformly over the landmarks
    close(out[0, 0, :dh], w2[0, 0].double().mean(0), 0.0, 2e-2 * float(out_ref.abs().max()), "uniform row")

    dqkv = torch.full_like(qkv_d, float("nan"))
    dw2 = torch.zeros((B, h, m, dh), device=DEV)
    dlm = torch.zeros((B, m, 2 * D), device=DEV)
    K.nys_attn1_bwd(qkv_d, lm_d, w2_d, dout_d, lse1, o1, torch.empty_like(lse1), dqkv, dw2, dlm, h, scale, kmask=kmask)
    K.nys_attn3_bwd(qkv_d, lm_d, av, dav_d, lse3, dqkv, dlm, h, scale, kmask=kmask)
    for name, got, ref in (("dqkv", dqkv, qkv_r.grad), ("dw2", dw2, w2_r.grad), ("dlm", dlm, lm_r.grad)):
        close(got, ref, 0.0, 2e-2 * float(ref.abs().max()), "masked " + name)
        rel = float((got.float().cpu().double() - ref).norm() / ref.norm())
        assert rel < 1e-2, (name, rel)
    # q / k rows that are masked out get no gradient through the similarities (v rows still do, through attn3 only if valid)
    assert float(dqkv[0, 0, :2 * D].abs().max()) == 0.0


def test_shadow_cache_does_not_outlive_its_tensor():
    """A weight allocated at the address of a freed one (same shape, same version counter) must not inherit its cached bf16
    copy: the shadow caches are tied to the lifetime of the tensor they were made for."""
    from mirror_amd import functional as Fn
    w1 = torch.full((64, 128), 1.5, device=DEV)
    s1 = Fn.shadow(w1, Fn.BF16)
    assert float(s1.float().mean()) == 1.5
    ptr = w1.data_ptr()
    del w1, s1
    w2 = torch.full((64, 128), -2.0, device=DEV)
    if w2.data_ptr() != ptr:
        pytest.skip("the allocator did not reuse the block")
    assert float(Fn.shadow(w2, Fn.BF16).float().mean()) == -2.0
    p = torch.nn.Parameter(torch.full((64, 128), 3.0, device=DEV))
    Fn.register_shadow(p, torch.full((64, 128), 3.0, device=DEV, dtype=torch.bfloat16))
    pp = p.data_ptr()
    del p
    w3 = torch.full((64, 128), 0.5, device=DEV)
    if w3.data_ptr() == pp:
        assert float(Fn.shadow(w3, Fn.BF16).float().mean()) == 0.5


def test_attn1_forward_writes_the_e4m3_copy_of_its_output_with_the_delayed_scale():
    """mh_nys_attn1_fwd_q8 == mh_nys_attn1_fwd followed by mh_quant_fp8_delayed on `out` (accumulate mode, as the model calls it):
    same bf16 output and lse, same bytes, scale and ring update."""
    gen = g(654)
    B, h, n_p, m, dh = 2, 2, 512, 256, 64
    D = h * dh
    bf = torch.bfloat16
    qkv = torch.randn(B, n_p, 3 * D, generator=gen).to(DEV, bf)
    lm = torch.randn(B, m, 2 * D, generator=gen).to(DEV, bf)
    w2 = torch.randn(B, h, m, dh, generator=gen).to(DEV, bf)
    base = torch.randn(B, n_p, D, generator=gen).to(DEV, bf)
    tick = torch.full((1,), 7.0, device=DEV)
    ring_a = torch.zeros(3, device=DEV, dtype=torch.int32)
    ring_a[0] = torch.tensor([3.0], device=DEV).view(torch.int32)[0]       # (7 + 2) % 3 = 0: the previous step's maximum
    ring_a[2] = 77                                                         # (7 + 1) % 3 = 2: cleared
    ring_b = ring_a.clone()
    o1 = base.clone()
    l1 = K.nys_attn1_fwd(qkv, lm, w2, o1, h, dh ** -0.5, accumulate=True)
    q1, s1 = K.quant_fp8_delayed(o1, ring_a, tick)
    o2 = base.clone()
    q2 = torch.empty((B, n_p, D), device=DEV, dtype=torch.uint8)
    l2, s2 = K.nys_attn1_fwd_q8(qkv, lm, w2, o2, h, dh ** -0.5, True, q2, ring_b, tick)
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    assert torch.equal(q1, q2) and float(s1) == float(s2) and torch.equal(ring_a, ring_b)


def test_layernorm_writes_the_e4m3_copy_of_its_output_with_the_delayed_scale():
    """mh_layernorm_fwd_q8 == mh_layernorm_fwd followed by mh_quant_fp8_delayed on its bf16 output: same bf16 rows, mean / rstd,
    the same bytes, scale and ring update (behind `pad` front rows, as the Nystrom layers call it)."""
    gen = g(321)
    Bn, T, D, pad = 3, 70, 512, 6
    x = (torch.randn(Bn, T, D, generator=gen) * 2 + 0.3).to(DEV)
    gamma = (1 + 0.1 * torch.randn(D, generator=gen)).to(DEV)
    beta = (0.1 * torch.randn(D, generator=gen)).to(DEV)
    tick = torch.full((1,), 4.0, device=DEV)
    ring_a = torch.zeros(3, device=DEV, dtype=torch.int32)
    ring_a[0] = torch.tensor([2.5], device=DEV).view(torch.int32)[0]       # step 3's maximum (slot (4 + 2) % 3 = 0)
    ring_a[2] = 999                                                        # slot (4 + 1) % 3 = 2 must be cleared
    ring_b = ring_a.clone()
    bf = torch.bfloat16

    def bufs():
        y = torch.zeros((Bn, pad + T, D), device=DEV, dtype=bf)
        return y, torch.empty(Bn * T, device=DEV), torch.empty(Bn * T, device=DEV)
    y1, m1, r1 = bufs()
    K.layernorm_fwd(x, gamma, beta, y1[:, pad:], m1, r1, Bn, T, D, T * D, (pad + T) * D, 1e-5)
    q1, s1 = K.quant_fp8_delayed(y1[:, pad:].contiguous(), ring_a, tick)
    y2, m2, r2 = bufs()
    q2 = torch.zeros((Bn, pad + T, D), device=DEV, dtype=torch.uint8)
    s2 = K.layernorm_fwd_q8(x, gamma, beta, y2[:, pad:], m2, r2, Bn, T, D, T * D, (pad + T) * D, 1e-5, q2[:, pad:], ring_b, tick)
    assert torch.equal(y1, y2) and torch.equal(m1, m2) and torch.equal(r1, r2)
    assert torch.equal(q1, q2[:, pad:]) and float(s1) == float(s2) and int(q2[:, :pad].abs().max()) == 0
    assert torch.equal(ring_a, ring_b) and int(ring_b[2]) == 0


def test_fp8_delayed_scaling_quantisation_uses_last_steps_amax_and_rotates_its_ring():
    """mh_quant_fp8_delayed: scale = margin x amax of the PREVIOUS step (3-slot ring keyed by a device-side step counter), this
    step's amax lands in its own slot, the slot after it is cleared; values are bit-exact against torch.float8_e4m3fn at that
    scale, and values beyond the scale's range saturate at +-448."""
    gen = g(123)
    x1 = torch.randn(64, 512, generator=gen).to(DEV, torch.bfloat16)
    x2 = (torch.randn(64, 512, generator=gen) * 3).to(DEV, torch.bfloat16)
    ring = torch.zeros(3, device=DEV, dtype=torch.int32)
    tick = torch.zeros(1, device=DEV)
    a1 = float(x1.float().abs().max())
    ring[0] = torch.tensor([a1], device=DEV).view(torch.int32)[0]        # step 0 seeded by the exact path
    ring[2] = 12345                                                      # stale content of the slot step 1 must clear
    tick.fill_(1.0)
    q, sc = K.quant_fp8_delayed(x2, ring, tick, margin=1.25)
    assert abs(float(sc) - 1.25 * a1 / 448.0) <= 1e-6 * a1
    want = (x2.float() / sc).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q, want)
    got_amax = float(ring[1:2].view(torch.float32))
    assert got_amax == float(x2.float().abs().max()) and int(ring[2]) == 0 and float(ring[0:1].view(torch.float32)) == a1
    assert int((q.view(torch.float8_e4m3fn).float().abs() == 448).sum()) > 0     # 3x larger values at 1.25x headroom: some saturate
    tick.fill_(2.0)
    q3, sc3 = K.quant_fp8_delayed(x1, ring, tick, margin=1.0)
    assert abs(float(sc3) - got_amax / 448.0) <= 1e-6 * got_amax and int(ring[0]) == 0


@pytest.mark.parametrize("B,n_src,lead,wrap,l", [(8, 8192, 1, 89, 22), (2, 100, 1, 0, 7), (3, 64, 0, 0, 4), (2, 37, 1, 12, 5), (1, 5, 1, 4, 3)])
def test_keymask_plan_one_launch_equals_the_composed_plan(B, n_src, lead, wrap, l):  # noqa: E741
    """mh_keymask_plan (BASELINE config 4): row mask of the front-padded sequence [pad zeros | cls | mask | mask[:, :wrap]],
    landmark-group valid flags and l / (valid count + 1e-8), bit for bit what the ATen composition in Fn.KeyMask.plan gives
    (the `mask` handling of [3P] NystromAttention.forward); an all-masked group keeps flag 0 and the huge finite scale."""
    from mirror_amd import functional as Fn
    g = torch.Generator().manual_seed(B * 1000 + n_src)
    mask = torch.rand(B, n_src, generator=g) > 0.4
    mask[0, : min(n_src, 3 * l)] = False                    # whole groups without a valid row
    mask = mask.to(DEV)
    n = lead + n_src + wrap
    pad = (l - n % l) % l
    km = Fn.KeyMask(mask, lead=lead, wrap=wrap)
    assert tuple(km.shape) == (B, n)
    got = km.plan(pad, l)
    assert km.plan(pad, l) is got                           # kept per geometry: the next layer launches nothing
    Fn._KEYMASK_PLAN = False
    try:
        ref = Fn.KeyMask(mask, lead=lead, wrap=wrap).plan(pad, l)
    finally:
        Fn._KEYMASK_PLAN = True
    for a, b, name in zip(got, ref, ("mrow", "mlm", "lscale")):
        assert a.shape == b.shape and a.dtype == b.dtype == torch.float32, name
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
    assert bool(torch.isfinite(got[2]).all())
    if n_src >= 4 * l:
        assert float(got[1][0].min()) == 0.0 and float(got[2][0].max()) > 1e8
    with pytest.raises(K.MirrorHipError):
        K._lib.call("mh_keymask_plan", mask.data_ptr(), got[0].data_ptr(), got[1].data_ptr(), got[2].data_ptr(), B, n_src, lead, wrap, pad + 1, l,
                    stream=torch.cuda.current_stream().cuda_stream)
