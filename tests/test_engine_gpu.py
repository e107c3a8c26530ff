"""GPU: TrainEngine step semantics.  (1) one process: a step equals forward/backward + torch.optim.Adam on the
oracle-checked model; (2) two processes sharing the one GPU over gloo: the bucketed gradient reduction, the
parameter broadcast and the global-batch InfoNCE gather keep ranks in lock-step and match a single process that
sees the concatenated batch."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(wsi_embed_dim=64, rna_embed_dim=48, embed_dim=64, wsi_num_tokens=60, rna_encoder_depth=1, rna_num_heads=8,
           style_mlp_hidden_dim=64, style_mlp_out_dim=32, style_latent_dim=16, num_prototypes=50)


# D = 512: dh = 64, m = 256 landmarks -> the geometry of the fused Nystrom kernels and the one-launch pinv chain of the bf16
# policy (what bench.py times); 1000 tokens -> n = 1025 (cls + 1000 + 24 wrap-around), n_p = 1280, l = 5
CFG512 = dict(wsi_embed_dim=128, rna_embed_dim=96, embed_dim=512, wsi_num_tokens=1000, rna_encoder_depth=2, rna_num_heads=8,
              rna_mlp_ratio=4.0, style_mlp_hidden_dim=128, style_mlp_out_dim=64, style_latent_dim=32, num_prototypes=300)


def _make(seed=0, cfg=CFG):
    import mirror_amd.models as M
    torch.manual_seed(seed)
    return M.mirror(**cfg).cuda().eval()       # eval: dropout off so runs are comparable


def _batch(b, seed, cfg=CFG):
    g = torch.Generator().manual_seed(seed)
    n, f, gd, d, lat = cfg["wsi_num_tokens"], cfg["wsi_embed_dim"], cfg["rna_embed_dim"], cfg["embed_dim"], cfg["style_latent_dim"]
    wsi, rna = torch.randn(b, n, f, generator=g), torch.randn(b, gd, generator=g)
    noise = {"wsi_mask": torch.rand(b, n, generator=g), "rna_mask": torch.rand(b, d, generator=g),
             "wsi_eps": torch.randn(b, lat, generator=g), "rna_eps": torch.randn(b, lat, generator=g)}
    return wsi.cuda(), rna.cuda(), {k: v.cuda() for k, v in noise.items()}


def test_engine_step_equals_autograd_plus_torch_adam():
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    ref = _make()
    ref.precision = "fp32"
    model = _make()
    eng = TrainEngine(model, MIRRORLoss(), lr=1e-3, precision="fp32")
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    for step in range(3):
        wsi, rna, noise = _batch(4, 10 + step)
        with torch.no_grad():
            ref.prototypes.weight.copy_(torch.nn.functional.normalize(ref.prototypes.weight, dim=1))
        opt.zero_grad()
        l_ref = MIRRORLoss()(*ref(wsi, rna, noise=noise))
        l_ref[0].backward()
        opt.step()
        with torch.no_grad():
            ref.logit_scale.clamp_(0, 4.6052)
        l_eng = eng.step(wsi, rna, noise=noise)
        assert torch.allclose(l_eng[0], l_ref[0].detach(), rtol=1e-4), (step, l_eng[0], l_ref[0])
    # Adam normalises the step: an element whose gradient is rounding-level noise (f32 atomics order) can move by
    # +-lr in either run, so compare the two trajectories against the size of the update, not element-wise.
    init = dict(_make().named_parameters())
    num = den = 0.0
    for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        num += float((p - q).double().pow(2).sum())
        den += float((p - init[k]).double().pow(2).sum())
    assert num ** 0.5 < 0.05 * den ** 0.5, (num, den)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mirror_amd.engine import TrainEngine
        from mirror_amd.losses import MIRRORLoss
        model = _make(seed=rank)                      # different init per rank: the engine must broadcast rank 0's
        eng = TrainEngine(model, MIRRORLoss(gather_distributed=True), lr=1e-3, precision="fp32", bucket_mb=0.05)
        assert len(eng.buckets) > 2
        wsi, rna, noise = _batch(8, 77)               # the global batch; each rank takes its half
        sl = slice(rank * 4, rank * 4 + 4)
        losses = None
        for _ in range(2):
            losses = eng.step(wsi[sl], rna[sl], noise={k: v[sl] for k, v in noise.items()})
        flat = eng.master.detach().cpu().numpy()      # numpy: no tensor-sharing handshake with an exiting process
        # the same two ranks with bf16 gradient buckets on the wire (BASELINE config 5, grad_reduce_dtype="bf16"): the reduced
        # arena is the f32 one rounded per bucket — ranks stay identical, gradients within bf16 rounding of the f32 reduction
        m32, m16 = _make(seed=5), _make(seed=5)
        e32 = TrainEngine(m32, MIRRORLoss(gather_distributed=True), lr=1e-3, precision="fp32", bucket_mb=0.05, snapshot_grads=True)
        e16 = TrainEngine(m16, MIRRORLoss(gather_distributed=True), lr=1e-3, precision="fp32", bucket_mb=0.05, snapshot_grads=True,
                          grad_reduce_dtype="bf16")
        nz = {k: v[sl] for k, v in noise.items()}
        e32.step(wsi[sl], rna[sl], noise=nz)
        e16.step(wsi[sl], rna[sl], noise=nz)
        g32, g16 = e32.grad_snap, e16.grad_snap
        bf_rel = float((g16 - g32).norm() / g32.norm())
        bf_same = float((g16 - g16.to(torch.bfloat16).float()).abs().max())     # every reduced value is a bf16 number
        g16c = g16.cpu().numpy()
        q.put((rank, flat, float(losses[1]), bf_rel, bf_same, g16c))
    finally:
        dist.destroy_process_group()


def test_two_ranks_match_single_process_on_concatenated_batch():
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    import numpy as np
    assert np.array_equal(res[0][1], res[1][1]), "ranks diverged"
    for r in res:
        assert r[3] < 6e-3 and r[4] == 0.0, (r[3], r[4])        # bf16 wire format: 2^-8 relative rounding per element
    assert np.array_equal(res[0][5], res[1][5]), "ranks diverged under bf16 gradient buckets"
    # single process, whole batch.  The cluster/style/retention terms are batch means, so their gradients agree with
    # the average of the two half-batch gradients; the pinv initial scaling couples samples inside a rank's batch
    # (tensor-wide max), so agreement is to ~1e-3, not bitwise.
    model = _make(seed=0)
    eng = TrainEngine(model, MIRRORLoss(), lr=1e-3, precision="fp32")
    wsi, rna, noise = _batch(8, 77)
    for _ in range(2):
        losses = eng.step(wsi, rna, noise=noise)
    rel = (eng.master.cpu() - torch.from_numpy(res[0][1])).norm() / eng.master.cpu().norm()
    assert rel < 5e-3, rel
    assert abs(float(losses[1]) - 0.5 * (res[0][2] + res[1][2])) < 5e-3 * abs(float(losses[1]))


def test_graphed_step_replays_with_fresh_state():
    """Without injected noise the step is captured into a HIP graph after two eager steps.  Everything that changes
    from step to step must live on the device: Adam's t / bias corrections / lr, the dropout base offset, the
    noise draws.  Checks that replays keep training, advance the device state and honour an lr change."""
    import mirror_amd.models as M
    from mirror_amd import functional as Fn
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    torch.manual_seed(3)
    model = M.mirror(**CFG).cuda().train()
    eng = TrainEngine(model, MIRRORLoss(), lr=1e-3, precision="bf16", graph=True)
    Fn.manual_seed(7)
    wsi, rna, _ = _batch(4, 5)
    wsi = wsi.to(torch.bfloat16)
    snaps, losses, bases = [], [], []
    for step in range(6):
        out = eng.step(wsi, rna)
        losses.append(float(out[0]))
        snaps.append(eng.master.clone())
        bases.append(int(Fn._dropout_state["base"]))
    assert eng._graph is not None, "the step was not captured"
    assert all(torch.isfinite(torch.tensor(losses))), losses
    assert float(eng._state[0]) == 6.0 and abs(float(eng._state[1]) - (1 - 0.9 ** 6)) < 1e-6, eng._state
    for a, b in zip(snaps[:-1], snaps[1:]):
        assert float((a - b).abs().max()) > 0, "a replay did not update the parameters"
    assert bases[0] > 0 and all(b2 - b1 == bases[0] for b1, b2 in zip(bases[:-1], bases[1:])), bases
    assert len(set(losses[2:])) > 1, "replays produced identical losses: noise / dropout are not redrawn"
    eng.lr = 0.0                                   # an lr scheduler writes engine.lr between steps
    before = eng.master.clone()
    eng.step(wsi, rna)
    proto = model.prototypes.weight                # re-normalised at every step (train_mirror.py:1133-1136), lr or not
    for prm, off in zip(eng.params, eng.offsets):
        if prm.data_ptr() == proto.data_ptr():
            continue
        sl = slice(off, off + prm.numel())
        assert float((eng.master[sl] - before[sl]).abs().max()) == 0.0, "lr change was not published to the graphed step"
    assert float(eng._state[0]) == 7.0
    # static inputs: the same, unmodified batch tensor is not copied again; a refilled one (version bump) or a new tensor is
    assert eng._g_src[0] is wsi and torch.equal(eng._g_in[0], wsi)
    wsi.mul_(0.5)
    eng.step(wsi, rna)
    assert torch.equal(eng._g_in[0], wsi), "an in-place refill of the batch tensor did not reach the graph's static input"
    wsi2 = wsi.clone() + 1
    eng.step(wsi2, rna)
    assert torch.equal(eng._g_in[0], wsi2) and eng._g_src[0] is wsi2


def test_grad_clip_and_accumulation_match_torch():
    """clip_grad (global L2 norm, timm "norm" mode) and accum_steps against torch.optim.Adam + clip_grad_norm_ on the same
    model: two micro-batches of 2 accumulate to the gradient of their mean loss, then one clipped update."""
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    ref = _make()
    ref.precision = "fp32"
    model = _make()
    eng = TrainEngine(model, MIRRORLoss(), lr=1e-3, precision="fp32", clip_grad=0.05, accum_steps=2)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    init = {k: v.detach().clone() for k, v in _make().named_parameters()}
    for upd in range(2):
        with torch.no_grad():
            ref.prototypes.weight.copy_(torch.nn.functional.normalize(ref.prototypes.weight, dim=1))
        opt.zero_grad()
        for micro in range(2):
            wsi, rna, noise = _batch(2, 50 + 2 * upd + micro)
            (MIRRORLoss()(*ref(wsi, rna, noise=noise))[0] / 2).backward()
            eng.step(wsi, rna, noise=noise)
        gn = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.05)
        assert abs(float(eng._state[5]) - float(gn)) < 2e-3 * float(gn), (float(eng._state[5]), float(gn))
        opt.step()
        with torch.no_grad():
            ref.logit_scale.clamp_(0, 4.6052)
    assert float(eng._state[0]) == 2.0
    num = den = 0.0
    for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        num += float((p.detach() - q.detach()).double().pow(2).sum())
        den += float((p.detach() - init[k]).double().pow(2).sum())
    assert num ** 0.5 < 0.05 * den ** 0.5, (num, den)


def test_clip_mode_value_clamps_the_averaged_gradient_and_agc_is_refused():
    """timm's dispatch_clip_grad modes (train_mirror.py:1219-1229): 'value' = torch.nn.utils.clip_grad_value_ on the gradient Adam
    sees; 'agc' is not built and says so."""
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    with pytest.raises(NotImplementedError):
        TrainEngine(_make(), MIRRORLoss(), precision="fp32", clip_grad=0.1, clip_mode="agc")
    ref = _make()
    ref.precision = "fp32"
    model = _make()
    eng = TrainEngine(model, MIRRORLoss(), lr=1e-3, precision="fp32", clip_grad=1e-3, clip_mode="value", snapshot_grads=True)
    wsi, rna, noise = _batch(4, 61)
    with torch.no_grad():
        ref.prototypes.weight.copy_(torch.nn.functional.normalize(ref.prototypes.weight, dim=1))
    MIRRORLoss()(*ref(wsi, rna, noise=noise))[0].backward()
    eng.step(wsi, rna, noise=noise)
    torch.nn.utils.clip_grad_value_(ref.parameters(), 1e-3)
    want = torch.cat([p.grad.reshape(-1) for p in reversed(list(ref.parameters()))])
    got = torch.cat([eng.grad_snap[o:o + p.numel()] for p, o in zip(eng.params, eng.offsets)])
    # grad_snap is taken before the clamp: apply it here and compare; then check the update used the clamped values
    assert float((got.clamp(-1e-3, 1e-3) - want).norm()) < 2e-3 * float(want.norm())
    assert float(want.abs().max()) == pytest.approx(1e-3) and float(got.abs().max()) > 1e-3      # the clamp was active


def test_validate_matches_reference_loop_and_state_roundtrip(tmp_path):
    """validate() == the reference loop (eval mode, no grad, batch-size-weighted means), and the
    Adam state survives CheckpointSaver -> resume_checkpoint (train_mirror.py:772-780, :1053-1062)."""
    from mirror_amd.checkpoint import CheckpointSaver, resume_checkpoint
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    model = _make().train()                      # train mode going in: validate() must switch to eval and switch back
    eng = TrainEngine(model, MIRRORLoss(), lr=1e-3, precision="fp32", graph=False)
    batches, noises = [], []
    for i, b in enumerate((4, 4, 2)):           # ragged last batch: weights matter
        wsi, rna, noise = _batch(b, 50 + i)
        batches.append((wsi.cpu(), rna.cpu()))
        noises.append(noise)
    got = eng.validate(batches, noise=noises)
    assert model.training                        # mode restored
    model.eval()
    want = torch.zeros(6, dtype=torch.float64)
    with torch.no_grad():
        for (wsi, rna), noise in zip(batches, noises):
            ls = MIRRORLoss()(*model(wsi.cuda(), rna.cuda(), noise=noise))
            want += torch.stack([x.double().cpu() for x in ls]) * wsi.shape[0]
    want /= 10                                   # the model stays in eval from here on: dropout off keeps the two runs comparable
    assert list(got) == ["loss", "alignment_loss", "wsi_retention_loss", "rna_retention_loss", "style_loss", "cluster_loss"]
    assert torch.allclose(torch.tensor(list(got.values()), dtype=torch.float64), want, rtol=1e-6), (got, want)
    # two steps, checkpoint, two more steps  ==  resume into a fresh engine and take the same two steps
    for i in range(2):
        eng.step(*_batch(4, 70 + i))
    saver = CheckpointSaver(model, eng, checkpoint_dir=str(tmp_path), max_history=1)
    saver.save_checkpoint(0, metric=float(got["loss"]))
    tail = [_batch(4, 80 + i) for i in range(2)]
    for b in tail:
        eng.step(*b)
    model2 = _make(seed=5)
    eng2 = TrainEngine(model2, MIRRORLoss(), lr=1e-3, precision="fp32", graph=False)
    assert resume_checkpoint(model2, os.path.join(tmp_path, "last.pth.tar"), eng2) == 1
    assert eng2.step_count == 2
    for b in tail:
        eng2.step(*b)
    for (k, p), (_, q) in zip(model.named_parameters(), model2.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), k


# ---------------------------------------------------------------- graphed RNA branch of the eager step (mirror_amd/graphed.py)
def _run_eager(rna_graph: bool, steps: int, gather: bool = False, cfg=CFG, batch: int = 4, bucket_mb: float = 0.05,
               lr: float = 1e-3, grad_dtype: str = "f32", note=None):
    import mirror_amd.models as M
    from mirror_amd import functional as Fn
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    torch.manual_seed(11)
    model = M.mirror(**cfg, rna_proj_drop_rate=0.1).cuda().train()
    if note is not None:
        note("model built")
    eng = TrainEngine(model, MIRRORLoss(gather_distributed=gather), lr=lr, precision="bf16", graph=False, bucket_mb=bucket_mb,
                      grad_reduce_dtype=grad_dtype)
    if note is not None:
        note(f"engine built, {len(getattr(eng, 'buckets', []))} buckets")
    if not rna_graph:
        eng._rna_branch_state = "off"
    Fn.manual_seed(5)                  # the same dropout stream on every rank and in every variant: runs stay comparable
    init = eng.master.clone()
    losses = []
    eng.grad_snaps = []
    inner = eng._finish_reduce

    def snap():                      # runs after backward (+ side-stream join), before clip / Adam read the arena
        inner()
        eng.grad_snaps.append(eng.grad.clone())
    eng._finish_reduce = snap
    rank = dist.get_rank() if dist.is_initialized() else 0
    for s in range(steps):
        wsi, rna, noise = _batch(batch, 100 + 10 * s + rank, cfg)
        losses.append([float(x) for x in eng.step(wsi.to(torch.bfloat16), rna, noise=noise)])
        if note is not None:
            note(f"step {s} done")
    return eng, init, losses


def _traj_close(pa, pb, init, tol=0.05):
    num = float((pa - pb).double().pow(2).sum()) ** 0.5
    den = float((pa - init).double().pow(2).sum()) ** 0.5
    assert num < tol * den, (num, den)


def test_rna_branch_graph_matches_eager_launch():
    """Eager launch mode (what N > 1 runs): after two warm steps the RNA branch is replayed from a forward and a backward
    HIP graph.  Same dropout offsets, same kernels, same gradient arena: the trajectory must match the all-eager step."""
    eng_off, init, l_off = _run_eager(False, 6)
    eng_on, _, l_on = _run_eager(True, 6)
    assert eng_off._rna_branch_state == "off" and eng_on._rna_branch_state == "on"
    assert len(eng_on.model._rna_graph[0].params) > 20          # the RNA encoder's parameters are finished by the graph
    # ... and the recorded branch is what the later steps RUN (round 5: noise draws that took dropout offsets in front of the branch
    # silently sent every step down the eager branch — its offsets are baked in from 0)
    assert getattr(eng_on.model._rna_graph[0], "replays", 0) >= 3
    # step 2 is the first replayed one: every parameter's gradient must be there when Adam reads the arena (the replay
    # runs on the RNA side stream) and equal the eager gradient up to the f32-atomics noise of the two earlier steps
    g_off, g_on = eng_off.grad_snaps[2], eng_on.grad_snaps[2]
    for p, o in zip(eng_on.params, eng_on.offsets):
        a, b = g_off[o:o + p.numel()], g_on[o:o + p.numel()]
        assert float((a - b).norm()) <= 2e-2 * float(a.norm()) + 1e-6, (o, float(a.norm()), float(b.norm()))
    for a, b in zip(l_off, l_on):     # bf16 + Adam amplify rounding noise from step to step (eager vs eager: ~2e-3 by step 5)
        for x, y in zip(a, b):
            assert abs(x - y) <= 1e-2 * max(1.0, abs(x)), (a, b)
    _traj_close(eng_off.master, eng_on.master, init)
    # a different batch size falls back to the eager branch (no capture for it), and evaluation never replays
    wsi, rna, noise = _batch(2, 999)
    out = eng_on.step(wsi.to(torch.bfloat16), rna, noise=noise)
    assert all(torch.isfinite(x) for x in out)


def _worker_rna_graph(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng_off, init, l_off = _run_eager(False, 5, gather=True)
        eng_on, _, l_on = _run_eager(True, 5, gather=True)
        # step 1 is an eager, bucketed step in both engines: the reduced gradients must be reproducible (they were not
        # while sunk parameters reported to their bucket twice and buckets were reduced early and again at the end)
        a, b = eng_off.grad_snaps[1], eng_on.grad_snaps[1]
        rerun = float((a - b).norm() / a.norm())
        q.put((rank, eng_off.master.cpu().numpy(), eng_on.master.cpu().numpy(), init.cpu().numpy(), eng_on._rna_branch_state,
               l_off[-1][0], l_on[-1][0], rerun))
    finally:
        dist.destroy_process_group()


def test_rna_branch_graph_under_bucketed_all_reduce_world2():
    """Two gloo ranks on the one GPU: the replayed backward graph reports its parameters to the bucket logic itself
    (no autograd hooks fire for them), ranks stay bit-identical and the result matches the all-eager ranks."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_worker_rna_graph, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][4] == "on" and res[1][4] == "on"
    # (a doubly reduced bucket is off by a factor of two; the weights behind step 1 carry one Adam update, whose first step is
    # lr * sign(g), so parameters whose gradient is f32-atomics noise around zero make the re-run differ by up to ~1e-3)
    assert res[0][7] < 2e-2 and res[1][7] < 2e-2, (res[0][7], res[1][7])
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2]), "ranks diverged"
    _traj_close(torch.from_numpy(res[0][1]), torch.from_numpy(res[0][2]), torch.from_numpy(res[0][3]))
    assert abs(res[0][5] - res[0][6]) < 5e-3 * abs(res[0][5])


def test_transposed_shadow_follows_the_optimizer_for_unmanaged_shapes():
    """The engine keeps W^T shadows only for weights whose dims are multiples of 32; for the others (here the style decoder
    with a 16-wide latent) functional.shadow_t must transpose the LIVE bf16 shadow: mh_adam rewrites the master through a raw
    pointer, so a version-keyed cache would serve the first step's W^T for ever."""
    import mirror_amd.models as M
    from mirror_amd import functional as Fn
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    torch.manual_seed(5)
    model = M.mirror(**CFG).cuda().train()
    eng = TrainEngine(model, MIRRORLoss(), lr=1e-2, precision="bf16", graph=False)
    prec = Fn.POLICIES["bf16"]
    odd = [p for p in eng.params if p.dim() == 2 and (p.shape[0] % 32 or p.shape[1] % 32) and p.shape[0] % 32 == 0]
    assert odd, "the test configuration lost its non-multiple-of-32 weight"
    for s in range(3):
        wsi, rna, noise = _batch(4, 40 + s)
        eng.step(wsi.to(torch.bfloat16), rna, noise=noise)
        for w in odd:
            assert torch.equal(Fn.shadow_t(w, prec), Fn.shadow(w, prec).t().contiguous()), (s, tuple(w.shape))


# ---------------------------------------------------------------- the same engine paths at D = 512 (fused kernels + pinv chain)
def test_d512_graph_replay_matches_eager_launch():
    """Whole-step HIP graph vs eager launch at D = 512 (bf16 policy, train mode): the fused Nystrom kernels, the one-launch
    pinv chain on its side stream, the deferred v columns and the gradient sink all sit inside the captured step.  Same
    seeds: losses of every step and the final parameters agree to f32-atomics noise."""
    import mirror_amd.models as M
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    runs = []
    for graph in (True, False):
        torch.manual_seed(21)
        model = M.mirror(**CFG512, rna_proj_drop_rate=0.1).cuda().train()
        eng = TrainEngine(model, MIRRORLoss(), lr=1e-4, precision="bf16", graph=graph, seed=77, snapshot_grads=True)
        if not graph:
            eng._rna_branch_state = "off"
        init = eng.master.clone()
        wsi, rna, _ = _batch(4, 5, CFG512)
        wsi = wsi.to(torch.bfloat16)
        torch.manual_seed(123)
        losses = [[float(x) for x in eng.step(wsi, rna)] for _ in range(6)]
        assert (eng._graph is not None) == graph
        runs.append((losses, eng.master.clone(), eng.grad_snap.clone(), init))
    (la, pa, ga, init), (lb, pb, gb, _) = runs
    for a, b in zip(la, lb):
        for x, y in zip(a, b):
            assert abs(x - y) <= 2e-3 * max(1.0, abs(y)), (la, lb)
    assert float((ga - gb).norm()) <= 2e-2 * float(gb.norm())
    _traj_close(pa, pb, init)


def test_early_adam_for_the_rna_encoder_equals_the_single_launch(monkeypatch):
    """Round 5: the RNA encoder's parameters (one contiguous range of the arena) are updated on the RNA branch's stream as soon as that
    stream has flushed its weight gradients, the rest behind the backward with tick=False and a hole (TrainEngine.step, mh_adam).  Same
    seeds, whole-step graph: losses, first / second moments and the parameter trajectory equal those of the one-launch step up to the
    gradients' own f32-atomics noise; the device step counter advances once per step; configurations that couple the ranges (global
    clipping norm, accumulation) keep the single launch."""
    import mirror_amd.models as M
    from mirror_amd import engine as E
    from mirror_amd import kernels as K
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    calls = []
    real = K.adam

    def counting(*a, **kw):
        calls.append((a[0].numel(), kw.get("tick", True), kw.get("hole")))
        return real(*a, **kw)
    monkeypatch.setattr(K, "adam", counting)
    runs = []
    for early in (True, False):
        monkeypatch.setattr(E, "_EARLY_ADAM", early)
        torch.manual_seed(21)
        model = M.mirror(**CFG512, rna_proj_drop_rate=0.1).cuda().train()
        eng = TrainEngine(model, MIRRORLoss(), lr=1e-6, precision="bf16", graph=True, seed=77)
        lo, hi = eng._early_range
        assert hi - lo > 0.3 * eng.numel and lo % 8 == 0 and hi % 8 == 0
        init = eng.master.clone()
        wsi, rna, _ = _batch(4, 5, CFG512)
        wsi = wsi.to(torch.bfloat16)
        calls.clear()
        losses = [[float(x) for x in eng.step(wsi, rna)] for _ in range(5)]
        assert eng._graph is not None and float(eng._state[0]) == 5.0
        if early:       # (eager steps before the capture + the capture itself: two launches each)
            assert any(c == (hi - lo, "early", None) for c in calls) and any(c == (eng.numel, False, (lo, hi)) for c in calls)
        else:
            assert all(c[1] is True and c[2] is None and c[0] == eng.numel for c in calls)
        runs.append((losses, eng.master.clone(), eng.m.clone(), eng.v.clone(), init))
    (la, pa, ma, va, init), (lb, pb, mb, vb, _) = runs
    for a, b in zip(la, lb):
        for x, y in zip(a, b):
            assert abs(x - y) <= 2e-3 * max(1.0, abs(y)), (la, lb)
    _traj_close(pa, pb, init)
    assert float((ma - mb).norm()) <= 2e-2 * float(mb.norm()) and float((va - vb).norm()) <= 4e-2 * float(vb.norm())
    # a global clipping norm couples the two ranges: one launch behind the backward
    monkeypatch.setattr(E, "_EARLY_ADAM", True)
    torch.manual_seed(21)
    eng = TrainEngine(M.mirror(**CFG512).cuda().train(), MIRRORLoss(), lr=1e-6, precision="bf16", graph=False, seed=77, clip_grad=1.0)
    calls.clear()
    eng.step(wsi, rna)
    assert calls == [(eng.numel, True, None)]


def test_d512_graph_replay_with_key_padding_mask_matches_eager_launch():
    """BASELINE config 4 under the whole-step HIP graph (round 5): the key-padding mask is one more STATIC INPUT of the captured step,
    refreshed like the batch.  Graph replay vs eager launch at D = 512 with padded slides (bf16 policy, train mode, same seeds): losses
    of every step, the last step's gradients and the parameter trajectory agree to f32-atomics noise; a batch with a DIFFERENT mask in
    the same tensor reaches the replay (its loss differs, and equals the eager run's), and a maskless batch on a masked graph falls
    back to the eager step instead of replaying with a stale mask."""
    import mirror_amd.models as M
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    n = CFG512["wsi_num_tokens"]
    runs = []
    for graph in (True, False):
        torch.manual_seed(21)
        model = M.mirror(**CFG512, rna_proj_drop_rate=0.1).cuda().train()
        # lr 1e-6: at 1e-4 six Adam steps (each ~lr per element whatever the gradient's size) amplify the f32-atomics noise of step 1 into
        # ~1 % of the step-6 gradient, and into 5 % of _fc1's bias gradient (a cancelling column sum) in 3 runs of 10 — measured on the
        # round-4 kernels too; the trajectory check below is relative to the distance travelled, so it does not depend on lr
        eng = TrainEngine(model, MIRRORLoss(), lr=1e-6, precision="bf16", graph=graph, seed=77, snapshot_grads=True)
        if not graph:
            eng._rna_branch_state = "off"
        init = eng.master.clone()
        wsi, rna, _ = _batch(4, 5, CFG512)
        lens = torch.tensor([n, 700, 333, 512], device="cuda")
        mask = torch.arange(n, device="cuda")[None, :] < lens[:, None]
        # the padded rows keep their (random) features: zeroed rows that the SECOND mask below declares real would all have the
        # pre-activation _fc1.bias[c] — zero at init, +-lr per Adam step — so hundreds of rows' ReLU gates in channel c would follow
        # the sign of a gradient at the noise floor (measured: 10 % of the bias gradient jumps between two identical eager runs)
        wsi = wsi.to(torch.bfloat16)
        torch.manual_seed(123)
        losses = [[float(x) for x in eng.step(wsi, rna, wsi_key_padding_mask=mask)] for _ in range(5)]
        assert (eng._graph is not None) == graph
        if graph:
            assert eng._g_in[2] is not None and torch.equal(eng._g_in[2], mask)
        # another mask written INTO the same tensor (the version counter moves): the replay must see it
        mask.copy_(torch.arange(n, device="cuda")[None, :] < torch.tensor([600, n, 400, 900], device="cuda")[:, None])
        losses.append([float(x) for x in eng.step(wsi, rna, wsi_key_padding_mask=mask)])
        if graph:
            assert torch.equal(eng._g_in[2], mask)
        runs.append((losses, eng.master.clone(), eng.grad_snap.clone(), init))
        if graph:        # a maskless batch must not replay the masked graph
            before = eng._graph
            eng.step(wsi, rna)
            assert eng._graph is before
    (la, pa, ga, init), (lb, pb, gb, _) = runs
    for a, b in zip(la, lb):
        for x, y in zip(a, b):
            assert abs(x - y) <= 2e-3 * max(1.0, abs(y)), (la, lb)
    assert abs(la[-1][0] - la[-2][0]) > 1e-4, "the second mask did not change the loss"
    assert float((ga - gb).norm()) <= 2e-2 * float(gb.norm())
    _traj_close(pa, pb, init)


def _worker_d512(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # lr 2e-5 (the bench's): Adam's first update moves EVERY element by lr * sign(g), so with a large lr the elements whose
        # gradient is rounding noise already separate two runs by ~1 % in the step-1 gradients
        eng_off, init, l_off = _run_eager(False, 4, gather=True, cfg=CFG512, batch=2, bucket_mb=1.0, lr=2e-5)
        eng_on, _, l_on = _run_eager(True, 4, gather=True, cfg=CFG512, batch=2, bucket_mb=1.0, lr=2e-5)
        eng_off2, _, _ = _run_eager(False, 2, gather=True, cfg=CFG512, batch=2, bucket_mb=1.0, lr=2e-5)
        a, b, c = eng_off.grad_snaps[1], eng_on.grad_snaps[1], eng_off2.grad_snaps[1]   # step 1: eager + bucketed in all three
        rerun = float((a - b).norm() / a.norm())
        floor = float((a - c).norm() / a.norm())     # all-eager vs all-eager: bf16 1-ulp flips from the f32 atomics order
        q.put((rank, eng_off.master.cpu().numpy(), eng_on.master.cpu().numpy(), init.cpu().numpy(), eng_on._rna_branch_state,
               l_off[-1][0], l_on[-1][0], rerun, len(eng_on.buckets), floor))
    finally:
        dist.destroy_process_group()


def test_d512_bucketed_all_reduce_and_rna_graph_world2():
    """Two gloo ranks on the one GPU at D = 512, bf16 policy, global-batch InfoNCE: the pinv chain's side stream, the fused
    attention backward and the graphed RNA branch all feed the bucketed all-reduce; ranks must stay bit-identical, the
    reduced gradients reproducible, and the graphed-branch trajectory must match the all-eager one."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 90)
    procs = [ctx.Process(target=_worker_d512, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][8] > 2, "the test needs several buckets"
    assert res[0][4] == "on" and res[1][4] == "on"
    # reproducible up to the run-to-run floor of this bf16 configuration (measured ~6e-3 for both differences)
    assert res[0][7] < 3 * res[0][9] + 1e-4 and res[0][9] < 2e-2, (res[0][7], res[0][9])
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2]), "ranks diverged"
    _traj_close(torch.from_numpy(res[0][1]), torch.from_numpy(res[0][2]), torch.from_numpy(res[0][3]), tol=0.1)
    assert abs(res[0][5] - res[0][6]) < 1e-2 * abs(res[0][5])


def test_transposed_shadows_are_current_for_a_backward_outside_the_engine():
    """TrainEngine rebuilds the transposed bf16 weight copies at the START of its next step (beside the forward) instead of
    behind Adam.  Code that differentiates through the model between two steps must still see the weights of the last update:
    functional.shadow_t asks the engine to catch up first."""
    from mirror_amd import functional as Fn
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    model = _make().train()
    eng = TrainEngine(model, MIRRORLoss(), lr=1e-2, precision="bf16", graph=False)
    for s in range(2):
        wsi, rna, noise = _batch(2, 700 + s)
        eng.step(wsi.to(torch.bfloat16), rna, noise=noise)
    assert eng.shadow_t is not None and eng._t_stale
    prec = Fn.POLICIES["bf16"]
    checked = 0
    for p, _ in eng._t_params[:6]:
        wt = Fn.shadow_t(p, prec)                       # the lookup a Linear backward does
        assert torch.equal(wt, Fn.shadow(p, prec).t().contiguous()), "stale transposed copy"
        checked += 1
    assert checked > 0 and not eng._t_stale


@pytest.mark.parametrize("graph", [False, True])
def test_a_backward_outside_step_never_leaks_into_the_next_update(graph):
    """A backward pass between two steps accumulates into the arena views (p.grad).  The engine clears the arena for it
    (refresh_transposes_now) AND the next step — eager, or the step that is captured next and every replay of it — must
    still start from zero: step, outside backward, step equals the same steps without the outside backward."""
    from mirror_amd import functional as Fn
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss

    def run(pollute):
        torch.manual_seed(21)
        Fn.manual_seed(22)
        model = _make(seed=6)
        eng = TrainEngine(model, MIRRORLoss(), lr=1e-4, precision="bf16", graph=graph, snapshot_grads=True)
        wsi, rna, noise = _batch(2, 810)
        wsi = wsi.to(torch.bfloat16)
        snaps = []
        for s in range(4):
            eng.step(wsi, rna)
            snaps.append(eng.grad_snap.clone())
            if pollute and s == 1:                    # with graph=True: after the second warm step, before the capture
                assert eng._zero_pending
                MIRRORLoss()(*model(wsi, rna, noise=noise))[0].backward()
                assert float(eng.grad.abs().max()) > 0.0 and eng._zero_pending
        if graph:
            assert eng._graph is not None, "the step was not captured"
        torch.cuda.synchronize()
        return snaps

    clean, dirty = run(False), run(True)
    for s, (a, b) in enumerate(zip(clean, dirty)):
        assert float((a - b).norm()) < 5e-2 * float(a.norm()), (s, float((a - b).norm()), float(a.norm()))


# BASELINE configs[2] / [4] composition at configs[1] shapes: two ranks, global-batch InfoNCE, bf16 gradient buckets on the wire
CFG_C2 = dict(wsi_embed_dim=1024, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=4096, rna_encoder_depth=6, rna_num_heads=8,
              rna_mlp_ratio=4.0, style_mlp_hidden_dim=512, style_mlp_out_dim=256, style_latent_dim=128, num_prototypes=3000)


def _worker_c2(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(root, exist_ok=True)

        def note(msg):                      # heartbeat (a long silent GPU run is taken to be hung)
            with open(os.path.join(root, f"r04_world2_c2_progress_rank{rank}.txt"), "a") as fh:
                fh.write(f"{time.strftime('%H:%M:%S')} {msg}\n")
        note("start")
        kw = dict(gather=True, cfg=CFG_C2, batch=2, bucket_mb=25.0, lr=2e-5, grad_dtype="bf16")
        e1, init, l1 = _run_eager(True, 4, note=note, **kw)
        torch.cuda.synchronize()
        note(f"4 steps done, losses {l1[-1]}")
        # host time of one eager step (what every rank of an N > 1 job pays per step): steps 4.. with the RNA branch replayed
        wsi, rna, noise = _batch(2, 555 + rank, CFG_C2)
        wsi = wsi.to(torch.bfloat16)
        host = []
        for _ in range(3):
            t0 = time.perf_counter()
            e1.step(wsi, rna, noise=noise)
            host.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        note(f"host ms {host}")
        g1 = e1.grad_snaps[1].clone()
        m1 = e1.master.cpu().numpy()
        state1 = e1._rna_branch_state
        nb = len(e1.buckets)
        del e1
        torch.cuda.empty_cache()
        e2, _, l2 = _run_eager(True, 2, **kw)             # the same two steps again: the run-to-run floor of this configuration
        floor = float((g1 - e2.grad_snaps[1]).norm() / g1.norm())
        bf_exact = float((g1 - g1.to(torch.bfloat16).float()).abs().max())
        note(f"rerun done, floor {floor}")
        import hashlib
        q.put((rank, hashlib.sha256(m1.tobytes()).hexdigest(), state1, l1[-1], floor, bf_exact, nb, host, torch.isfinite(g1).all().item()))
    finally:
        dist.destroy_process_group()


def test_c2_shapes_world2_global_infonce_bf16_buckets():
    """configs[1] shapes (4096 x 1024-d tokens, 2048 genes, D = 512, RNA depth 6, B = 2 per rank), two gloo ranks sharing the GPU,
    `gather_distributed=True` (BASELINE config 3's global-batch InfoNCE) + bf16 gradient buckets (config 5's wire format), eager launch
    with the RNA branch replayed from its HIP graphs — what every rank of the 8-GPU job runs.  Ranks end bit-identical, every reduced
    gradient is a bf16 number, and a re-run reproduces the step-1 gradient arena to the run-to-run floor of this configuration.
    Writes the wall time of the eager steps to gpurun_out/r04_world2_c2_host_ms.json (with gloo they are dominated by its blocking CPU
    all-reduce of 6 x 25 MiB; the launch-side host time of the eager step is bench.py's MIRROR_GRAPH=0 MIRROR_BENCH_HOSTTIME=1 line)."""
    import json
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_worker_c2, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue as _queue
    res = []
    for _ in range(600):                   # a worker that died must fail the test at once, not after the queue's timeout
        try:
            res.append(q.get(timeout=2))
        except _queue.Empty:
            assert all(p.exitcode in (None, 0) for p in procs), [p.exitcode for p in procs]
        if len(res) == 2:
            break
    assert len(res) == 2, "workers timed out"
    res.sort(key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][2] == "on" and res[1][2] == "on"
    assert res[0][1] == res[1][1], "ranks diverged"          # sha256 of the f32 master arena (136 MB per rank)
    assert res[0][8] and res[1][8]
    assert res[0][5] == 0.0 and res[1][5] == 0.0              # bf16 wire format: every reduced value is a bf16 number
    assert res[0][4] < 3e-2 and res[1][4] < 3e-2, (res[0][4], res[1][4])
    assert res[0][6] >= 4                                      # ~136 MB of f32 gradients in 25 MiB buckets
    assert all(np.isfinite(res[0][3]))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "r04_world2_c2_host_ms.json"), "w") as fh:
        json.dump({"config": "c2 shapes, B = 2 per rank, 2 gloo ranks on one GPU, gather_distributed, bf16 buckets, eager + graphed RNA branch",
                   "wall_ms_per_step_rank0_incl_gloo_cpu_allreduce": res[0][7], "wall_ms_per_step_rank1_incl_gloo_cpu_allreduce": res[1][7],
                   "buckets": res[0][6],
                   "run_to_run_grad_floor": [res[0][4], res[1][4]]}, fh)


def test_force_update_flushes_a_partial_accumulation_window_and_ranks_seed_dropout_differently():
    """(1) train_mirror.py:1128-1131: `need_update = last_batch or (batch_idx + 1) % accum_steps == 0` — the last, partial
    window of an epoch still updates, and the reference divides its tail batches by `last_accum_steps`, the number of batches
    in that window (:1117-1131, :1192-1196).  With accum_steps = 3, two micro-steps and force_update on the second: one
    optimizer step, gradients = (g1 + g2) / 2, nothing left for the next window.
    (2) `seed=` folds the rank in (utils.random_seed(args.seed, args.rank), :682): two engines built with ranks' seeds draw
    different dropout masks, the same seed reproduces."""
    from mirror_amd import functional as Fn
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    ref = _make()
    ref.precision = "fp32"
    model = _make()
    eng = TrainEngine(model, MIRRORLoss(), lr=1e-3, precision="fp32", accum_steps=3, snapshot_grads=True)
    with torch.no_grad():
        ref.prototypes.weight.copy_(torch.nn.functional.normalize(ref.prototypes.weight, dim=1))
    for micro in range(2):
        wsi, rna, noise = _batch(2, 300 + micro)
        (MIRRORLoss()(*ref(wsi, rna, noise=noise))[0] / 2).backward()
        eng.step(wsi, rna, noise=noise, force_update=(micro == 1))
    assert float(eng._state[0]) == 1.0 and eng._micro == 0
    want = torch.cat([p.grad.reshape(-1) for p in reversed(list(ref.parameters()))])
    assert eng.last_grad_scale == 0.5          # what mh_adam / mh_grad_clip were handed: 1 / (world * micro-steps in the window)
    got = torch.cat([eng.grad_snap[o:o + p.numel()] for p, o in zip(eng.params, eng.offsets)]) * eng.last_grad_scale
    assert float((got - want).norm()) < 2e-3 * float(want.norm())
    # nothing leaks into the next window: the arena is cleared behind the update, or (default) beside the next step's forward —
    # then the first micro-step of the next window must leave exactly its own gradient there
    assert eng._zero_pending or float(eng.grad.abs().max()) == 0.0
    for p in ref.parameters():
        p.grad = None
    wsi, rna, noise = _batch(2, 302)
    eng.step(wsi, rna, noise=noise)
    torch.cuda.synchronize()
    assert eng._micro == 1 and float(eng._state[0]) == 1.0
    ref.load_state_dict(model.state_dict())            # the engine's weights after its one update
    (MIRRORLoss()(*ref(wsi, rna, noise=noise))[0]).backward()
    want = torch.cat([p.grad.reshape(-1) for p in reversed(list(ref.parameters()))])
    got = torch.cat([eng.grad[o:o + p.numel()] for p, o in zip(eng.params, eng.offsets)])
    assert float((got - want).norm()) < 2e-3 * float(want.norm()), (float((got - want).norm()), float(want.norm()))

    def masks(seed):
        m = _make().train()
        TrainEngine(m, MIRRORLoss(), precision="bf16", seed=seed, graph=False)
        x = torch.ones(4, 4096, device="cuda")
        return Fn.dropout(x, 0.5, True)
    a, b, c = masks(10), masks(11), masks(10)
    assert torch.equal(a, c) and not torch.equal(a, b)


def test_host_feeder_graph_replay_trains_on_every_batch_across_epochs():
    """HostFeeder (train_mirror.py:1138-1139 made asynchronous) in front of the graph-replayed step: host f32 batches are cast
    to bf16 on the device by a raw kernel into per-slot buffers.  With an odd number of batches per epoch the last batch of
    epoch e and the first of epoch e + 1 used to land in the same slot tensor at the same torch version, so the replay skipped
    its copy into the static input and trained on the previous batch's slide features.  Two epochs of three batches: after
    every step the graph's static input must hold exactly the batch that was fed, and the losses must equal a run that is
    handed resident device tensors."""
    from mirror_amd import functional as Fn
    from mirror_amd.data import HostFeeder
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    host = []
    for i in range(3):
        w, r, _ = _batch(2, 900 + i)
        host.append((w.cpu().float() * (1.0 + i), r.cpu().float()))

    def run(use_feeder):
        torch.manual_seed(11)
        Fn.manual_seed(13)
        model = _make(seed=4)
        eng = TrainEngine(model, MIRRORLoss(), lr=1e-3, precision="bf16", graph=True)
        out, seen = [], []
        for _epoch in range(2):
            if use_feeder:
                src = HostFeeder(host, "cuda", wsi_dtype=torch.bfloat16) if _epoch == 0 else src
                it = iter(src)
            else:
                it = iter([(w.cuda().to(torch.bfloat16), r.cuda()) for w, r in host])
            for i, (w, r) in enumerate(it):
                assert w.dtype == torch.bfloat16 and w.is_cuda
                losses = eng.step(w, r)
                out.append([float(x) for x in losses])
                if eng._graph is not None:
                    seen.append(bool(torch.equal(eng._g_in[0], host[i][0].cuda().to(torch.bfloat16))))
        torch.cuda.synchronize()
        return out, seen, eng

    got, seen, eng = run(True)
    assert eng._graph is not None, "the step was not captured: this test has to exercise the replay path"
    assert seen and all(seen), seen
    want, _, _ = run(False)
    # two bf16 runs of six Adam steps at lr = 1e-3 differ by their f32 atomics order (measured up to 3e-3 on the smallest term); a batch
    # that was skipped or fed twice moves every term by tens of percent
    for a, b in zip(got, want):
        for x, y in zip(a, b):
            assert abs(x - y) <= 1e-2 * max(abs(y), 1e-3), (got, want)
