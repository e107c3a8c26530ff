"""CPU, world_size 2 over gloo: the data-parallel identities of SURVEY.md §8e.

(1) global-batch InfoNCE: mean over ranks of the local-rows loss on gathered embeddings == ClipLoss on the
    concatenated batch, and the gradient each rank receives (through the product's all-gather autograd function,
    whose backward reduce-scatters) equals P x the concatenated-batch gradient of its slice (DDP then averages).
(2) bucketed gradient averaging over arena slices == averaging the flat gradient."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from mirror_amd.engine import plan_buckets
from mirror_amd.losses.mirror_loss import _AllGatherCat, _gather_inflight, prefetch_alignment_gather
from oracle import mirror_oracle as O

WORLD, B, D = 2, 5, 16


def _worker(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        g = torch.Generator().manual_seed(0)
        w_all = torch.randn(WORLD * B, D, generator=g)
        r_all = torch.randn(WORLD * B, D, generator=g)
        scale = torch.tensor(9.0)
        w = w_all[rank * B:(rank + 1) * B].clone().requires_grad_(True)
        r = r_all[rank * B:(rank + 1) * B].clone().requires_grad_(True)
        both = _AllGatherCat.apply(torch.cat([w, r], dim=1))
        wa, ra = both[:, :D], both[:, D:]
        lab = torch.arange(B) + rank * B
        local = 0.5 * (F.cross_entropy(scale * w @ ra.T, lab) + F.cross_entropy(scale * r @ wa.T, lab))
        local.backward()
        tot = local.detach().clone()
        dist.all_reduce(tot)
        # reference: ClipLoss on the concatenated batch (oracle restatement of losses/mirror_loss.py:37-52)
        wc, rc = w_all.clone().requires_grad_(True), r_all.clone().requires_grad_(True)
        ref = O.clip_loss(wc, rc, scale)
        ref.backward()
        ok1 = torch.allclose(tot / WORLD, ref.detach(), rtol=1e-5, atol=1e-6)
        ok2 = torch.allclose(w.grad / WORLD, wc.grad[rank * B:(rank + 1) * B], rtol=1e-4, atol=1e-6)
        ok3 = torch.allclose(r.grad / WORLD, rc.grad[rank * B:(rank + 1) * B], rtol=1e-4, atol=1e-6)
        # (2) bucketed AVG all-reduce over arena slices
        sizes = [13, 200, 7, 64, 300]
        buckets, _ = plan_buckets(sizes, cap_elems=128)
        flat = torch.arange(buckets[-1][1], dtype=torch.float32) * (rank + 1)
        expect = torch.arange(buckets[-1][1], dtype=torch.float32) * (sum(range(1, WORLD + 1)) / WORLD)
        for s, e, _n in buckets:
            dist.all_reduce(flat[s:e])
            flat[s:e] /= WORLD
        ok4 = torch.allclose(flat, expect)
        # (3) the same gather issued ahead of the loss (prefetch_alignment_gather, what MIRROR.forward does right behind the heads):
        # the loss-side node only awaits it; values and gradients are those of the synchronous gather
        w2, r2 = w.detach().clone().requires_grad_(True), r.detach().clone().requires_grad_(True)
        prefetch_alignment_gather(w2, r2)
        pre = _gather_inflight.pop((w2.data_ptr(), r2.data_ptr()))
        both2 = _AllGatherCat.apply(torch.cat([w2, r2], dim=1), None, pre)
        ok5 = torch.equal(both2, both.detach())
        wa2, ra2 = both2[:, :D], both2[:, D:]
        (0.5 * (F.cross_entropy(scale * w2 @ ra2.T, lab) + F.cross_entropy(scale * r2 @ wa2.T, lab))).backward()
        ok5 = ok5 and torch.allclose(w2.grad, w.grad) and torch.allclose(r2.grad, r.grad)
        q.put((rank, ok1, ok2, ok3, ok4, ok5))
    finally:
        dist.destroy_process_group()


def test_global_infonce_and_bucketed_average_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(WORLD)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(WORLD))
    for r in res:
        assert all(r[1:]), r
