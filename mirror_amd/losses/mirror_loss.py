"""MIRRORLoss / ClipLoss on HIP kernels — host-side mirror of the reference's `losses/mirror_loss.py`.

Same constructor kwargs, argument order and 6-tuple result (losses/mirror_loss.py:55-135).  Each term is one
fused kernel pair (forward + hand-derived backward); only the weighted sum of five 0-d scalars is left to torch.

Build-only extension: `gather_distributed=True` contrasts the local rows against the embeddings of ALL ranks
(one RCCL all-gather of [B, 2D]; backward = reduce-scatter).  Default False = the reference's rank-local loss.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn

from .. import functional as Fn

f32 = torch.float32


# Alignment embeddings whose all-gather is already in flight (prefetch_alignment_gather): (w ptr, r ptr) -> (gathered, work, keep)
_gather_inflight: dict = {}
_comm_streams: dict = {}


def _comm_stream(device) -> "torch.cuda.Stream":
    key = torch.device(device).index or 0
    st = _comm_streams.get(key)
    if st is None:
        st = _comm_streams[key] = torch.cuda.Stream(device=device)
    return st


def prefetch_alignment_gather(wsi_emb: torch.Tensor, rna_emb: torch.Tensor, group=None) -> None:
    """Issue the global-batch InfoNCE all-gather of [wsi_alignment_emb | rna_alignment_emb] NOW, on a communication stream,
    instead of synchronously inside the loss (north_star: "the all-gather of projected embeddings ... overlapped ... on a side
    HIP stream").  MIRROR.forward calls this right after the two alignment heads; the retention decoder (a
    whole TransLayer) and the style head then run while the [B, 2D] message crosses xGMI.  clip_loss_terms picks the result up
    by the embeddings' storage; if nobody does (no gather configured), the entry is dropped at the next forward."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    x = torch.cat([wsi_emb.detach().float(), rna_emb.detach().float()], dim=1).contiguous()
    world = dist.get_world_size(group)
    out = torch.empty((world * x.shape[0], x.shape[1]), device=x.device, dtype=x.dtype)
    if x.is_cuda:
        comm = _comm_stream(x.device)
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            work = dist.all_gather_into_tensor(out, x, group=group, async_op=True)
        x.record_stream(comm)
        out.record_stream(comm)
    else:
        work = dist.all_gather_into_tensor(out, x, group=group, async_op=True)
    _gather_inflight.clear()                # one pair per forward
    _gather_inflight[(wsi_emb.data_ptr(), rna_emb.data_ptr())] = (out, work, x)


class _AllGatherCat(torch.autograd.Function):
    """all_gather along dim 0 with a gradient: backward reduce-scatters (sums) the slices back to their owners.
    pre = (gathered, work, _): the gather was issued earlier (prefetch_alignment_gather); only its completion is awaited here."""

    @staticmethod
    def forward(ctx, x, group=None, pre=None):
        x = x.contiguous()
        ctx.group = group
        world = dist.get_world_size(group)
        if pre is not None and tuple(pre[0].shape) == (world * x.shape[0],) + tuple(x.shape[1:]):
            out, work, _ = pre
            work.wait()                     # RCCL: the current stream waits for the collective; gloo: the host does
            if out.is_cuda:
                out.record_stream(torch.cuda.current_stream())   # allocated on the heads stream, read on this one from here on
            return out
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
        dist.all_gather_into_tensor(out, x, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        world = dist.get_world_size(ctx.group)
        out = torch.empty((g.shape[0] // world,) + tuple(g.shape[1:]), device=g.device, dtype=g.dtype)
        dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=ctx.group)
        return out, None, None


def clip_loss_terms(wsi, rna, logit_scale, gather: bool = False, group=None):
    """0.5 * (CE(s W R^T) + CE(s R W^T)) with labels on the (rank-shifted) diagonal; returns a 0-d tensor.
    `group`: the process group whose ranks are contrasted (None = the default group); gather size, label offset and the
    backward reduce-scatter all use it, so it must be the group the gradients are averaged over."""
    w, r = wsi.float(), rna.float()
    B = w.shape[0]
    off = 0
    w_all, r_all = w, r
    if gather and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        pre = _gather_inflight.pop((wsi.data_ptr(), rna.data_ptr()), None)     # issued by MIRROR.forward right after the heads
        both = _AllGatherCat.apply(torch.cat([w, r], dim=1), group, pre)       # one message: [P*B, 2D]
        D = w.shape[1]
        w_all, r_all = both[:, :D], both[:, D:]
        off = dist.get_rank(group) * B
    g_img = Fn.MatmulNTFn.apply(w, r_all)                            # [B, P*B]
    # rank-local loss: R W^T is the transpose of W R^T — one product (and one pair of gradient products) instead of two
    g_txt = g_img.t().contiguous() if w_all is w else Fn.MatmulNTFn.apply(r, w_all)
    li = Fn.CERowsFn.apply(g_img, logit_scale, 1.0, off, 0.5 / B, False)
    lt = Fn.CERowsFn.apply(g_txt, logit_scale, 1.0, off, 0.5 / B, False)
    return (li + lt).reshape(())


class ClipLoss(nn.Module):
    """losses/mirror_loss.py:16-52 (labels are implicit: the kernel indexes the diagonal, nothing to cache)."""

    def __init__(self, cache_labels: bool = False, gather_distributed: bool = False, process_group=None):
        super().__init__()
        self.cache_labels = cache_labels
        self.gather_distributed = gather_distributed
        self.process_group = process_group        # TrainEngine(process_group=...) sets it to its own group

    def forward(self, wsi_features, rna_features, logit_scale, output_dict: bool = False):
        total = clip_loss_terms(wsi_features, rna_features, logit_scale, self.gather_distributed, self.process_group)
        return {"contrastive_loss": total} if output_dict else total


class MIRRORLoss(nn.Module):
    def __init__(self, clip_loss_cache_labels=True, alignment_loss_weight=0.5, wsi_retention_loss_weight=0.1,
                 rna_retention_loss_weight=0.1, style_loss_weight=0.1, cluster_loss_weight=0.2,
                 gather_distributed: bool = False, process_group=None):
        super().__init__()
        self.clip_loss = ClipLoss(cache_labels=clip_loss_cache_labels, gather_distributed=gather_distributed,
                                  process_group=process_group)
        self.alignment_loss_weight = alignment_loss_weight
        self.wsi_retention_loss_weight = wsi_retention_loss_weight
        self.rna_retention_loss_weight = rna_retention_loss_weight
        self.style_loss_weight = style_loss_weight
        self.cluster_loss_weight = cluster_loss_weight

    def forward(self, wsi_alignment_emb, wsi_retention_emb, wsi_retention_target, wsi_mask, wsi_score, wsi_mu,
                wsi_logstd, rna_alignment_emb, rna_retention_emb, rna_retention_target, rna_mask, rna_score, rna_mu,
                rna_logstd, logit_scale):
        D = wsi_retention_emb.shape[-1]
        sw = self.style_loss_weight
        weights = (self.alignment_loss_weight, self.wsi_retention_loss_weight, self.rna_retention_loss_weight, sw, sw,
                   self.cluster_loss_weight)
        cl = self.clip_loss
        gathered = (cl.gather_distributed and dist.is_available() and dist.is_initialized()
                    and dist.get_world_size(cl.process_group) > 1)
        align = None if gathered else (wsi_alignment_emb, rna_alignment_emb, logit_scale)
        if Fn.loss_terms_fusable(align, (rna_retention_emb, rna_retention_target, rna_mask), (wsi_mu, wsi_logstd, rna_mu, rna_logstd),
                                 (wsi_score, rna_score)):
            # every term but the WSI retention MSE in one launch (and one in the backward): the ~28 tiny launches of the
            # composed form below sit back to back on the critical path between the forward and the backward of the step
            ext = cl(wsi_alignment_emb, rna_alignment_emb, logit_scale).reshape(()) if gathered else None
            wa, ra, sc = align if align is not None else (None, None, None)
            return Fn.MirrorLossTermsFn.apply(weights, wa, ra, sc, ext, wsi_retention_emb, wsi_retention_target, wsi_mask, D,
                                              getattr(wsi_retention_target, "_fan_token", None), rna_retention_emb,
                                              rna_retention_target, rna_mask, wsi_mu, wsi_logstd, rna_mu, rna_logstd,
                                              wsi_score, rna_score)
        alignment_loss = self.clip_loss(wsi_alignment_emb, rna_alignment_emb, logit_scale)
        wsi_retention_loss = Fn.masked_mse(wsi_retention_emb, wsi_retention_target, wsi_mask, D)
        rna_retention_loss = Fn.masked_mse(rna_retention_emb, rna_retention_target, rna_mask, 1)
        B = wsi_mu.shape[0]
        style_w = Fn.StyleKLFn.apply(wsi_mu, wsi_logstd, 0.5 / B)
        style_r = Fn.StyleKLFn.apply(rna_mu, rna_logstd, 0.5 / rna_mu.shape[0])
        cluster_loss = Fn.SymKLFn.apply(wsi_score, rna_score, 0.5 / wsi_score.shape[0])
        # losses/mirror_loss.py:121-127 as ONE kernel (and one in the backward) instead of ~20 scalar torch launches
        total_loss = Fn.WeightedSumFn.apply(
            weights, alignment_loss.reshape(()), wsi_retention_loss, rna_retention_loss, style_w.reshape(()), style_r.reshape(()),
            cluster_loss.reshape(()))
        style_loss = (style_w.detach() + style_r.detach()).reshape(())
        return (total_loss, alignment_loss.reshape(()), wsi_retention_loss, rna_retention_loss, style_loss,
                cluster_loss.reshape(()))
