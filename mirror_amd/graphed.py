"""HIP-graph replay of a parameter-only branch of the model inside an EAGER training step.

The multi-GPU step cannot be replayed as one HIP graph (the RCCL bucket all-reduces are issued from autograd hooks and the
global-batch InfoNCE gathers in the middle of the loss), and it is launch-bound on the host: ~556 launches at ~20 us of
Python + HIP launch each against ~10 ms of GPU time.  The RNA branch (train_mirror.py's `rna_encoder`: embedding MLP, 6
pre-norm blocks, alignment / retention heads — models/mirror.py:155-289, :483-520) is ~230 of those launches for 0.06 % of
the FLOPs and contains no collective, so its forward and its backward are captured ONCE as two HIP graphs and replayed from
an autograd node: inputs are copied into static buffers, weight gradients accumulate straight into the engine's gradient
arena exactly as in the eager path (the captured kernels hold the arena addresses), and the engine is told which
parameters are complete so that the bucket all-reduces start as early as before.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
from torch.autograd import Function

from . import functional as Fn


class GraphedBranch:
    def __init__(self, fn: Callable, inputs: Sequence[torch.Tensor], anchor: torch.Tensor, engine):
        """fn(*inputs) -> tuple of tensors; `anchor`: any parameter of the branch (gives the autograd node an input that
        requires grad); `engine`: the TrainEngine (gradient sink + bucket bookkeeping)."""
        self.engine, self.anchor = engine, anchor
        self.static_in = [t.detach().clone() for t in inputs]
        st = Fn._dropout_state
        host_off = st["offset"]
        st["offset"] = 0                 # the branch is the first consumer of the step's dropout offsets
        Fn._res_grads.clear()
        torch.cuda.synchronize()
        self.g_f = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_f, capture_error_mode="thread_local"):
            outs = fn(*self.static_in)
        self.drop_n = st["offset"]
        st["offset"] = host_off
        self.outs = tuple(outs)
        self.req = [i for i, o in enumerate(self.outs) if o.requires_grad]
        self.static_g = [torch.zeros_like(self.outs[i]) for i in self.req]
        self.g_b = torch.cuda.CUDAGraph()
        engine._capturing = []
        Fn.set_grad_sink(engine)
        try:
            with torch.cuda.graph(self.g_b, pool=self.g_f.pool(), capture_error_mode="thread_local"):
                torch.autograd.backward([self.outs[i] for i in self.req], self.static_g)
        finally:
            Fn.set_grad_sink(None)
            touched, engine._capturing = engine._capturing, None
        Fn._res_grads.clear()
        seen, self.params = set(), []
        for i in touched:               # parameters whose gradient the backward graph completes, in completion order
            if i not in seen:
                seen.add(i)
                self.params.append(i)
        torch.cuda.synchronize()

    def matches(self, inputs: Sequence[torch.Tensor]) -> bool:
        return len(inputs) == len(self.static_in) and all(
            a.shape == b.shape and a.dtype == b.dtype and a.device == b.device for a, b in zip(inputs, self.static_in))

    def __call__(self, *inputs: torch.Tensor):
        self.replays = getattr(self, "replays", 0) + 1       # (tests: the branch must actually be replayed, not only recorded)
        return _GraphedFn.apply(self, self.anchor, *inputs)


class _GraphedFn(Function):
    @staticmethod
    def forward(ctx, br: GraphedBranch, anchor, *inputs):
        for s, t in zip(br.static_in, inputs):
            s.copy_(t, non_blocking=True)
        br.g_f.replay()
        Fn._dropout_state["offset"] += br.drop_n       # the eager rest of the step continues behind the branch's offsets
        ctx.br, ctx.n_in = br, len(inputs)
        outs = tuple(o.detach() for o in br.outs)
        ctx.mark_non_differentiable(*[o for i, o in enumerate(outs) if i not in br.req])
        return outs

    @staticmethod
    def backward(ctx, *grads):
        br = ctx.br
        for k, i in enumerate(br.req):
            if grads[i] is None:
                br.static_g[k].zero_()
            else:
                br.static_g[k].copy_(grads[i], non_blocking=True)
        br.g_b.replay()
        br.engine._branch_done(br.params)
        return (None, None) + (None,) * ctx.n_in
