"""Optimisation step of the hot path (reference: dss_vae/bin/train.py:84-88,119-126 and
utils/model_utils.py:187-196): zero_grad -> get_loss -> mean(Loss).backward() -> global-norm clip
5.0 over ALL parameters -> Adam(lr, betas=(0.9, 0.995)).

MI355X-first layout: every parameter is a view into ONE flat fp32 buffer and every gradient a view
into ONE flat gradient buffer, so that
  * data-parallel training is a handful of large RCCL all-reduces over contiguous memory
    (bucketed, issued on a side stream; xGMI is per-link bound, so few big collectives win),
  * grad-norm is one reduction kernel and Adam one elementwise kernel over the flat buffers
    (HBM-bound: 16 B read + 12 B written per parameter), with the clip coefficient read on device
    (no host sync in the step).
"""
import os

import torch

from . import ops
from .parallel import GradBuckets, world_size


def build_model_params_(model, seed=None, bound=0.08):
    """build_model's re-initialisation: every parameter ~ U(-0.08, 0.08) (model_utils.py:189-190)."""
    g = None
    if seed is not None:
        g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.empty(p.shape).uniform_(-bound, bound, generator=g).to(p.device))
    return model


class FlatAdam(object):
    """Adam + clip_grad_norm_ over flat parameter / gradient buffers, optionally data-parallel."""

    def __init__(self, model, lr=0.0005, betas=(0.9, 0.995), eps=1e-8, clip_grad=5.0, bucket_mb=64,
                 process_group=None, data_parallel=True):
        params = [p for p in model.parameters()]
        assert params and all(p.is_cuda and p.dtype == torch.float32 for p in params), \
            "FlatAdam needs the model on the MI355X (model.cuda()) in fp32"
        dev = params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]  # keep every view 16-B aligned
        total = sum(sizes)
        self.flat = torch.zeros(total, device=dev)
        self.grad = torch.zeros(total, device=dev)
        self.m = torch.zeros(total, device=dev)
        self.v = torch.zeros(total, device=dev)
        off = 0
        for p, n in zip(params, sizes):
            view = self.flat[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[off:off + p.numel()].view(p.shape)
            off += n
        self.params, self.lr, self.betas, self.eps, self.clip = params, lr, betas, eps, clip_grad
        self.t = 0
        self.sumsq = torch.zeros(1, device=dev)
        self.world = world_size(process_group) if data_parallel else 1
        self.buckets = None
        force = os.environ.get("DSSVAE_DP_FORCE", "0") == "1" and data_parallel and world_size(process_group) == 1 \
            and torch.distributed.is_initialized()
        if self.world > 1 or force:  # force: rehearse the whole DP schedule on a 1-rank group
            offsets, o = [], 0
            for n in sizes:
                offsets.append(o)
                o += n
            overlap = os.environ.get("DSSVAE_DP_OVERLAP", "1") != "0"
            self.buckets = GradBuckets(self.grad, params, offsets, max(1, int(bucket_mb * (1 << 20) // 4)),
                                       group=process_group, overlap=overlap)
            for i, p in enumerate(params):
                p.register_post_accumulate_grad_hook(lambda _p, i=i: self.buckets.mark_ready(i))

    def zero_grad(self, tag="main"):
        """tag: kind of step that follows (steps of one kind share an autograd graph shape; the
        data-parallel bucket schedule is learned per kind)."""
        self.grad.zero_()
        for p in self.params:  # autograd may have replaced a view (e.g. after requires_grad toggling)
            if p.grad is None or p.grad.data_ptr() < self.grad.data_ptr() or \
                    p.grad.data_ptr() >= self.grad.data_ptr() + self.grad.numel() * 4:
                self._rebind()
                break
        if self.buckets is not None:
            self.buckets.reset(tag)

    def _rebind(self):
        off = 0
        for p in self.params:
            p.grad = self.grad[off:off + p.numel()].view(p.shape)
            off += (p.numel() + 3) // 4 * 4

    def all_reduce_grads(self):
        """gradient average over RCCL (xGMI): buckets not yet sent by the backward hooks go now."""
        if self.buckets is not None:
            self.buckets.finish()

    def step(self):
        self.all_reduce_grads()
        self.t += 1
        self.sumsq.zero_()
        ops.sumsq(self.grad, self.sumsq)
        ops.adam_step(self.flat, self.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.t,
                      self.sumsq if self.clip > 0.0 else None, self.clip)

    def grad_norm(self):
        """global gradient norm of the LAST step (device sync; for logging / tests)."""
        return float(self.sumsq.sqrt().item())


def train_step(model, optimizer, examples, train_iter):
    """one main-pass iteration; returns the loss dict of get_loss."""
    optimizer.zero_grad()
    ret = model.get_loss(examples, train_iter)
    torch.mean(ret["Loss"]).backward()
    optimizer.step()
    return ret
