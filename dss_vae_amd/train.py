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
import torch
import torch.distributed as dist

from . import ops


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
                 process_group=None):
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
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.bucket = max(1, int(bucket_mb * (1 << 20) // 4))
        self.comm = torch.cuda.Stream(device=dev) if self.world > 1 else None

    def zero_grad(self):
        self.grad.zero_()
        for p in self.params:  # autograd may have replaced a view (e.g. after requires_grad toggling)
            if p.grad is None or p.grad.data_ptr() < self.grad.data_ptr() or \
                    p.grad.data_ptr() >= self.grad.data_ptr() + self.grad.numel() * 4:
                self._rebind()
                break

    def _rebind(self):
        off = 0
        for p in self.params:
            p.grad = self.grad[off:off + p.numel()].view(p.shape)
            off += (p.numel() + 3) // 4 * 4

    def all_reduce_grads(self):
        """bucketed gradient averaging over RCCL (xGMI) on a side stream."""
        if self.world == 1:
            return
        self.comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm):
            works = []
            for lo in range(0, self.grad.numel(), self.bucket):
                chunk = self.grad[lo:lo + self.bucket]
                if dist.get_backend(self.group) == "nccl":
                    works.append(dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
                else:
                    works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            for w in works:
                w.wait()
            if dist.get_backend(self.group) != "nccl":
                self.grad.div_(self.world)
        torch.cuda.current_stream().wait_stream(self.comm)

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
