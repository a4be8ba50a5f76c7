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
import warnings

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


# heads that run on a side stream (decoder/rnn_decoder.py: score_many) hand their parameter gradients to
# AccumulateGrad nodes of the default stream: the engine orders the two streams itself and says so once per run
warnings.filterwarnings("ignore", message="The AccumulateGrad node's stream does not match")


class FlatAdam(object):
    """Adam + clip_grad_norm_ over flat parameter / gradient buffers, optionally data-parallel.

    torch.optim.Adam semantics are kept where the reference's loop depends on them: a parameter that
    received NO gradient in a step (the frozen adversaries in the main pass, everything but the
    adversaries in a discriminator pass) is skipped entirely -- no moment decay, no update, and its own
    step count (Adam's bias correction is per parameter) does not advance.  Which parameters received a
    gradient is recorded by tensor hooks; the update then runs over the contiguous ranges of
    the flat buffer that are live (a handful of launches)."""

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
        self.offsets, off = [], 0
        self.views = []   # parameter i's slice of the flat gradient buffer
        self._by_ptr = {}
        for i, (p, n) in enumerate(zip(params, sizes)):
            view = self.flat[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            self.views.append(self.grad[off:off + p.numel()].view(p.shape))
            self._by_ptr[p.data_ptr()] = i
            p.grad = None
            self.offsets.append(off)
            off += n
        self.claimed = set()
        self.adopted = 0   # gradients that autograd adopted in place (observability / tests)
        self.copied = 0    # gradients produced outside the flat buffer and copied in
        self.sizes = sizes
        self.params, self.betas, self.eps, self.clip = params, betas, eps, clip_grad
        self.param_groups = [{"lr": lr}]  # the reference's lr decay writes optimizer.param_groups[0]['lr']
        self.steps = [0] * len(params)    # Adam's per-parameter step count
        self.fired = set()
        self.sumsq = torch.zeros(1, device=dev)
        self.world = world_size(process_group) if data_parallel else 1
        self.buckets = None
        force = os.environ.get("DSSVAE_DP_FORCE", "0") == "1" and data_parallel and world_size(process_group) == 1 \
            and torch.distributed.is_initialized()
        if self.world > 1 or force:  # force: rehearse the whole DP schedule on a 1-rank group
            overlap = os.environ.get("DSSVAE_DP_OVERLAP", "1") != "0"
            self.buckets = GradBuckets(self.grad, params, self.offsets, max(1, int(bucket_mb * (1 << 20) // 4)),
                                       group=process_group, overlap=overlap)
        for i, p in enumerate(params):
            # a tensor hook sees the arriving gradient: None when it is undefined (the parameter's consumers
            # got no gradient this step); a leaf frozen after its use still gets the call but accumulates
            # nothing.  (A post-accumulate hook fires in all of these cases alike.)
            p.register_hook(lambda g, i=i: self._on_arrive(i, g))
            p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_accumulated(i))
        import weakref
        self._self_ref = weakref.ref(self)
        ops.GRAD_SINK = self._self_ref  # the newest optimiser takes the gradients of its parameters (ops._claim)

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @property
    def t(self):
        """number of optimisation steps taken by the most-updated parameter"""
        return max(self.steps)

    def _on_arrive(self, i, g):
        if g is not None and self.params[i].requires_grad:
            self.fired.add(i)
            if self.buckets is not None:
                return self.buckets.intercept(i, g)  # None unless the gradient is late for its bucket
        return None

    def zero_grad(self, tag="main"):
        """tag: kind of step that follows (steps of one kind share an autograd graph shape; the
        data-parallel bucket schedule is learned per kind).  Every p.grad becomes None: a backward that
        fills a parameter's slice of the flat buffer itself (ops._claim) hands that slice to autograd, which
        adopts it as p.grad without an accumulate kernel; anything else is copied in (_on_accumulated)."""
        self.grad.zero_()
        self.fired = set()
        self.claimed = set()
        ops.GRAD_SINK = self._self_ref
        for p in self.params:
            p.grad = None
        if self.buckets is not None:
            self.buckets.reset(tag)

    def claim(self, t):
        """ops._claim: the slice of the flat gradient buffer that belongs to parameter tensor `t` (zero at this
        point), or None -- not one of this optimiser's parameters, frozen by now (its gradient is to be
        dropped, as AccumulateGrad does), already handed out in this step (a second use of the parameter:
        autograd sums the contributions), or its data-parallel bucket is already on the wire."""
        i = self._by_ptr.get(t.data_ptr())
        if i is None or i in self.claimed:
            return None
        p = self.params[i]
        if t.shape != p.shape or not t.is_contiguous() or not p.requires_grad:
            return None
        if self.buckets is not None and self.buckets.already_sent(i):
            return None
        self.claimed.add(i)
        return self.views[i].view(p.shape)  # a tensor of its own: autograd adopts a gradient nobody else holds

    def _on_accumulated(self, i):
        """post-accumulate hook: parameter i's gradient is complete.  Bring it home if it was produced outside
        the flat buffer (a torch op's gradient, or contributions summed by the engine), then report it to the
        data-parallel schedule."""
        p = self.params[i]
        g, view = p.grad, self.views[i]
        if g is not None:
            late = self.buckets is not None and i in self.buckets.late
            if g.data_ptr() == view.data_ptr():
                self.adopted += 1
            elif not late:  # a late gradient was taken aside by intercept(): p.grad holds its zero stand-in
                view.copy_(g)
                self.copied += 1
            p.grad = view
        if self.buckets is not None:
            self.buckets.mark_ready(i)

    def all_reduce_grads(self):
        """gradient average over RCCL (xGMI): buckets not yet sent by the backward hooks go now."""
        if self.buckets is not None:
            self.buckets.finish()

    def _live_ranges(self):
        """contiguous [lo, hi) element ranges of parameters that got a gradient, split where the
        per-parameter step counts differ -> [(lo, hi, step_after_this_update)]"""
        out = []
        for i in sorted(self.fired):
            lo, hi, st = self.offsets[i], self.offsets[i] + self.sizes[i], self.steps[i] + 1
            if out and out[-1][1] == lo and out[-1][2] == st:
                out[-1] = (out[-1][0], hi, st)
            else:
                out.append((lo, hi, st))
        return out

    def step(self):
        self.all_reduce_grads()
        self.sumsq.zero_()
        ops.sumsq(self.grad, self.sumsq)
        # the update is gated on device by the sticky error word of the persistent GRU launches: a step whose
        # recurrence gave up waiting (garbage gradients) is dropped, not applied; the host learns of it at its
        # next ops.check_persistent_errors()
        err = ops.persistent_error_word(self.flat.device)
        for lo, hi, st in self._live_ranges():
            ops.adam_step(self.flat[lo:hi], self.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], self.lr, self.betas[0],
                          self.betas[1], self.eps, st, self.sumsq if self.clip > 0.0 else None, self.clip, err)
        for i in self.fired:
            self.steps[i] += 1

    def grad_norm(self):
        """global gradient norm of the LAST step (device sync; for logging / tests)."""
        return float(self.sumsq.sqrt().item())

    # ---- checkpoint (the reference saves optimizer.state_dict() next to model.bin) ---------------
    def state_dict(self):
        return {"lr": self.lr, "steps": list(self.steps), "exp_avg": self.m.clone(), "exp_avg_sq": self.v.clone()}

    def load_state_dict(self, sd):
        self.param_groups[0]["lr"] = sd["lr"]
        self.steps = list(sd["steps"])
        self.m.copy_(sd["exp_avg"])
        self.v.copy_(sd["exp_avg_sq"])


def train_iteration(model, optimizer, examples, train_iter, adv_syn=False, adv_sem=False, global_batch=None):
    """universal_training + the main step (bin/train.py:90-126): with `adv_train`, a discriminator step on
    'dis syn' and one on 'dis sem' (each its own get_loss(is_dis=True) / zero_grad / backward / clip /
    Adam) precede the main pass.

    Data parallel (optimizer.world > 1): the ranks' gradients are AVERAGED, so each pass scales its loss
    such that the average equals the single-process gradient at the global batch: the discriminator
    losses are plain sums over sentences (x world), the main loss is a sum / B_local (x B_local * world /
    B_global; 1 for equal shards; `global_batch` = number of sentences over all ranks, None = equal shards)."""
    world = getattr(optimizer, "world", 1)
    for key, tag, on in (("dis syn", "dis_syn", adv_syn), ("dis sem", "dis_sem", adv_sem)):
        if on:
            ret = model.get_loss(examples=examples, train_iter=train_iter, is_dis=True)
            optimizer.zero_grad(tag)
            (ret[key] * float(world) if world > 1 else ret[key]).backward()
            optimizer.step()
    return train_step(model, optimizer, examples, train_iter, global_batch)


def train_step(model, optimizer, examples, train_iter, global_batch=None):
    """one main-pass iteration; returns the loss dict of get_loss."""
    optimizer.zero_grad("main")
    ret = model.get_loss(examples, train_iter)
    loss = torch.mean(ret["Loss"])
    world = getattr(optimizer, "world", 1)
    if world > 1 and global_batch is not None and len(examples) * world != global_batch:
        loss = loss * (len(examples) * world / float(global_batch))  # unequal shards stay exact
    loss.backward()
    optimizer.step()
    return ret
