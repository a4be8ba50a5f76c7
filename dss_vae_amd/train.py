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
        self.buckets_off = False  # GraphedTrainer: the all-reduce runs between two graphs, not from the hooks
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
            if self.buckets is not None and not self.buckets_off:
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
        if self.buckets is not None and not self.buckets_off:
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
        if p.grad is not None:
            # a second backward without zero_grad(): the slice already holds the first one's gradient, and a claiming
            # op would overwrite it and AccumulateGrad then double it; the copy path accumulates as torch does
            return None
        if self.buckets is not None and not self.buckets_off and self.buckets.already_sent(i):
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
            late = self.buckets is not None and not self.buckets_off and i in self.buckets.late
            if g.data_ptr() == view.data_ptr():
                self.adopted += 1
            elif not late:  # a late gradient was taken aside by intercept(): p.grad holds its zero stand-in
                view.copy_(g)
                self.copied += 1
            p.grad = view
        if self.buckets is not None and not self.buckets_off:
            self.buckets.mark_ready(i)

    def all_reduce_grads(self):
        """gradient average over RCCL (xGMI): buckets not yet sent by the backward hooks go now."""
        if self.buckets is not None:
            self.buckets.finish()

    def reduce_flat(self, fired, tag="main"):
        """the gradient average of a step whose backward ran without the hooks (a replayed graph): every bucket
        that holds a parameter of `fired`, in bucket order, then visible to the current stream"""
        b = self.buckets
        b.reset(tag)
        b.expect, b.known, b.fired = None, frozenset(fired), set(fired)
        b.finish()

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

    def step(self, reduce=True):
        if reduce:
            self.all_reduce_grads()
        self.sumsq.zero_()
        ops.sumsq(self.grad, self.sumsq)
        # the update is gated on device by the sticky error word of the persistent GRU launches: a step whose
        # recurrence gave up waiting (garbage gradients) is dropped, not applied; the host learns of it at its
        # next ops.check_persistent_errors()
        # (a timed-out launch is fatal for the run: the host's step counts advance although the device dropped the
        # update, and under data parallel the word is rank-local -- bin/train.py checks it every log_every steps and
        # raises; nothing continues training past it)
        err = ops.persistent_error_word(self.flat.device)
        dyn = ops.DYN
        for lo, hi, st in self._live_ranges():
            if dyn is None:
                ops.adam_step(self.flat[lo:hi], self.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], self.lr, self.betas[0],
                              self.betas[1], self.eps, st, self.sumsq if self.clip > 0.0 else None, self.clip, err)
            else:
                # under capture: lr and the bias corrections of this range are three per-step device floats, which
                # `write_hyper` refreshes in front of every replay
                members = [i for i in sorted(self.fired) if lo <= self.offsets[i] < hi]
                slot = dyn.slot("adam%d" % len(dyn.adam), 3)
                dyn.adam.append((slot, members))
                ops.adam_step_dyn(self.flat[lo:hi], self.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], dyn.ptr(slot),
                                  self.betas[0], self.betas[1], self.eps, self.sumsq if self.clip > 0.0 else None,
                                  self.clip, err)
        for i in self.fired:
            self.steps[i] += 1

    def write_hyper(self, dyn):
        """in front of a replay of a captured step: (lr, 1 - beta1^t, sqrt(1 - beta2^t)) of every Adam range of the
        graph into its per-step slots, and the step counts advanced as step() would.  -> False when the members of a
        range no longer share one step count (eager steps of another kind ran in between): the graph is stale."""
        steps = list(self.steps)
        plan = []
        for slot, members in dyn.adam:  # in the order the captured step() calls ran: a parameter may be in several
            if len(set(steps[i] for i in members)) != 1:
                return False
            st = steps[members[0]] + 1
            plan.append((slot, st))
            for i in members:
                steps[i] = st
        for slot, st in plan:
            dyn.set_float(slot, self.lr)
            dyn.set_float(slot + 1, 1.0 - self.betas[0] ** st)
            dyn.set_float(slot + 2, (1.0 - self.betas[1] ** st) ** 0.5)
        self.steps = steps
        return True

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
    # the caller gets numbers, not a graph: a loss dictionary that still referenced the step's autograd nodes would
    # keep every parameter's AccumulateGrad node (and the stream it was made on) alive into the next step
    return {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in ret.items()}


# Stream-capture mode of the step graphs.  HIP's default ("global") forbids potentially unsafe calls in EVERY thread
# while any stream captures -- and a process with an RCCL communicator has such a thread: ProcessGroupNCCL's watchdog
# polls the events of finished collectives with hipEventQuery, which invalidates a capture that happens to be open
# (hipErrorStreamCaptureInvalidated at the next launch; seen once in ~10 runs of the 1-rank RCCL probe).  The batch
# prefetch thread's copies are the same kind of call.  "thread_local" checks the capturing thread only; nothing another
# thread enqueues touches the capture stream (torch's streams are non-blocking: no implicit join with the NULL stream).
CAPTURE_MODE = os.environ.get("DSSVAE_CAPTURE_MODE", "thread_local")


class GraphedTrainer(object):
    """`train_iteration` behind a HIP graph: after `warm` eager steps of a shape signature the whole iteration
    (zero_grad, forward, backward, norm, clip, Adam -- and with `adv_train` the two discriminator steps in front) is
    captured ONCE on a side stream and from then on replayed with one graph launch per step, so the host's share of
    a step falls from ~270 Python/ctypes/autograd-driven launches to a handful (the batch copy, one launch that
    rewrites the per-step scalars, the graph).  What changes from step to step is read on the device
    (ops.StepScalars): dropout / word-dropout seeds, the annealed word-dropout rate, the KL weight, Adam's learning
    rate and bias corrections; latent noise comes from torch's graph-aware generator.

    A signature = batch shapes (+ non-pad row counts), the requires_grad flags at the step's start (EnhanceSyntaxVAE
    freezes shared embeddings during its first forward), `train_iter > warm_up`, training mode.  Steps whose signature
    has no graph yet (ragged real data: many shapes) run eagerly, as do data-parallel steps with unequal shards; at
    most `max_graphs` signatures are kept.  Injected noise (parity tests), the contract check of the fused CE and the
    per-kernel event timer need the eager path and switch the graphs off.

    Data parallel: the gradient average sits between backward and the optimiser, so a step is TWO graphs -- zero_grad +
    forward + backward, then norm + clip + Adam -- around eager all-reduces of the flat gradient buffer's live buckets
    (no overlap with backward in this mode: DSSVAE_GRAPH=0 keeps the hook-driven overlapped schedule instead)."""

    def __init__(self, model, optimizer, adv_syn=False, adv_sem=False, enabled=None, warm=2, max_graphs=4):
        self.model, self.opt, self.adv = model, optimizer, (bool(adv_syn), bool(adv_sem))
        if enabled is None:
            enabled = os.environ.get("DSSVAE_GRAPH", "1") != "0"
        self.enabled = bool(enabled) and getattr(model, "GRAPH_SAFE", False)
        self.warm, self.max_graphs = int(warm), int(max_graphs)
        self.seen, self.graphs, self.pool = {}, {}, None
        self.replays = self.eager = self.captures = 0
        # Every step this object runs -- eager or captured -- is enqueued on ONE stream of its own (forked from and
        # joined to the caller's current stream per call).  Autograd gives a parameter's AccumulateGrad node the
        # stream that was current when the node was made, and the node lives as long as any graph that used the
        # parameter (a loss dict the caller still holds is enough).  Were the eager steps on the default stream,
        # a capture on a side stream would find such nodes, the engine would make the NULL stream wait for the
        # capturing one, and HIP's stream capture does not survive that (a crash in hipStreamEndCapture).
        self.stream = torch.cuda.Stream(device=optimizer.flat.device) if self.enabled else None

    # ---- bookkeeping -----------------------------------------------------------------------------
    def _signature(self, batch, train_iter):
        m = self.model
        return (batch.signature(), tuple(p.requires_grad for p in self.opt.params), m.training,
                train_iter > getattr(m.args, "warm_up", 0), self.adv)

    def _graphable(self, batch, global_batch):
        if not self.enabled or ops.PROFILE is not None or ops.CHECK_UNIFORM_UPSTREAM or ops.FORCE_STEPWISE_GRU:
            return False
        if self.model.noise or not hasattr(batch, "signature"):
            return False
        world = getattr(self.opt, "world", 1)
        if world > 1 and global_batch is not None and len(batch) * world != global_batch:
            return False
        return True

    def _eager(self, batch, train_iter, global_batch):
        self.eager += 1
        if self.stream is None:
            return train_iteration(self.model, self.opt, batch, train_iter, self.adv[0], self.adv[1], global_batch)
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            ret = train_iteration(self.model, self.opt, batch, train_iter, self.adv[0], self.adv[1], global_batch)
        cur.wait_stream(self.stream)
        return ret

    def __call__(self, examples, train_iter, global_batch=None):
        batch = examples
        if not self._graphable(batch, global_batch):
            return self._eager(batch, train_iter, global_batch)
        key = self._signature(batch, train_iter)
        rec = self.graphs.get(key)
        if rec is None:
            n = self.seen[key] = self.seen.get(key, 0) + 1
            if n <= self.warm or len(self.graphs) >= self.max_graphs:
                return self._eager(batch, train_iter, global_batch)
            rec = self.graphs[key] = self._capture(batch, train_iter)
        return self._replay(key, rec, batch, train_iter, global_batch)

    # ---- capture ---------------------------------------------------------------------------------
    def _capture(self, batch, train_iter):
        model, opt = self.model, self.opt
        dev = opt.flat.device
        rec = dict(static=batch.clone(), dyn=ops.StepScalars(dev), graphs=[])
        split = opt.buckets is not None   # data parallel: backward | all-reduce | optimiser
        steps_before = list(opt.steps)
        flags = [p.requires_grad for p in opt.params]
        ops.persistent_error_word(dev)
        torch.cuda.synchronize()
        ops._PERSISTENT_DONE.clear()   # events of eager launches must not be waited for inside the capture
        ops.DYN = rec["dyn"]
        try:
            if not split:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.pool, stream=self.stream, capture_error_mode=CAPTURE_MODE):
                    rec["ret"] = train_iteration(model, opt, rec["static"], train_iter, self.adv[0], self.adv[1])
                rec["graphs"] = [g]
            else:
                if any(self.adv):
                    raise RuntimeError("GraphedTrainer: the 3-pass adversarial iteration under data parallel runs eagerly")
                g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                opt.buckets_off = True     # hooks only record which parameters fired; nothing is sent under capture
                with torch.cuda.graph(g1, pool=self.pool, stream=self.stream, capture_error_mode=CAPTURE_MODE):
                    opt.zero_grad("main")
                    rec["ret"] = model.get_loss(rec["static"], train_iter)
                    torch.mean(rec["ret"]["Loss"]).backward()
                if self.pool is None:
                    self.pool = g1.pool()
                with torch.cuda.graph(g2, pool=self.pool, stream=self.stream, capture_error_mode=CAPTURE_MODE):
                    opt.step(reduce=False)
                rec["graphs"] = [g1, g2]
                rec["fired"] = frozenset(opt.fired)
            if self.pool is None:
                self.pool = rec["graphs"][0].pool()
        finally:
            ops.DYN = None
            opt.buckets_off = False
            ops._PERSISTENT_DONE.clear()   # ... nor may eager launches wait for events recorded under capture
        # the capture ran the step's host logic but none of its kernels: undo the host-side effects a replay repeats
        opt.steps = steps_before
        rec["flag_changes"] = [(p, p.requires_grad) for p, f in zip(opt.params, flags) if p.requires_grad != f]
        for p, f in zip(opt.params, flags):
            p.requires_grad = f
        self.captures += 1
        return rec

    # ---- replay ----------------------------------------------------------------------------------
    def _replay(self, key, rec, batch, train_iter, global_batch):
        from .networks.rnn_base import GLOBAL_SEEDS
        model, opt = self.model, self.opt
        dyn = rec["dyn"]
        if not opt.write_hyper(dyn):
            del self.graphs[key]           # stale: re-captured after the next eager steps of this signature
            self.seen[key] = self.warm
            return self._eager(batch, train_iter, global_batch)
        sc = model.step_scalars(train_iter)
        for name, value in sc.items():
            if name in dyn.slots:
                dyn.set_float(dyn.slots[name], value)
        dyn.set_seed_base(GLOBAL_SEEDS.next())
        if batch is not rec["static"]:
            rec["static"].copy_from(batch)
        dyn.upload()
        graphs = rec["graphs"]
        graphs[0].replay()
        if len(graphs) == 2:
            opt.reduce_flat(rec["fired"])
            graphs[1].replay()
        # host-side state the eager step would have left behind
        for p, f in rec["flag_changes"]:
            p.requires_grad = f
        model.step_unk_rate = sc.get("unk_rate", 0.0)
        ret = rec["ret"]
        if "KL Weight" in ret:
            ret["KL Weight"] = sc["kl_weight"]
        self.replays += 1
        return ret
