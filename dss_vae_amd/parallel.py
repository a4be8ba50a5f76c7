"""Data-parallel pieces of the training step (the reference has none: SURVEY 8e).

The path shards by minibatch: rank r takes every world-th sentence of the length-sorted global batch
(so every rank sees the same length distribution), re-sorted by -len locally as packing needs.
Parameters are replicated.  The one exchange per step is a gradient average over RCCL (xGMI):
gradients live in ONE flat fp32 buffer, cut into a few large contiguous buckets (xGMI is a
point-to-point mesh, so a ring collective is per-link bound: few large all-reduces, not many small
ones); a bucket's all-reduce is launched on a side stream as soon as autograd has produced all of its
gradients, so it overlaps the rest of backward.  Every loss term is (sum over sentences) / B_local, so
with equal shards the gradient average equals the single-process gradient at the global batch.
"""
import torch
import torch.distributed as dist


def shard_examples(examples, rank, world):
    """strided deal of a (length-sorted) global batch; each shard stays sorted by -len(src)."""
    part = list(examples[rank::world])
    part.sort(key=lambda e: -len(e.src))
    return part


def world_size(group=None):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


class GradBuckets(object):
    """Bucketed, overlapped gradient averaging over one flat gradient buffer.

    params[i] owns flat[offsets[i] : offsets[i] + params[i].numel()].  Buckets are contiguous ranges of
    the flat buffer of about `bucket_elems` elements, numbered from the END of the buffer (gradients
    arrive roughly in reverse registration order).  `mark_ready(i)` is called from a post-accumulate
    hook of parameter i.

    Which parameters receive a gradient is a property of the autograd graph of the step, not of
    requires_grad at any one moment (EnhanceSyntaxVAE freezes its adversaries half-way through forward),
    so it is LEARNED: the first step of a kind (`tag`) reduces everything at `finish()` and records
    the set of parameters whose hooks fired; the next step of that kind expects the same set and
    all-reduces a bucket asynchronously as soon as all of its expected parameters have reported -- but
    only once every lower-numbered bucket has been sent, so all ranks issue their collectives in ONE
    fixed order whatever order autograd fires the hooks in (RCCL requires identical issue order on
    every rank; the fired sets are identical on all ranks because the graphs are).  A gradient that
    arrives for a bucket that is already on the wire (or was skipped) -- the graph of this step differs from
    the last one of its kind -- is taken aside by `intercept()` (a tensor hook, before accumulation),
    averaged on its own at `finish()` and added to the bucket's averaged slice; the next step of the kind
    expects it.  `finish()` sends whatever has not been sent and makes the averaged gradients visible to
    the current stream.  A bucket none of whose parameters got a gradient in the last step of this kind is
    not sent at all (its slice is zero on every rank).
    """

    def __init__(self, flat_grad, params, offsets, bucket_elems, group=None, overlap=True):
        self.flat, self.params, self.offsets, self.group = flat_grad, params, offsets, group
        self.world = world_size(group)
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.overlap = overlap
        self.comm = torch.cuda.Stream(device=flat_grad.device) if flat_grad.is_cuda else None
        # bucket boundaries: whole parameters, contiguous, from the end
        self.buckets = []  # (lo, hi, [param indices])
        hi, cur, lo = flat_grad.numel(), [], flat_grad.numel()
        for i in range(len(params) - 1, -1, -1):
            cur.append(i)
            lo = offsets[i]
            if hi - lo >= bucket_elems:
                self.buckets.append((lo, hi, cur))
                hi, cur = lo, []
        if cur:
            self.buckets.append((lo, hi, cur))
        if self.buckets:
            _, hi0, idx0 = self.buckets[-1]
            self.buckets[-1] = (0, hi0, idx0)  # alignment padding in front of the first parameter
        self.bucket_of = {}
        for b, (_, _, idx) in enumerate(self.buckets):
            for i in idx:
                self.bucket_of[i] = b
        self.expected = {}  # tag -> frozenset of parameter indices that fired in the last step of that kind
        self.known, self.skipped = None, 0
        self.sent_early = 0  # buckets sent from hooks in the last step (observability / tests)
        self.reset()

    def reset(self, tag="main"):
        self.tag = tag
        self.known = self.expected.get(tag)            # parameters that fired in the last step of this kind
        self.expect = self.known if self.overlap else None
        self.skipped = 0   # buckets not sent at all in this step (observability / tests)
        self.fired = set()
        if self.expect is not None:
            self.pending = [sum(1 for i in idx if i in self.expect) for _, _, idx in self.buckets]
        self.next = 0      # buckets [0, next) have been sent
        self.works = []
        self.sent_early = 0
        self.late = {}     # parameter index -> gradient that arrived after its bucket was dealt with

    def _send(self, b):
        lo, hi, idx = self.buckets[b]
        if self.known is not None and not any(i in self.known or i in self.fired for i in idx):
            # no parameter of this bucket gets a gradient in this kind of step (frozen adversaries in a main
            # pass -- 60 % of the buffer at V = 30k -- or everything but the adversaries in a discriminator
            # pass): its slice is zero on every rank, the average is zero, nothing to send.  The fired sets
            # are identical on all ranks, so all ranks skip the same buckets; a gradient that does show up
            # for a skipped bucket raises in mark_ready like one for a bucket already on the wire.
            self.skipped += 1
            return
        chunk = self.flat[lo:hi]
        if self.comm is not None:
            self.comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm):
                self._reduce(chunk)
        else:
            self._reduce(chunk)

    def _reduce(self, chunk):
        if self.backend == "nccl":  # RCCL: the product path
            self.works.append((dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None))
        elif chunk.is_cuda:
            # gloo with device buffers is only a rehearsal of the N>1 schedule on a one-GPU box: staged
            # through the host synchronously (gloo's own asynchronous device path stalls under HIP
            # when many works are outstanding)
            host = chunk.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            chunk.copy_(host.div_(self.world))
        else:
            self.works.append((dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True), chunk))

    def already_sent(self, i):
        """has parameter i's bucket been dealt with (sent or skipped) in this step?  Its slice of the flat
        buffer must then not be written directly: the gradient has to go through intercept()."""
        return self.expect is not None and self.bucket_of[i] < self.next

    def intercept(self, i, g):
        """tensor hook of parameter i (runs before the gradient is accumulated): a gradient for a bucket that
        has already been sent or skipped is kept aside and replaced by zeros, so the bucket's averaged slice
        stays intact; `finish()` averages it separately.  Returns the replacement or None (no change)."""
        if g is None or self.expect is None or self.bucket_of[i] >= self.next:
            return None
        held = self.late.get(i)
        self.late[i] = g.detach().clone() if held is None else held.add_(g.detach())
        return torch.zeros_like(g)

    def mark_ready(self, i):
        self.fired.add(i)
        if self.expect is None:
            return  # everything goes at finish(); _send() keeps every bucket that fired this step
        b = self.bucket_of[i]
        if b < self.next:
            if i in self.late:
                return
            raise RuntimeError(
                "gradient of parameter %d arrived after its bucket was all-reduced and no tensor hook routed it "
                "through GradBuckets.intercept(): register it (FlatAdam does), set DSSVAE_DP_OVERLAP=0 or give "
                "the steps different tags (tag %r)" % (i, self.tag))
        if i in self.expect:
            self.pending[b] -= 1
            while self.next < len(self.buckets) and self.pending[self.next] <= 0:
                self._send(self.next)
                self.next += 1
                self.sent_early += 1

    def finish(self):
        """all buckets reduced and visible to the current stream"""
        while self.next < len(self.buckets):
            self._send(self.next)
            self.next += 1
        for work, chunk in self.works:
            work.wait()
            if chunk is not None:
                if self.comm is not None:
                    with torch.cuda.stream(self.comm):
                        chunk.div_(self.world)
                else:
                    chunk.div_(self.world)
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)
        self.works = []
        for i in sorted(self.late):  # same graph on every rank => same late set, reduced in index order
            g = self.late[i]
            if self.backend == "nccl":
                dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group)
            elif g.is_cuda:
                host = g.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                g.copy_(host.div_(self.world))
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
                g.div_(self.world)
            off = self.offsets[i]
            self.flat[off:off + g.numel()].add_(g.reshape(-1))
        self.late = {}
        self.expected[self.tag] = frozenset(self.fired)
