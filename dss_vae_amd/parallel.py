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


class GradBuckets(object):
    """Bucketed, overlapped gradient averaging over one flat gradient buffer.

    params[i] owns flat[offsets[i] : offsets[i] + params[i].numel()].  Buckets are contiguous ranges of
    the flat buffer of about `bucket_elems` elements, walked from the END of the buffer (gradients
    arrive roughly in reverse registration order).  `mark_ready(i)` is called from a post-accumulate
    hook of parameter i; when every parameter of a bucket that can receive a gradient this step has
    reported, the bucket is all-reduced asynchronously.  `finish()` reduces whatever has not been sent
    (parameters that got no gradient still take part: their slice holds zeros on every rank) and waits.
    """

    def __init__(self, flat_grad, params, offsets, bucket_elems, group=None):
        self.flat, self.params, self.offsets, self.group = flat_grad, params, offsets, group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.backend = dist.get_backend(group) if self.world > 1 else None
        self.cuda = flat_grad.is_cuda
        self.comm = torch.cuda.Stream(device=flat_grad.device) if (self.cuda and self.world > 1) else None
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
            lo0, hi0, idx0 = self.buckets[-1]
            self.buckets[-1] = (0, hi0, idx0)  # alignment padding in front of the first parameter
        self.bucket_of = {}
        for b, (_, _, idx) in enumerate(self.buckets):
            for i in idx:
                self.bucket_of[i] = b
        self.reset()

    def reset(self):
        self.pending = [sum(1 for i in idx if self.params[i].requires_grad) for _, _, idx in self.buckets]
        self.sent = [False] * len(self.buckets)
        self.works = []

    def _launch(self, b):
        if self.sent[b] or self.world == 1:
            self.sent[b] = True
            return
        self.sent[b] = True
        lo, hi, _ = self.buckets[b]
        chunk = self.flat[lo:hi]
        if self.comm is not None:
            self.comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm):
                self._reduce(chunk)
        else:
            self._reduce(chunk)

    def _reduce(self, chunk):
        if self.backend == "nccl":
            self.works.append((dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None))
        else:
            self.works.append((dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True), chunk))

    def mark_ready(self, i):
        if self.world == 1:
            return
        b = self.bucket_of[i]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def finish(self):
        """all buckets reduced and visible to the current stream"""
        if self.world == 1:
            return
        for b in range(len(self.buckets)):
            self._launch(b)
        for work, chunk in self.works:
            work.wait()
            if chunk is not None:
                chunk.div_(self.world)
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)
        self.works = []
