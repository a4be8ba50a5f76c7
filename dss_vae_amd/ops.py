"""autograd wrappers over the C-ABI (include/dssvae_hip.h).  Every op launches hand-written HIP
kernels on the current stream; torch is used for memory and tiny glue only.  No CPU path."""
import ctypes as C

import torch
from torch.autograd import Function

from . import lib as L

_CUS = None


class KernelTimer(object):
    """HIP-event spans around single launches on the stream they are launched on (torch's current
    stream).  bench.py installs one as ``ops.PROFILE`` for its timed region to get each kernel's
    average launch duration live; with ``PROFILE = None`` (default) nothing is recorded."""

    def __init__(self):
        self.spans = []  # (kernel name, start event, end event, algorithmic flops, algorithmic bytes, timesteps)

    def summary(self):
        """-> {kernel: dict(launches, total_ms, avg_ms, flops, bytes)}; call after a device sync."""
        out = {}
        for name, e0, e1, fl, by, st, _detail, launches in self.spans:
            d = out.setdefault(name, dict(launches=0, total_ms=0.0, flops=0.0, bytes=0.0, timesteps=0))
            d["launches"] += launches
            d["total_ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
            d["timesteps"] += st
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / d["launches"]
        return out

    def by_shape(self):
        """-> [(kernel, shape string, launches, total_ms, TFLOP/s)] sorted by time (development aid)."""
        out = {}
        for name, e0, e1, fl, by, st, detail, _launches in self.spans:
            d = out.setdefault((name, detail), [0, 0.0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
            d[2] += fl
        rows = [(k[0], k[1], v[0], v[1], v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0) for k, v in out.items()]
        return sorted(rows, key=lambda r: -r[3])


PROFILE = None


class _Span(object):
    __slots__ = ("name", "flops", "bytes", "e0", "steps", "detail", "launches")

    def __init__(self, name, flops, nbytes, steps=0, detail=None):
        self.name, self.flops, self.bytes, self.steps, self.detail = name, flops, nbytes, steps, detail
        self.launches = 1  # launches of the named kernel inside the span (a store GEMM may run as two: set by the caller)

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILE.spans.append((self.name, self.e0, e1, float(self.flops), float(self.bytes), int(self.steps),
                                  self.detail, int(self.launches)))
        return False


# ----------------------------------------------------------------------------------------------
# Per-step scalars read on the device (include/dssvae_hip.h: dssvae_set_words and the *_dyn entry points).
# While a training step is being CAPTURED into a HIP graph (train.GraphedTrainer) `DYN` is the graph's StepScalars:
# every value that changes from step to step is then a slot of its device buffer instead of a host scalar baked into
# a kernel argument, and one `upload()` launch in front of each replay rewrites the slots.  Outside a capture `DYN`
# is None and every op passes its host values as before (bit-identical to the plain entry points).
# ----------------------------------------------------------------------------------------------
class Dyn(float):
    """a python float that also names a slot of the capturing graph's StepScalars: ops that can read the value on the
    device take the slot's address, everything else sees the number"""

    def __new__(cls, value, slot):
        obj = float.__new__(cls, value)
        obj.slot = slot
        return obj


class StepScalars(object):
    SEED_WORDS = 2                       # words 0-1: the 64-bit base of the step's dropout / word-dropout seeds
    SEED_STRIDE = 0xD1B54A32D192ED03     # k-th seed of a captured step = base + k * SEED_STRIDE (mod 2^64)

    def __init__(self, device):
        self.buf = torch.zeros(L.STEP_WORDS, device=device, dtype=torch.int32)
        self.f32 = self.buf.view(torch.float32)
        self.host = (C.c_uint32 * L.STEP_WORDS)()
        self.slots = {}
        self.n = self.SEED_WORDS
        self.seed_k = 0
        self.adam = []       # train.FlatAdam.step: (slot, parameter indices) per Adam range of the captured step
        self.keepalive = []  # events recorded under capture: destroying one before the capture ends leaves the
        #                      runtime's capture bookkeeping with a dangling pointer (a crash in hipStreamEndCapture)

    def slot(self, name, words=1):
        if name not in self.slots:
            if self.n + words > L.STEP_WORDS:
                raise RuntimeError("StepScalars: more than %d per-step words" % L.STEP_WORDS)
            self.slots[name] = self.n
            self.n += words
        return self.slots[name]

    def ptr(self, slot):
        return C.c_void_p(self.buf.data_ptr() + 4 * slot)

    def set_float(self, slot, value):
        self.host[slot] = C.c_uint32.from_buffer_copy(C.c_float(value)).value

    def set_seed_base(self, base):
        base &= 2 ** 64 - 1
        self.host[0], self.host[1] = base & 0xFFFFFFFF, base >> 32

    def next_seed(self):
        """-> (seed offset of the next RNG use of the captured step, device pointer to the base)"""
        self.seed_k += 1
        return (self.seed_k * self.SEED_STRIDE) & (2 ** 64 - 1), self.ptr(0)

    def value(self, name, value):
        slot = self.slot(name)
        self.set_float(slot, value)
        return Dyn(value, slot)

    def tensor(self, d):
        """0-d float32 view of a Dyn's slot, for torch ops that take the value as a tensor"""
        return self.f32[d.slot]

    def upload(self):
        L.check(L.load().dssvae_set_words(L.ptr(self.buf), self.host, self.n, L.stream()), "dssvae_set_words")


DYN = None


def _event():
    """an event for cross-stream ordering; under capture it lives as long as the graph's record"""
    ev = torch.cuda.Event()
    if DYN is not None:
        DYN.keepalive.append(ev)
    return ev


def dyn_value(name, value):
    """`value` as is, or -- while a step is being captured -- tagged with its per-step slot"""
    return value if DYN is None else DYN.value(name, float(value))


def dyn_tensor(v):
    """a scalar for torch arithmetic: the number itself, or the device view of its per-step slot"""
    return DYN.tensor(v) if (DYN is not None and isinstance(v, Dyn)) else v


def _seed_args(seed):
    if DYN is None:
        return int(seed) & (2 ** 64 - 1), None
    return DYN.next_seed()


def n_cus():
    global _CUS
    if _CUS is None:
        n = C.c_int(0)
        L.check(L.load().dssvae_device_cus(C.byref(n)), "dssvae_device_cus")
        _CUS = n.value
    return _CUS


# Gradient sink: an optimiser that keeps all gradients in one flat buffer (train.FlatAdam) installs itself here.
# A backward that produces the gradient of a PARAMETER asks `_claim(param)` for that parameter's slice of the
# flat buffer, computes straight into it and returns the slice as the gradient: autograd's AccumulateGrad then
# adopts the tensor (p.grad is None at that point) instead of launching an add kernel per parameter.
GRAD_SINK = None  # weakref.ref to the optimiser


def _claim(t, need=True):
    if not need or GRAD_SINK is None or t is None:
        return None
    sink = GRAD_SINK()
    return None if sink is None else sink.claim(t)


def _f32(t):
    if t.dtype != torch.float32:
        raise RuntimeError("expected float32, got %s" % t.dtype)
    return t


def _rows(t):
    """2-D view [R, C] with unit column stride -> (tensor, leading dimension)."""
    assert t.dim() == 2 and (t.size(1) == 1 or t.stride(1) == 1), (t.shape, t.stride())
    return t, (t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1)))


# ----------------------------------------------------------------------------------------------
# raw GEMM helpers (no autograd)
# ----------------------------------------------------------------------------------------------
def _gemm(A, lda, ta, B, ldb, tb, Cm, ldc, bias, M, N, K, alpha=1.0, beta=0.0, a_rows=None):
    L.require_cuda(A, B, Cm)
    with _Span("gemm_store_kernel", 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N),
               detail="%s%s M=%d N=%d K=%d%s" % ("t" if ta else "n", "n" if tb else "t", M, N, K, " +=" if beta else "")) as sp:
        L.check(L.load().dssvae_gemm_f32(L.ptr(A), lda, int(ta), L.ptr(a_rows), L.ptr(B), ldb, int(tb), L.ptr(Cm),
                                         ldc, L.ptr(bias), M, N, K, alpha, beta, L.stream()), "dssvae_gemm_f32")
        if PROFILE is not None:
            sp.launches = L.load().dssvae_gemm_last_launches()


def mm_nt(x, w, bias=None, out=None, beta=0.0):
    """x[M,K] @ w[N,K]^T (+bias) -> [M,N]"""
    x, ldx = _rows(_f32(x))
    w, ldw = _rows(_f32(w))
    M, K = x.shape
    N = w.size(0)
    assert w.size(1) == K
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    out, ldo = _rows(out)
    if M and N:
        _gemm(x, ldx, 0, w, ldw, 0, out, ldo, bias, M, N, K, 1.0, beta)
    return out


def mm_nn(dy, w, out=None, beta=0.0):
    """dy[M,N] @ w[N,K] -> [M,K]"""
    dy, ldy = _rows(_f32(dy))
    w, ldw = _rows(_f32(w))
    M, N = dy.shape
    K = w.size(1)
    assert w.size(0) == N
    if out is None:
        out = torch.empty(M, K, device=dy.device, dtype=torch.float32)
    out, ldo = _rows(out)
    if M and K:
        _gemm(dy, ldy, 0, w, ldw, 1, out, ldo, None, M, K, N, 1.0, beta)
    return out


def mm_tn(dy, x, out=None, beta=0.0):
    """dy[M,N]^T @ x[M,K] -> [N,K]"""
    dy, ldy = _rows(_f32(dy))
    x, ldx = _rows(_f32(x))
    M, N = dy.shape
    K = x.size(1)
    assert x.size(0) == M
    if out is None:
        out = torch.empty(N, K, device=dy.device, dtype=torch.float32)
    out, ldo = _rows(out)
    if N and K:
        _gemm(dy, ldy, 1, x, ldx, 1, out, ldo, None, N, K, M, 1.0, beta)
    return out


def _gemm_grouped(kind, pairs, biases=None, outs_given=None):
    """kind 'nt': x[M,K] @ w[N,K]^T (+bias) | 'nn': dy[M,N] @ w[N,K] | 'tn': dy[M,N]^T @ x[M,K], for a list
    of (left, right) operand pairs, in as few launches as the group limit allows; -> list of outputs
    (views of one buffer)."""
    ta, tb = {"nt": (0, 0), "nn": (0, 1), "tn": (1, 1)}[kind]
    probs = []
    for i, (l, r) in enumerate(pairs):
        l, ldl = _rows(_f32(l))
        r, ldr = _rows(_f32(r))
        if kind == "nt":
            M, K, N = l.size(0), l.size(1), r.size(0)
            assert r.size(1) == K
        elif kind == "nn":
            M, K, N = l.size(0), l.size(1), r.size(1)
            assert r.size(0) == K
        else:
            M, K, N = l.size(1), l.size(0), r.size(1)
            assert r.size(0) == K
        probs.append((l, ldl, r, ldr, M, N, K, None if biases is None else biases[i]))
    L.require_cuda(*[p[0] for p in probs])
    dev = probs[0][0].device
    given = [None] * len(probs) if outs_given is None else list(outs_given)
    sizes = [0 if given[i] is not None else (p[4] * p[5] + 3) // 4 * 4 for i, p in enumerate(probs)]
    buf = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
    outs, off = [], 0
    for i, (p, n) in enumerate(zip(probs, sizes)):
        if given[i] is not None:
            assert given[i].shape == (p[4], p[5]) and given[i].is_contiguous()
            outs.append(given[i])
            continue
        outs.append(buf[off:off + p[4] * p[5]].view(p[4], p[5]))
        off += n
    lib = L.load()
    live = [i for i, p in enumerate(probs) if p[4] and p[5] and p[6]]
    for i, p in enumerate(probs):
        if i not in live and p[4] and p[5]:
            outs[i].zero_()  # empty inner dimension
    for c0 in range(0, len(live), L.GEMM_MAX_GROUP):
        idx = live[c0:c0 + L.GEMM_MAX_GROUP]
        arr = (L.GemmProblem * len(idx))()
        fl = by = 0.0
        for k, i in enumerate(idx):
            l, ldl, r, ldr, M, N, K, b = probs[i]
            arr[k] = L.GemmProblem(l.data_ptr(), ldl, r.data_ptr(), ldr, outs[i].data_ptr(), N,
                                   None if b is None else b.data_ptr(), M, N, K)
            fl += 2.0 * M * N * K
            by += 4.0 * (M * K + N * K + M * N)
        with _Span("gemm_group_kernel", fl, by,
                   detail="grouped %s x%d %s" % (kind, len(idx), " ".join("%dx%dx%d" % probs[i][4:7] for i in idx))):
            L.check(lib.dssvae_gemm_f32_grouped(arr, len(idx), ta, tb, L.stream()), "dssvae_gemm_f32_grouped")
    return outs


def _gemm_dlogits(slab, ldl, rows, cols, scale, trans_a, B, ldb, Cm, ldc, N, beta):
    """C (+)= dlogits(slab) . B or dlogits(slab)^T . B with the softmax gradient applied while the slab is staged"""
    M_, K_ = (cols, rows) if trans_a else (rows, cols)
    with _Span("gemm_store_kernel", 2.0 * M_ * N * K_, 4.0 * (M_ * K_ + N * K_ + M_ * N),
               detail="%s dlogits M=%d N=%d K=%d%s" % ("tn" if trans_a else "nn", M_, N, K_, " +=" if beta else "")) as sp:
        L.check(L.load().dssvae_gemm_f32_dlogits(L.ptr(slab), ldl, rows, cols, L.ptr(scale), scale.stride(0),
                                                 int(trans_a), L.ptr(B), ldb, L.ptr(Cm), ldc, N, float(beta), L.stream()),
                "dssvae_gemm_f32_dlogits")
        if PROFILE is not None:
            sp.launches = L.load().dssvae_gemm_last_launches()


_ONES_COL = {}


def _ones_col(M, dev):
    """[M, 4] = (1, 0, 0, 0) per row: appended to the wgrad's right operand, the bias gradient comes out of the
    GEMM as one more output column"""
    # one per stream: the fill is stream-ordered, another stream would read the column before it is written
    key = (dev, torch.cuda.current_stream().cuda_stream)
    t = None if DYN is not None else _ONES_COL.get(key)
    if t is None or t.size(0) < M:
        t = torch.zeros(max(M, 8192), 4, device=dev, dtype=torch.float32)
        t[:, 0] = 1.0
        if DYN is None:  # (under capture the fill is a node of the graph and the memory the graph's own)
            _ONES_COL[key] = t
    return t[:M]


def colsum(x, out=None, accumulate=False):
    x, ldx = _rows(_f32(x))
    M, N = x.shape
    if out is None:
        out = torch.empty(N, device=x.device, dtype=torch.float32)
        accumulate = False
    L.check(L.load().dssvae_colsum(L.ptr(x), ldx, M, N, L.ptr(out), int(accumulate), L.stream()), "dssvae_colsum")
    return out


# ----------------------------------------------------------------------------------------------
# Linear
# ----------------------------------------------------------------------------------------------
class _LinearFn(Function):
    @staticmethod
    def forward(ctx, x, w, b):
        L.require_cuda(x, w)
        # an output that receives no gradient must leave the inputs' gradients undefined (None), not zero:
        # torch.optim.Adam semantics (a parameter without a gradient is skipped) depend on it
        ctx.set_materialize_grads(False)
        x2 = x.reshape(-1, x.size(-1))
        ctx.save_for_backward(x2, w)
        ctx.has_bias = b is not None
        ctx.bias_ref = b  # not needed for the math: identifies the parameter whose gradient slice the backward fills
        ctx.xshape = x.shape
        return mm_nt(x2, w, b).view(*x.shape[:-1], w.size(0))

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.size(-1)).contiguous()
        dx = mm_nn(dy2, w).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        dw = mm_tn(dy2, x2, out=_claim(w)) if ctx.needs_input_grad[1] else None
        db = colsum(dy2, out=_claim(ctx.bias_ref)) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


def linear(x, w, b=None):
    return _LinearFn.apply(x, w, b)


class _LinearManyFn(Function):
    """several independent nn.Linear in grouped launches (forward, dgrad and wgrad each): tensors =
    (x_0, w_0, b_0, x_1, w_1, b_1, ...); every b_i is a tensor (GRU input projections always have one)."""

    @staticmethod
    def forward(ctx, *tensors):
        ctx.set_materialize_grads(False)
        xs, ws, bs = tensors[0::3], tensors[1::3], tensors[2::3]
        L.require_cuda(*xs)
        x2 = [x.reshape(-1, x.size(-1)) for x in xs]
        outs = _gemm_grouped("nt", list(zip(x2, ws)), biases=[b.contiguous() for b in bs])
        ctx.save_for_backward(*x2, *ws)
        ctx.n = len(xs)
        ctx.bias_refs = list(bs)
        ctx.xshapes = [x.shape for x in xs]
        return tuple(o.view(*x.shape[:-1], w.size(0)) for o, x, w in zip(outs, xs, ws))

    @staticmethod
    def backward(ctx, *dys):
        n = ctx.n
        saved = ctx.saved_tensors
        x2, ws = saved[:n], saved[n:]
        need = ctx.needs_input_grad
        dy2 = [None if dy is None else dy.reshape(-1, dy.size(-1)).contiguous() for dy in dys]
        ix = [i for i in range(n) if dy2[i] is not None and need[3 * i]]
        iw = [i for i in range(n) if dy2[i] is not None and need[3 * i + 1]]
        dxs = dict(zip(ix, _gemm_grouped("nn", [(dy2[i], ws[i]) for i in ix]))) if ix else {}
        dws = dict(zip(iw, _gemm_grouped("tn", [(dy2[i], x2[i]) for i in iw],
                                         outs_given=[_claim(ws[i]) for i in iw]))) if iw else {}
        out = []
        for i in range(n):
            dx = dxs[i].view(ctx.xshapes[i]) if i in dxs else None
            db = colsum(dy2[i], out=_claim(ctx.bias_refs[i])) if (dy2[i] is not None and need[3 * i + 2]) else None
            out += [dx, dws.get(i), db]
        return tuple(out)


def linear_many(triples):
    """[(x, w, b), ...] -> [x @ w^T + b, ...]; the products share grouped launches."""
    if len(triples) == 1:
        return [linear(*triples[0])]
    flat = []
    for x, w, b in triples:
        flat += [x, w, b]
    return list(_LinearManyFn.apply(*flat))


# ----------------------------------------------------------------------------------------------
# Embedding rows (+ optional dropout mask)
# ----------------------------------------------------------------------------------------------
class _EmbedFn(Function):
    @staticmethod
    def forward(ctx, table, ids, mask):
        L.require_cuda(table, ids)
        ctx.set_materialize_grads(False)
        flat = ids.reshape(-1).contiguous()
        M, E = flat.numel(), table.size(1)
        out = torch.empty(M, E, device=table.device, dtype=torch.float32)
        L.check(L.load().dssvae_embed_gather(L.ptr(table), table.stride(0), L.ptr(flat), M, E, L.ptr(mask),
                                             L.ptr(out), E, L.stream()), "dssvae_embed_gather")
        ctx.save_for_backward(flat, mask)
        ctx.tshape = table.shape
        ctx.table_ref = table
        return out.view(*ids.shape, E)

    @staticmethod
    def backward(ctx, g):
        flat, mask = ctx.saved_tensors
        if g is None or not ctx.needs_input_grad[0]:
            return None, None, None
        V, E = ctx.tshape
        g2 = g.reshape(-1, E).contiguous()
        # the table's slice of the flat gradient buffer (zero so far) when this is its first use in the step;
        # further uses of a shared table scatter into a temporary and autograd sums
        dt = _claim(ctx.table_ref)
        if dt is None:
            dt = torch.zeros(V, E, device=g.device, dtype=torch.float32)
        L.check(L.load().dssvae_embed_scatter_add(L.ptr(dt), E, L.ptr(flat), flat.numel(), E, L.ptr(mask),
                                                  L.ptr(g2), E, L.stream()), "dssvae_embed_scatter_add")
        return dt, None, None


def embed(table, ids, mask=None):
    return _EmbedFn.apply(table, ids, mask)


def dropout_mask(shape, p, seed, device):
    m = torch.empty(shape, device=device, dtype=torch.float32)
    seed, base = _seed_args(seed)
    L.check(L.load().dssvae_dropout_mask_dyn(L.ptr(m), m.numel(), float(p), seed, base, L.stream()),
            "dssvae_dropout_mask")
    return m


class _DropoutFn(Function):
    """y = x * mask(p, seed) / (1 - p); the mask is regenerated from the seed in backward (never stored)"""

    @staticmethod
    def forward(ctx, x, p, seed):
        L.require_cuda(x)
        ctx.set_materialize_grads(False)
        x = _f32(x).contiguous()
        y = torch.empty_like(x)
        ctx.p = float(p)
        ctx.seed, ctx.seed_base = _seed_args(seed)
        L.check(L.load().dssvae_dropout_apply_dyn(L.ptr(x), x.numel(), ctx.p, ctx.seed, ctx.seed_base, L.ptr(y),
                                                  L.stream()), "dssvae_dropout_apply")
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        L.check(L.load().dssvae_dropout_apply_dyn(L.ptr(dy), dy.numel(), ctx.p, ctx.seed, ctx.seed_base, L.ptr(dx),
                                                  L.stream()), "dssvae_dropout_apply")
        return dx, None, None


def dropout(x, p, seed):
    """nn.Dropout(p) in train mode with the mask of dropout_mask(x.shape, p, seed), in one launch each way"""
    if not p > 0.0 or x.numel() == 0:
        return x
    return _DropoutFn.apply(x, p, seed)


def unk_replace(ids, rate, entry, rand=None, seed=0):
    """word dropout on device (utils/nn_funcs.py:13-23)."""
    rate_dev = DYN.ptr(rate.slot) if (DYN is not None and isinstance(rate, Dyn)) else None
    if rate_dev is None and not rate > 0.0:
        return ids
    L.require_cuda(ids)
    ids = ids.contiguous()
    out = torch.empty_like(ids)
    if rand is not None:
        rand = rand.to(device=ids.device, dtype=torch.float32).contiguous()
    seed, base = _seed_args(seed)
    L.check(L.load().dssvae_unk_replace_dyn(L.ptr(ids), ids.numel(), L.ptr(rand), float(rate), rate_dev, seed, base,
                                            entry.sos_id, entry.pad_id, entry.eos_id, entry.unk_id, L.ptr(out),
                                            L.stream()), "dssvae_unk_replace")
    return out


# ----------------------------------------------------------------------------------------------
# Latent heads: reparameterisation + KL
# ----------------------------------------------------------------------------------------------
class _ReparamKLFn(Function):
    @staticmethod
    def forward(ctx, mean, logv, eps):
        L.require_cuda(mean, logv)
        ctx.set_materialize_grads(False)
        mean, logv = mean.contiguous(), logv.contiguous()
        eps = None if eps is None else eps.contiguous()
        z = torch.empty_like(mean)
        kl = torch.zeros((), device=mean.device, dtype=torch.float32)
        L.check(L.load().dssvae_reparam_kl_fwd(L.ptr(mean), L.ptr(logv), L.ptr(eps), mean.numel(), L.ptr(z),
                                               L.ptr(kl), L.stream()), "dssvae_reparam_kl_fwd")
        ctx.save_for_backward(mean, logv, eps)
        return z, kl

    @staticmethod
    def backward(ctx, dz, dkl):
        mean, logv, eps = ctx.saved_tensors
        if dz is None and dkl is None:
            return None, None, None
        dz = None if dz is None else dz.contiguous()
        dkl = None if dkl is None else dkl.contiguous()
        dm, dv = torch.empty_like(mean), torch.empty_like(logv)
        L.check(L.load().dssvae_reparam_kl_bwd(L.ptr(mean), L.ptr(logv), L.ptr(eps), L.ptr(dz), L.ptr(dkl),
                                               mean.numel(), L.ptr(dm), L.ptr(dv), L.stream()),
                "dssvae_reparam_kl_bwd")
        return dm, dv, None


class _WeightedSumsFn(Function):
    """out_j = sum_k coef[j][k] * sum(x_k): the scalar loss assembly in one launch (and one for its backward).  The
    outputs are separate 0-d views of one buffer, so backward gets one gradient per output (None for the reported-only
    ones) and hands their addresses to the kernel: no stacking, no select-backward zero fills."""

    @staticmethod
    def forward(ctx, coef, coef_dyn, dyn_ptr, *xs):
        ctx.set_materialize_grads(False)
        L.require_cuda(*xs)
        flat = [_f32(x).reshape(-1).contiguous() for x in xs]
        n_in, n_out = len(flat), len(coef)
        out = torch.empty(n_out, device=flat[0].device, dtype=torch.float32)
        ptrs = (C.c_void_p * n_in)(*[x.data_ptr() for x in flat])
        lens = (C.c_int32 * n_in)(*[x.numel() for x in flat])
        cf = (C.c_float * (n_in * n_out))(*[float(c) for row in coef for c in row])
        cfd = None if coef_dyn is None else (C.c_float * (n_in * n_out))(*[float(c) for row in coef_dyn for c in row])
        L.check(L.load().dssvae_weighted_sums_fwd_dyn(ptrs, lens, n_in, cf, cfd, dyn_ptr, n_out, L.ptr(out),
                                                      L.stream()), "dssvae_weighted_sums_fwd")
        if coef_dyn is not None:  # which inputs a live output reads: either part of the coefficient counts
            coef = tuple(tuple(1.0 if (a != 0.0 or b != 0.0) else 0.0 for a, b in zip(ra, rb))
                         for ra, rb in zip(coef, coef_dyn))
        ctx.coef, ctx.cf, ctx.lens, ctx.cfd, ctx.dyn_ptr = coef, cf, lens, cfd, dyn_ptr
        ctx.shapes = [x.shape for x in xs]
        ctx.dev = out.device
        return tuple(out[j] for j in range(n_out))

    @staticmethod
    def backward(ctx, *douts):
        n_in, n_out = len(ctx.shapes), len(ctx.coef)
        live_out = [j for j in range(n_out) if douts[j] is not None]
        # an input none of whose live outputs carries a non-zero coefficient gets NO gradient (None, not zeros):
        # the optimiser's "no gradient = parameter skipped" semantics depend on it (adversary terms before warm_up)
        want = [ctx.needs_input_grad[3 + k] and any(ctx.coef[j][k] != 0.0 for j in live_out) for k in range(n_in)]
        if not any(want):
            return (None,) * (3 + n_in)
        sizes = [ctx.lens[k] if want[k] else 0 for k in range(n_in)]
        buf = torch.empty(sum(sizes), device=ctx.dev, dtype=torch.float32)
        grads, off = [], 0
        for k in range(n_in):
            grads.append(buf[off:off + sizes[k]].view(ctx.shapes[k]) if want[k] else None)
            off += sizes[k]
        g = [None if d is None else _f32(d).contiguous() for d in douts]
        dptr = (C.c_void_p * n_out)(*[None if d is None else d.data_ptr() for d in g])
        iptr = (C.c_void_p * n_in)(*[None if x is None else x.data_ptr() for x in grads])
        L.check(L.load().dssvae_weighted_sums_bwd_dyn(iptr, ctx.lens, n_in, ctx.cf, ctx.cfd, ctx.dyn_ptr, n_out, dptr,
                                                      L.stream()), "dssvae_weighted_sums_bwd")
        return (None, None, None, *grads)


def weighted_sums(coef, xs, coef_dyn=None, dyn=None):
    """coef[j][k] (python floats), xs[k] (tensors of any shape) -> tuple of 0-d tensors out_j = sum_k coef[j][k] * xs[k].sum().
    coef_dyn / dyn (a Dyn): the coefficient is coef + coef_dyn * dyn, with dyn read on the device when a step is being
    captured (its value folded into coef on the host otherwise)."""
    coef = tuple(tuple(float(c) for c in row) for row in coef)
    if coef_dyn is None or dyn is None:
        return _WeightedSumsFn.apply(coef, None, None, *xs)
    if DYN is None or not isinstance(dyn, Dyn):
        d = float(dyn)
        return _WeightedSumsFn.apply(tuple(tuple(a + b * d for a, b in zip(ra, rb)) for ra, rb in zip(coef, coef_dyn)),
                                     None, None, *xs)
    return _WeightedSumsFn.apply(coef, tuple(tuple(float(c) for c in row) for row in coef_dyn), DYN.ptr(dyn.slot), *xs)


def reparam_kl(mean, logv, eps):
    """-> (z, kl_sum); eps None => z = mean."""
    return _ReparamKLFn.apply(mean, logv, eps)


# ----------------------------------------------------------------------------------------------
# Fused vocabulary projection + cross-entropy
# ----------------------------------------------------------------------------------------------
def _pick_split(M, N):
    """column splits of the LSE kernel: about two workgroups per CU (its occupancy); the kernel deals the
    128-column tiles evenly over the splits (no split more than one tile longer than another)"""
    tiles_m, tiles_n = (M + 127) // 128, (N + 127) // 128
    S = max(1, min(tiles_n, (2 * n_cus()) // tiles_m, 128))
    # a few column tiles and many row blocks (the tree heads: 3 tiles x 182 row blocks): one tile per workgroup even
    # when that is a little more than a round of the chip -- two splits of 2 + 1 tiles leave half the workgroups with
    # twice the work
    if S < tiles_n <= 4 and tiles_m * tiles_n <= 3 * n_cus() and _os.environ.get("DSSVAE_LSE_EVEN", "1") != "0":
        S = tiles_n
    return S


def _one_panel(xh, xe):
    """[xh | xe] as ONE row-major matrix: the projection kernels' 16-byte stagers take a single activation panel (two
    panels run on 8-byte granules, ~8 % slower); the copy is M x K floats against a product of 2 M V K flops"""
    return torch.cat((xh, xe), dim=1)


def _lse(xh, ldh, Kh, xe, lde, Ke, W, ldw, b, M, N, want_arg=False, logits=None, lse_out=None, tile_max=None):
    """tile_max ([ceil(N / 128), >= M] floats): `logits` receives the slab normalised per 128-column tile and the
    table the tiles' row maxima (dssvae_linear_lse_fwd_norm: tiles dealt evenly, no column splits) instead of raw logits"""
    if tile_max is not None:
        assert not want_arg
        if xe is not None and Ke > 0 and (Kh + Ke) % 4 == 0:
            xh, xe = _one_panel(xh[:M], xe[:M]), None
            ldh, Kh, lde, Ke = xh.stride(0), Kh + Ke, 0, 0
        lib = L.load()
        tsum = torch.empty_like(tile_max)
        tiles = tile_max.size(0)
        with _Span("linear_lse_kernel", 2.0 * M * N * (Kh + Ke), 4.0 * (M * (Kh + Ke) + N * (Kh + Ke) + 2 * M * tiles),
                   detail="M=%d N=%d K=%d+%d tiles normalised store" % (M, N, Kh, Ke)):
            L.check(lib.dssvae_linear_lse_fwd_norm(L.ptr(xh), ldh, Kh, L.ptr(xe), lde, Ke, L.ptr(W), ldw, L.ptr(b), M, N,
                                                   L.ptr(logits), logits.stride(0), L.ptr(tile_max), L.ptr(tsum),
                                                   tile_max.stride(0), L.stream()), "dssvae_linear_lse_fwd_norm")
        lse = torch.empty(M, device=xh.device, dtype=torch.float32) if lse_out is None else lse_out
        L.check(lib.dssvae_lse_combine_tiles(L.ptr(tile_max), L.ptr(tsum), M, tiles, tile_max.stride(0), L.ptr(lse),
                                             L.stream()), "dssvae_lse_combine_tiles")
        return lse, None
    if xe is not None and Ke > 0 and (Kh + Ke) % 4 == 0:
        xh, xe = _one_panel(xh[:M], xe[:M]), None
        ldh, Kh, lde, Ke = xh.stride(0), Kh + Ke, 0, 0
    S = _pick_split(M, N)
    dev = xh.device
    pm = torch.empty(M, S, device=dev, dtype=torch.float32)
    ps = torch.empty(M, S, device=dev, dtype=torch.float32)
    pa = torch.empty(M, S, device=dev, dtype=torch.int64) if want_arg else None
    lib = L.load()
    with _Span("linear_lse_kernel", 2.0 * M * N * (Kh + Ke), 4.0 * (M * (Kh + Ke) + N * (Kh + Ke) + 2 * M * S),
               detail="M=%d N=%d K=%d+%d S=%d%s" % (M, N, Kh, Ke, S, " store" if logits is not None else "")):
        L.check(lib.dssvae_linear_lse_fwd(L.ptr(xh), ldh, Kh, L.ptr(xe), lde, Ke, L.ptr(W), ldw, L.ptr(b), M, N, S,
                                          L.ptr(pm), L.ptr(ps), L.ptr(pa), int(want_arg), L.ptr(logits),
                                          0 if logits is None else logits.stride(0), L.stream()),
                "dssvae_linear_lse_fwd")
    lse = torch.empty(M, device=dev, dtype=torch.float32) if lse_out is None else lse_out
    arg = torch.empty(M, device=dev, dtype=torch.int64) if want_arg else None
    L.check(lib.dssvae_lse_combine(L.ptr(pm), L.ptr(ps), L.ptr(pa), M, S, L.ptr(lse), L.ptr(arg), L.stream()),
            "dssvae_lse_combine")
    return lse, arg


import os as _os
# dlogits slab of one row chunk.  Measured on the headline workload (bench.py --shapes, DSSVAE_SLAB_MB sweep):
# 128 MiB (inside the 256 MiB Infinity Cache) 5.88k sentences/s, 256 MiB 6.02k, 512 MiB 6.09k -- the consumers
# of the slab are MFMA-bound GEMMs that take it from HBM just as well, while longer chunks quantise better
# over the chip (LSE: 3.67 -> 14.2 rounds of workgroups) and pass over dW fewer times.  Round 2: 1024 MiB holds the
# headline's 7936 x 30000 logits in ONE chunk (28.5 rounds, dW written once): 31.05 -> 30.82 ms per step; 288 GB of HBM
# make a gigabyte of scratch per process a small price.
SLAB_BYTES = int(_os.environ.get("DSSVAE_SLAB_MB", "1024")) << 20


def _col_chunks(M, N):
    """split the vocabulary into equal chunks of whole 128-column tiles whose [M, nc] slab fits SLAB_BYTES"""
    nc_max = max(128, (SLAB_BYTES // (4 * M)) // 128 * 128)
    n_chunks = (N + nc_max - 1) // nc_max
    nc = ((N + n_chunks - 1) // n_chunks + 127) // 128 * 128
    return nc


# Row chunks of the fused CE can alternate between several streams (each with its own slab): +4 % on the
# step at 2 streams, but concurrent kernels blur the per-kernel durations the roofline report is built on,
# so the default stays at one stream.
CE_STREAMS = max(1, int(_os.environ.get("DSSVAE_CE_STREAMS", "1")))
# the fused CE schedule scales stored unit gradients by row 0's upstream gradient (the caller's contract: all rows
# receive the same one); with this flag every backward verifies it (one device sync each: tests / debugging only)
FUSE_DLOGITS = _os.environ.get("DSSVAE_FUSE_DLOGITS", "1") != "0"  # 0: separate softmax-grad pass over the slab (A/B)
CHECK_UNIFORM_UPSTREAM = _os.environ.get("DSSVAE_CHECK_UNIFORM", "0") == "1"
WGRAD_REM_COLS = _os.environ.get("DSSVAE_WGRAD_REM_COLS", "1") != "0"  # 0: the wgrad's remainder columns stay in the MFMA launch (A/B)
_SIDE = {}


def _side_stream(dev, i=0):
    key = (torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device(), i)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def _row_chunks(M, ldd, slab_bytes=None):
    """rows per chunk so that a [rows, ldd] logits slab fits slab_bytes (whole 128-row tiles, balanced)"""
    slab_bytes = SLAB_BYTES if slab_bytes is None else slab_bytes
    mc_max = max(128, (slab_bytes // (4 * ldd)) // 128 * 128)
    n_chunks = (M + mc_max - 1) // mc_max
    return ((M + n_chunks - 1) // n_chunks + 127) // 128 * 128


class _LinearCEFn(Function):
    """nll[m] = sum_t valid(m,t) * (lse[m] - logit[m, target[m,t]])  -- full logits never materialised.

    Two training schedules:
    * uniform=False (general): forward = fused projection + online LSE only; backward recomputes the
      projection per chunk of vocabulary columns (4 GEMM passes per trainable head in total).
    * uniform=True (the caller guarantees that every row receives the SAME upstream gradient, true for
      every loss term of the DSS-VAE models: sums of per-token NLL scaled by constants): forward walks
      row chunks; the LSE kernel also drops the chunk's logits into a reusable slab (SLAB_BYTES), which
      is turned into dlogits in place and consumed by the dgrad / wgrad GEMMs right away.  The
      gradients for upstream = 1 are kept and scaled in backward (dX per row, dW / db by the common
      factor): 3 GEMM passes per trainable head, 2 per frozen one.  fp32 MFMA (157 TFLOP/s) makes the
      recompute ~7x dearer than the slab round trip through the cache."""

    @staticmethod
    def forward(ctx, xh, xe, W, b, target, ignore_index, uniform, group):
        L.require_cuda(xh, W, target)
        ctx.set_materialize_grads(False)
        xh, ldh = _rows(_f32(xh))
        Kh = xh.size(1)
        if xe is not None:
            xe, lde = _rows(_f32(xe))
            Ke = xe.size(1)
        else:
            lde, Ke = 0, 0
        K = Kh + Ke
        W, ldw = _rows(_f32(W))
        M, N = xh.size(0), W.size(0)
        assert W.size(1) == K
        target = target.reshape(M, -1).contiguous()
        T = target.size(1)
        dev = xh.device
        lib = L.load()
        cnt = torch.empty(M, device=dev, dtype=torch.float32)
        L.check(lib.dssvae_ce_count(L.ptr(target), T, int(ignore_index), M, L.ptr(cnt), L.stream()), "dssvae_ce_count")
        group = max(1, int(group))
        ctx.group, ctx.M = group, M
        need_xh, need_xe, need_w, need_b = ctx.needs_input_grad[:4]
        need_xe = need_xe and xe is not None
        need_b = need_b and b is not None
        ctx.ignore = int(ignore_index)
        ctx.fused = bool(uniform) and (need_xh or need_xe or need_w or need_b)
        if not ctx.fused:
            lse, _ = _lse(xh, ldh, Kh, xe, lde, Ke, W, ldw, b, M, N)
        else:
            ldd = (N + 3) // 4 * 4
            # Row chunks alternate between two streams, each with its own cache-sized slab, so two
            # chunks are in flight at once: the tail of one kernel (the last, partly filled wave of
            # workgroups) is filled by the other chain's kernel.  The only cross-chunk dependency is
            # the accumulation into dW, ordered by events.
            n_str = CE_STREAMS if (M * ldd * 4 > SLAB_BYTES // CE_STREAMS) else 1
            mc = _row_chunks(M, ldd, SLAB_BYTES // n_str)
            n_str = min(n_str, (M + mc - 1) // mc)
            slabs = [torch.empty(min(mc, M), ldd, device=dev, dtype=torch.float32) for _ in range(n_str)]
            lse = torch.empty(M, device=dev, dtype=torch.float32)
            need_x = need_xh or need_xe
            # only the panels whose input wants a gradient (a frozen adversary's embedding panel does not:
            # half of its dgrad GEMM)
            kx0, kx1 = (0 if need_xh else Kh), (K if need_xe else Kh)
            dx = torch.empty(M, kx1 - kx0, device=dev, dtype=torch.float32) if need_x else None
            # One target per row (the sequence decoders): the softmax gradient is applied by the dgrad / wgrad GEMMs
            # while they stage the slab (dssvae_gemm_f32_dlogits): no pass over the slab in between, and the bias
            # gradient is one more output column of the wgrad (a column of ones in its right operand).
            fuse = (FUSE_DLOGITS and T == 1 and N % 4 == 0 and K % 4 == 0 and ldw % 4 == 0 and
                    (W.data_ptr() + 4 * kx0) % 16 == 0 and (kx1 - kx0) % 4 == 0 and (need_w or not need_b))
            Kw = K + 4 if (fuse and need_b) else K
            dWp = torch.empty(N, Kw, device=dev, dtype=torch.float32) if need_w else None
            db = torch.zeros(N, device=dev, dtype=torch.float32) if (need_b and not fuse) else None
            ones = None if fuse else torch.ones(M, T, device=dev, dtype=torch.float32)
            # per stream-slab: the row maxima of every 128-column tile of the chunk, turned into the stagers' scales
            tabs = [torch.empty((N + 127) // 128, min(mc, M), device=dev, dtype=torch.float32) if fuse else None
                    for _ in range(n_str)]
            xcat = None
            if need_w or xe is not None:   # (one activation panel for the projection kernels as well: _one_panel)
                parts = [xh] if xe is None else [xh, xe]
                if Kw > K:
                    parts.append(_ones_col(M, dev))
                xcat = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
                if fuse and (xcat.stride(0) % 4 != 0 or xcat.data_ptr() % 16 != 0):
                    # the wgrad's right operand went in as the caller's own view (no concatenation made a fresh one):
                    # the staged-slab GEMM wants 16-byte rows
                    xcat = xcat.contiguous()
                    if xcat.stride(0) % 4 != 0:
                        xcat = torch.nn.functional.pad(xcat, (0, 4 - xcat.size(1) % 4))[:, :Kw]
            main = torch.cuda.current_stream()
            streams = [main] + [_side_stream(dev, i) for i in range(n_str - 1)]
            for st in streams[1:]:
                st.wait_stream(main)
            wg_done = None  # event: the previous chunk's wgrad (accumulation into dW) has finished
            for ci, m0 in enumerate(range(0, M, mc)):
                rows = min(mc, M - m0)
                st = streams[ci % n_str]
                with torch.cuda.stream(st):
                    dl = slabs[ci % n_str][:rows]
                    tab = tabs[ci % n_str]
                    if xe is not None:   # [h | e (| 1 0 0 0)] is one matrix already: its first K columns
                        _lse(xcat[m0:m0 + rows], xcat.stride(0), K, None, 0, 0, W, ldw, b, rows, N, logits=dl,
                             lse_out=lse[m0:m0 + rows], tile_max=tab)
                    else:
                        _lse(xh[m0:m0 + rows], ldh, Kh, None, 0, 0, W, ldw, b, rows, N, logits=dl,
                             lse_out=lse[m0:m0 + rows], tile_max=tab)
                    if fuse:
                        L.check(lib.dssvae_ce_rowscale(L.ptr(lse[m0:]), L.ptr(cnt[m0:]), rows, tab.size(0), L.ptr(tab),
                                                       tab.stride(0), L.stream()), "dssvae_ce_rowscale")
                        if need_x:
                            _gemm_dlogits(dl, ldd, rows, N, tab, 0, W[:, kx0:kx1], ldw, dx[m0:m0 + rows],
                                          dx.stride(0), kx1 - kx0, 0.0)
                    else:
                        L.check(lib.dssvae_softmax_grad_inplace(L.ptr(dl), ldd, L.ptr(lse[m0:]), L.ptr(cnt[m0:]), rows,
                                                                N, L.ptr(db), L.stream()), "dssvae_softmax_grad_inplace")
                        L.check(lib.dssvae_onehot_sub(L.ptr(dl), ldd, L.ptr(target[m0:]), T, ctx.ignore,
                                                      L.ptr(ones[m0:]), rows, 0, N, L.ptr(db), L.stream()),
                                "dssvae_onehot_sub")
                        if need_x:
                            mm_nn(dl[:, :N], W[:, kx0:kx1], out=dx[m0:m0 + rows])
                    if need_w:
                        if wg_done is not None and n_str > 1:
                            st.wait_event(wg_done)
                        if fuse:
                            # The last few output columns (904 = 7 x 128 + 8) would cost a 128x64 MFMA tile column for
                            # 8 live columns (0.25 ms): they go through dssvae_ce_wgrad_cols, a stream over the slab on the
                            # vector ALUs (0.20 ms), in front of the wide tiles' launch.  On a side stream beside that
                            # launch either form measured SLOWER per step than in line (the narrow MFMA launch by 0.25 ms,
                            # the stream by 0.19 ms): the wide launch loses more than the side launch takes.
                            rem = Kw % 128 if WGRAD_REM_COLS else 0
                            if 0 < rem <= 8 and Kw > 128:
                                with _Span("ce_wgrad_cols_kernel", 2.0 * N * rem * rows,
                                           4.0 * (rows * N + rows * rem + N * rem)):
                                    L.check(lib.dssvae_ce_wgrad_cols(
                                        L.ptr(dl), ldd, rows, N, L.ptr(tab), tab.stride(0),
                                        C.c_void_p(xcat.data_ptr() + 4 * (m0 * xcat.stride(0) + Kw - rem)), xcat.stride(0),
                                        rem, C.c_void_p(dWp.data_ptr() + 4 * (Kw - rem)), Kw, 0.0 if m0 == 0 else 1.0,
                                        L.stream()), "dssvae_ce_wgrad_cols")
                            else:
                                rem = 0
                            _gemm_dlogits(dl, ldd, rows, N, tab, 1, xcat[m0:m0 + rows], xcat.stride(0), dWp,
                                          Kw, Kw - rem, 0.0 if m0 == 0 else 1.0)
                        else:
                            mm_tn(dl[:, :N], xcat[m0:m0 + rows], out=dWp, beta=0.0 if m0 == 0 else 1.0)
                    if fuse and (need_x or need_w):
                        # the one-hot part of dlogits, as a correction of this chunk's two products
                        Wp = W[:, kx0:kx1]
                        L.check(lib.dssvae_ce_onehot_fixup(
                            L.ptr(target[m0:]), ctx.ignore, None, rows, L.ptr(Wp) if need_x else None, ldw, kx1 - kx0,
                            L.ptr(dx[m0:]) if need_x else None, dx.stride(0) if need_x else 0,
                            L.ptr(xcat[m0:]) if need_w else None, xcat.stride(0) if need_w else 0, Kw,
                            L.ptr(dWp) if need_w else None, Kw, L.stream()), "dssvae_ce_onehot_fixup")
                    if need_w and n_str > 1:
                        wg_done = _event()
                        wg_done.record(st)
            for st in streams[1:]:
                main.wait_stream(st)
            dW = None
            if dWp is not None:
                dW = dWp[:, :K] if Kw > K else dWp
                if Kw > K:
                    db = dWp[:, K]
            ctx.dWp, ctx.db_rides = dWp, Kw > K
            ctx.param_refs = (W, b)
            ctx.unit_grads = (dx, dW, db)
            ctx.Kh, ctx.kx0 = Kh, kx0
        # loss of every `group` consecutive rows (a sentence), gather + mask + sum in one launch
        out = torch.empty((M + group - 1) // group, device=dev, dtype=torch.float32)
        rows_nll = torch.empty(M, device=dev, dtype=torch.float32) if group > 1 else None
        L.check(lib.dssvae_ce_finish(L.ptr(xh), ldh, Kh, L.ptr(xe), lde, Ke, L.ptr(W), ldw, L.ptr(b), L.ptr(target), T,
                                     ctx.ignore, L.ptr(lse), M, group, L.ptr(rows_nll), L.ptr(out), L.stream()),
                "dssvae_ce_finish")
        if not ctx.fused:
            ctx.save_for_backward(xh, xe, W, b, target, lse, cnt)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            ctx.unit_grads = ctx.dWp = ctx.param_refs = None
            return None, None, None, None, None, None, None, None
        need_xh, need_xe, need_w, need_b = ctx.needs_input_grad[:4]
        group, M_rows = ctx.group, ctx.M
        if ctx.fused:
            # every row has the same upstream gradient (the caller's contract): scale the unit gradients
            if ctx.unit_grads is None:
                raise RuntimeError("linear_ce(uniform_upstream=True) keeps its gradients for ONE backward pass "
                                   "(retain_graph / a second backward is not supported on this schedule)")
            dx, dW, db = ctx.unit_grads
            ctx.unit_grads = None
            c = g[:1]
            if CHECK_UNIFORM_UPSTREAM and not bool((g == c).all()):  # DSSVAE_CHECK_UNIFORM=1: tests / debugging
                raise RuntimeError("linear_ce(uniform_upstream=True): rows received different upstream gradients")
            dxh = dxe = None
            if dx is not None:
                if group == 1:
                    dx.mul_(g.unsqueeze(1))
                elif M_rows % group == 0:
                    dx.view(M_rows // group, group, dx.size(1)).mul_(g.view(-1, 1, 1))
                else:
                    dx.mul_(g.repeat_interleave(group)[:M_rows].unsqueeze(1))
                dxh = dx[:, :ctx.Kh] if need_xh else None  # kx0 == 0 whenever the h panel is wanted
                dxe = dx[:, ctx.Kh - ctx.kx0:] if need_xe else None
            dWp, ctx.dWp = ctx.dWp, None
            W_ref, b_ref = ctx.param_refs
            ctx.param_refs = None
            if dW is not None:
                gW = _claim(W_ref)
                # c * (unit gradient), written into the parameter's slice of the flat gradient buffer when there
                # is one (one pass: no scaled temporary, no accumulate kernel behind it)
                dW = torch.mul(dW, c, out=gW) if gW is not None else dW.mul_(c)
            if db is not None:
                gb = _claim(b_ref)
                db = torch.mul(db, c, out=gb) if gb is not None else (db * c if ctx.db_rides else db.mul_(c))
            return dxh, dxe, dW, db, None, None, None, None
        # general path: per chunk of vocabulary columns, recompute dlogits into a
        # slab [M, nc] (+ the bias gradient by column sums in the same kernel), subtract the one-hot part,
        # then dX += dL.W_chunk (dgrad, accumulating) and dW_chunk = dL^T.[xh|xe] (wgrad, written once)
        xh, xe, W, b, target, lse, cnt = ctx.saved_tensors
        xh, ldh = _rows(xh)
        Kh = xh.size(1)
        if xe is not None:
            xe, lde = _rows(xe)
            Ke = xe.size(1)
        else:
            lde, Ke, need_xe = 0, 0, False
        K = Kh + Ke
        W, ldw = _rows(W)
        M, N = xh.size(0), W.size(0)
        T = target.size(1)
        dev = xh.device
        if group > 1:
            g = g.repeat_interleave(group)[:M]
        g = g.contiguous()
        coef = (g * cnt).contiguous()
        g_mt = g.unsqueeze(1).expand(M, T).contiguous()
        nc = _col_chunks(M, N)
        ldd = min(nc, (N + 3) // 4 * 4)
        slab = torch.empty(M, ldd, device=dev, dtype=torch.float32)
        need_x = need_xh or need_xe
        dx = torch.empty(M, K, device=dev, dtype=torch.float32) if need_x else None
        dW = torch.empty(N, K, device=dev, dtype=torch.float32) if need_w else None
        db = torch.zeros(N, device=dev, dtype=torch.float32) if (need_b and b is not None) else None
        xcat = None
        if need_w:
            xcat = xh if xe is None else torch.cat((xh, xe), dim=1)
        lib = L.load()
        for n0 in range(0, N, nc):
            n1 = min(N, n0 + nc)
            w = n1 - n0
            Wc = W[n0:n1]
            with _Span("linear_dlogits_kernel", 2.0 * M * w * K, 4.0 * (M * K + w * K + M * w)):
                L.check(lib.dssvae_linear_ce_dlogits(L.ptr(xh), ldh, Kh, L.ptr(xe), lde, Ke, L.ptr(Wc), ldw,
                                                     None if b is None else C.c_void_p(b.data_ptr() + 4 * n0),
                                                     L.ptr(lse), L.ptr(coef), M, w, L.ptr(slab), ldd,
                                                     None if db is None else C.c_void_p(db.data_ptr() + 4 * n0),
                                                     L.stream()), "dssvae_linear_ce_dlogits")
            L.check(lib.dssvae_onehot_sub(L.ptr(slab), ldd, L.ptr(target), T, ctx.ignore, L.ptr(g_mt), M, n0, n1,
                                          None if db is None else C.c_void_p(db.data_ptr() + 4 * n0), L.stream()),
                    "dssvae_onehot_sub")
            dlv = slab[:, :w]
            if need_x:
                mm_nn(dlv, Wc, out=dx, beta=0.0 if n0 == 0 else 1.0)
            if need_w:
                mm_tn(dlv, xcat, out=dW[n0:n1])
        dxh = dx[:, :Kh] if need_xh else None
        dxe = dx[:, Kh:] if need_xe else None
        return dxh, dxe, dW, db, None, None, None, None


def linear_ce(xh, xe, W, b, target, ignore_index=0, uniform_upstream=False, group=1):
    """-> per-row summed NLL [M] (group = 1) or the NLL of every `group` consecutive rows summed [ceil(M / group)]
    (the rows of a sentence); xe may be None; target [M] or [M,T].  uniform_upstream=True: the caller
    guarantees that every row's NLL receives the same upstream gradient (see _LinearCEFn)."""
    # under no_grad nothing will ever ask for gradients: take the forward-only path (inside Function.forward
    # grad mode is always off and needs_input_grad still mirrors requires_grad, so decide here)
    return _LinearCEFn.apply(xh, xe, W, b, target, ignore_index, bool(uniform_upstream) and torch.is_grad_enabled(),
                             group)


def linear_argmax(xh, xe, W, b):
    """argmax_n of the projection, fused (greedy decode; no logits materialised)."""
    L.require_cuda(xh, W)
    xh, ldh = _rows(_f32(xh))
    if xe is not None:
        xe, lde = _rows(_f32(xe))
        Ke = xe.size(1)
    else:
        lde, Ke = 0, 0
    W, ldw = _rows(W)
    _, arg = _lse(xh, ldh, xh.size(1), xe, lde, Ke, W, ldw, b, xh.size(0), W.size(0), want_arg=True)
    return arg


def linear_log_softmax(xh, xe, W, b, out=None):
    """log_softmax of the projection, materialised [M, N] (beam search keeps every step's rows: beam_decoder.py:152-153):
    the LSE kernel stores the logits beside its online log-sum-exp, one in-place pass then subtracts it.  `out`: a
    caller buffer [M, N] (row stride free) the rows are written to.  Forward only."""
    L.require_cuda(xh, W)
    xh, ldh = _rows(_f32(xh))
    if xe is not None:
        xe, lde = _rows(_f32(xe))
        Ke = xe.size(1)
    else:
        lde, Ke = 0, 0
    W, ldw = _rows(W)
    M, N = xh.size(0), W.size(0)
    if out is None:
        out = torch.empty(M, N, device=xh.device, dtype=torch.float32)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    lse, _ = _lse(xh, ldh, xh.size(1), xe, lde, Ke, W, ldw, b, M, N, logits=out)
    return out.sub_(lse.unsqueeze(1))


# ----------------------------------------------------------------------------------------------
# Persistent GRU
# ----------------------------------------------------------------------------------------------
GRU_LDS_BUDGET = 150 * 1024
FORCE_STEPWISE_GRU = False  # tests: send every recurrence down the stepwise path


GRU_PACK_ROWS = int(_os.environ.get("DSSVAE_GRU_PACK_ROWS", "128"))  # development knob: 64 = never merge into 8-wave launches


def _cells(B, H):
    """workgroups a recurrence needs at least in a persistent launch: one per (128 batch rows) x (slice of 16
    hidden units); the library takes 64-row workgroups when the whole launch fits that way"""
    return ((B + GRU_PACK_ROWS - 1) // GRU_PACK_ROWS) * ((H + 15) // 16)


def _fits_persistent(B, H):
    """can one persistent launch hold this recurrence?  (one workgroup per CU, its W_hh slice
    3 x 16 x H fp32 resident in LDS)"""
    if FORCE_STEPWISE_GRU:
        return False
    return _cells(B, H) <= n_cus() and 3 * ((H + 15) // 16) * 1024 <= GRU_LDS_BUDGET


def _gru_steps_fwd(x, whh, bhh, h0, lengths, reverse, o, col, hl, gates):
    """stepwise forward of one recurrence too large for the persistent kernel: per timestep the
    recurrent product on the MFMA GEMM (at these sizes the step is compute-bound) + the fused gate kernel"""
    B, T, _ = x.shape
    H = whh.size(1)
    dev = x.device
    lib = L.load()
    h = torch.zeros(B, H, device=dev, dtype=torch.float32) if h0 is None else h0
    gh = torch.empty(B, 3 * H, device=dev, dtype=torch.float32)
    nxt = [torch.empty(B, H, device=dev, dtype=torch.float32) for _ in range(2)]
    ldx, W = _xproj_ld(x), o.size(2)
    lp = None if lengths is None else lengths.data_ptr()
    for step in range(T):
        t = T - 1 - step if reverse else step
        mm_nt(h, whh, out=gh)
        hn = hl if step == T - 1 else nxt[step & 1]
        with _Span("gru_step_kernel", 0.0, 4.0 * B * H * (3 + 3 + 1 + 2 + (5 if gates is not None else 0))):
            L.check(lib.dssvae_gru_step_fwd(
                x.data_ptr() + 4 * t * ldx, T * ldx, L.ptr(gh), L.ptr(bhh), L.ptr(h), lp, t,
                B, H, L.ptr(hn), o.data_ptr() + 4 * (t * W + col), T * W,
                None if gates is None else gates.data_ptr() + 4 * t * 5 * H, T * 5 * H, L.stream()),
                "dssvae_gru_step_fwd")
        h = hn


def _gru_steps_bwd(whh, gates, lengths, reverse, do, col, dhl, dgx, dgh, dh0):
    B, T, K = dgx.shape
    H = K // 3
    dev = dgx.device
    lib = L.load()
    buf = [torch.empty(B, H, device=dev, dtype=torch.float32) for _ in range(2)]
    dh = dhl
    lp = None if lengths is None else lengths.data_ptr()
    for step in range(T):
        fstep = T - 1 - step
        t = T - 1 - fstep if reverse else fstep
        dhz = dh0 if step == T - 1 else buf[step & 1]
        with _Span("gru_step_kernel", 0.0, 4.0 * B * H * 13):
            L.check(lib.dssvae_gru_step_bwd(
                gates.data_ptr() + 4 * t * 5 * H, T * 5 * H,
                None if do is None else do.data_ptr() + 4 * (t * do.size(2) + col), 0 if do is None else T * do.size(2),
                L.ptr(dh), lp, t, B, H, dgx.data_ptr() + 4 * t * K, dgh.data_ptr() + 4 * t * K, T * K, L.ptr(dhz),
                L.stream()), "dssvae_gru_step_bwd")
        mm_nn(dgh[:, t, :], whh, out=dhz, beta=1.0)
        dh = dhz


def _pack(insts):
    """Group instances (B, H, T) into persistent launches whose workgroups are co-resident (one per
    CU) and that hold <= 8 instances, longest recurrences first (first-fit decreasing on T): a
    launch lasts as long as its longest instance, so equal lengths share launches."""
    cus = n_cus()
    launches = []  # [indices, workgroups]
    for i in sorted(range(len(insts)), key=lambda i: -insts[i][2]):
        B, H, _ = insts[i]
        need = _cells(B, H)
        if need > cus:
            raise RuntimeError("GRU instance B=%d H=%d needs %d workgroups > %d CUs: not supported by the "
                               "persistent kernel yet" % (B, H, need, cus))
        for l in launches:
            if l[1] + need <= cus and len(l[0]) < L.GRU_MAX_INST:
                l[0].append(i)
                l[1] += need
                break
        else:
            launches.append([[i], need])
    return [sorted(l[0]) for l in launches]


GRU_SEGMENT_MIN = int(_os.environ.get("DSSVAE_GRU_SEGMENT_MIN", "16"))  # 0 = never cut a launch in time


def _segments(Ts):
    """Cut a persistent launch in time where its recurrences end: [(t0, t1, members)] with members = positions in
    Ts of the recurrences still running in steps [t0, t1).  Once the short recurrences are over, the long ones
    continue in a launch of their own -- fewer cells, so the library can give them finer cells with the K range of
    a step split over more waves (csrc/gru.hip: pick_rows) instead of idling the finished recurrences' CUs.  A cut
    is made only where at least GRU_SEGMENT_MIN steps remain (it costs a launch boundary)."""
    cuts, t0 = [], 0
    ends = sorted(set(Ts))
    for k, t1 in enumerate(ends):
        last = k == len(ends) - 1
        if not last and (GRU_SEGMENT_MIN <= 0 or ends[-1] - t1 < GRU_SEGMENT_MIN or t1 - t0 < GRU_SEGMENT_MIN):
            continue
        cuts.append((t0, t1, [j for j, T in enumerate(Ts) if T > t0]))
        t0 = t1
    return cuts


GRU_SCHEDULE = _os.environ.get("DSSVAE_GRU_SCHEDULE", "1") != "0"   # 0 = time cuts only, the library picks the cells
_SCHEDULE_CACHE = {}


def _step_us(rows, H):
    """model of a persistent GRU step for the schedule below, fitted to the measured forward steps (5.8 / 7.7 /
    12.7 / 18.9 us at 1.5 / 3 / 6 / 12 us of MFMA): the hand-off plus the cell's fp32 MFMA time -- rows x 48 gate
    columns x H on one CU (4 SIMDs x 1024 MAC per 32 cycles, 2.4 GHz) -- which the hand-off only partly hides"""
    return 4.2 + 1.25 * rows * H * 48 / (128.0 * 2400.0)


def _layout_waves(Bs, Hs, rows, backward):
    """waves per workgroup of the layout the library would choose for such a launch (0 = not resident)"""
    n = len(Bs)
    arr = lambda v: (C.c_int32 * n)(*v)
    out = C.c_int(0)
    L.check(L.load().dssvae_gru_layout(n, arr(Bs), arr(Hs), arr(rows), backward, L.CELL_GRU, C.byref(out)),
            "dssvae_gru_layout")
    return out.value


def _schedule(specs):
    """The launches of one packed group of persistent GRU recurrences, in forward order:
    [[(position in specs, t0, t1, rows)]] -- every launch runs steps [t0, t1) of its members in cells of `rows`
    batch rows (0 = the library chooses one height for the launch).  specs = [(B, H, T)].

    The time cuts are those of `_segments`.  On top of them, for the first cut: (1) a recurrence that ends at the
    cut may run WHOLE in the second launch instead, beside the tails of the long ones (independent recurrences
    of a launch do not wait for each other), which frees its workgroups in the first launch; (2) the members of a
    launch may have different cell heights, so that the launch's MFMA-heaviest recurrence -- the one that sets its
    duration -- gets the finer cells.  Both are chosen by exhaustive search (<= 8 members) under the residency
    bound of the 16-wave layouts, one workgroup per CU, minimising the summed model time of the two launches.
    BASELINE configs[1]'s decoder layer [H600 T31 | 2 x H300 T91 | 2 x H300 T31] becomes
    [H600 in 64-row cells + the long pair's first 31 steps in 128-row cells] then [the long pair's tails in 64-row
    cells + the short pair whole in 128-row cells]: 228 workgroups each, and the first launch is no longer bound
    by 12 us of MFMA per step in the H600 cells."""
    env = _os.environ
    key = (tuple(specs), GRU_SEGMENT_MIN, GRU_SCHEDULE, FORCE_STEPWISE_GRU, env.get("DSSVAE_GRU_KSPLIT"),
           env.get("DSSVAE_GRU_WAVES"))
    hit = _SCHEDULE_CACHE.get(key)
    if hit is not None:
        return hit
    segs = _segments([T for _, _, T in specs])
    plan = [[(j, t0, min(t1, specs[j][2]), 0) for j in members] for t0, t1, members in segs]
    tunable = (GRU_SCHEDULE and len(segs) >= 2 and len(specs) <= 8 and not FORCE_STEPWISE_GRU and
               env.get("DSSVAE_GRU_KSPLIT", "1") != "0" and not env.get("DSSVAE_GRU_WAVES") and
               # the 16-wave layouts with 64-row cells: the W_hh slice, 12 partial-sum slots of 3 KB, the sync words
               all(3 * ((H + 15) // 16) * 1024 + 12 * 3 * 1024 + 256 <= 160 * 1024 for _, H, _ in specs))
    if tunable:
        cus = n_cus()
        cut = segs[0][1]
        short = [j for j, (_, _, T) in enumerate(specs) if T <= cut]
        span1 = segs[1][1] - segs[1][0]

        def best_rows(members, steps):
            """cell heights (64 / 128 per member) of one launch that minimise its model time -> (time, rows)"""
            best = None
            for mask in range(1 << len(members)):
                rows = [64 if (mask >> k) & 1 else 128 for k in range(len(members))]
                wg = sum(-(-specs[j][0] // r) * -(-specs[j][1] // 16) for j, r in zip(members, rows))
                if wg > cus:
                    continue
                t = max(n * _step_us(r, specs[j][1]) for j, r, n in zip(members, rows, steps))
                if best is None or (t, wg) < best[:2]:
                    best = (t, wg, rows)
            return best

        base0 = best_rows(segs[0][2], [min(cut, specs[j][2]) for j in segs[0][2]])
        base1 = best_rows(segs[1][2], [min(segs[1][1], specs[j][2]) - cut for j in segs[1][2]])
        best = None if base0 is None or base1 is None else (base0[0] + base1[0], (), base0[2], base1[2])
        for mask in range(1, 1 << len(short)):
            moved = [j for k, j in enumerate(short) if (mask >> k) & 1]
            m0 = [j for j in segs[0][2] if j not in moved]
            m1 = list(segs[1][2]) + moved
            if not m0:
                continue
            r0 = best_rows(m0, [min(cut, specs[j][2]) for j in m0])
            r1 = best_rows(m1, [min(segs[1][1], specs[j][2]) - cut if j not in moved else specs[j][2] for j in m1])
            if r0 is None or r1 is None:
                continue
            # a moved recurrence must not outlast the tails it runs beside (it would lengthen the second launch)
            if max(specs[j][2] * _step_us(r, specs[j][1]) for j, r in zip(m1, r1[2]) if j in moved) > \
                    max(span1 * _step_us(r, specs[j][1]) for j, r in zip(m1, r1[2]) if j not in moved) * 1.05:
                continue
            if best is None or r0[0] + r1[0] < best[0] - 1e-9:
                best = (r0[0] + r1[0], tuple(moved), r0[2], r1[2])
        if best is not None:
            _, moved, rows0, rows1 = best
            m0 = [j for j in segs[0][2] if j not in moved]
            m1 = list(segs[1][2]) + list(moved)
            tuned = [[(j, 0, min(cut, specs[j][2]), r) for j, r in zip(m0, rows0)],
                     [(j, 0 if j in moved else cut, specs[j][2] if j in moved else min(segs[1][1], specs[j][2]), r)
                      for j, r in zip(m1, rows1)]]
            # the library has the last word (occupancy of the kernels, LDS): both directions of both launches
            # must come out in the 16-wave layouts the model assumes, or the plan stays with the time cuts
            if all(_layout_waves([specs[j][0] for j, _, _, _ in e], [specs[j][1] for j, _, _, _ in e],
                                 [r for _, _, _, r in e], bwd) == 16 for e in tuned for bwd in (0, 1)):
                plan[0], plan[1] = tuned
    _SCHEDULE_CACHE[key] = plan
    return plan


_PERSISTENT_ERR = {}


def persistent_error_word(device):
    """sticky per-device int32: number of persistent launches that timed out since the last check"""
    key = torch.device(device).index
    if key is None:
        key = torch.cuda.current_device()
    if key not in _PERSISTENT_ERR:
        _PERSISTENT_ERR[key] = torch.zeros(1, device=torch.device("cuda", key), dtype=torch.int32)
        torch.cuda.current_stream().synchronize()  # once: every stream may use the word from now on
        # the library's persistent kernels raise it themselves from now on (no fold-in launch per recurrence)
        with torch.cuda.device(key):
            L.check(L.load().dssvae_gru_error_word(_PERSISTENT_ERR[key].data_ptr()), "dssvae_gru_error_word")
    return _PERSISTENT_ERR[key]


_ZERO_IDX = {}


def _note_persistent_error(wsb):
    """make sure the device's sticky error word exists and is registered with the library BEFORE the launch: a
    persistent kernel whose bounded spin gives up sets it (besides the first int32 of its own workspace), without
    a host sync; `check_persistent_errors` reads it, and the optimiser's update kernel reads it on device (a
    timed-out step never reaches the parameters)."""
    persistent_error_word(wsb.device)


# Persistent launches never overlap EACH OTHER, whatever streams they are on: two partly resident grids can starve
# one another of the CUs (LDS) the rest of their workgroups need, and each would spin until its limit.  Every
# persistent launch waits for the previous one's completion event and leaves its own (GEMMs and the other kernels
# of other streams still run beside a recurrence: that overlap is what the streams are for).
_PERSISTENT_DONE = {}


class _PersistentTurn(object):
    def __init__(self, dev):
        self.key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()

    def __enter__(self):
        prev = _PERSISTENT_DONE.get(self.key)
        if prev is not None:
            torch.cuda.current_stream().wait_event(prev)
        return self

    def __exit__(self, *exc):
        ev = _event()
        ev.record()
        _PERSISTENT_DONE[self.key] = ev
        return False


class ResidentNeighbour(object):
    """residency rehearsal (dssvae_hold_cus): `workgroups` x 256 threads holding `lds_bytes` each stay on the chip,
    on a stream of their own, until release() -- or max_ms, whichever comes first.  `held_ms()` afterwards = how long
    each workgroup stayed."""

    def __init__(self, workgroups=48, lds_bytes=32 * 1024, max_ms=2000, device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.ticks = torch.zeros(workgroups, device=dev, dtype=torch.int64)
        self.stream = torch.cuda.Stream(device=dev)
        with torch.cuda.device(dev):
            L.check(L.load().dssvae_hold_cus(self.flag.data_ptr(), workgroups, lds_bytes, max_ms,
                                             self.ticks.data_ptr(), self.stream.cuda_stream), "dssvae_hold_cus")

    def release(self):
        self.flag[0] = 1

    def held_ms(self):
        self.stream.synchronize()
        return (self.ticks.cpu().double() / 1e5).tolist()


def check_persistent_errors():
    """raise if any persistent GRU launch since the last call timed out waiting for a peer workgroup
    (its results are then garbage).  One device sync: call it where the loop synchronises anyway."""
    for key, t in _PERSISTENT_ERR.items():
        n = int(t.item())
        if n:
            t.zero_()
            raise RuntimeError("a persistent GRU launch on device %s gave up waiting for a peer workgroup "
                               "(DSSVAE_ETIMEOUT) since the last check: the grid was not co-resident" % (key,))


def _workspace(Bs, Hs, dev):
    n = len(Bs)
    nbytes = L.load().dssvae_gru_workspace_bytes(n, (C.c_int32 * n)(*Bs), (C.c_int32 * n)(*Hs))
    if nbytes <= 0:
        raise RuntimeError("dssvae_gru_workspace_bytes rejected the instance list")
    return torch.empty(nbytes, device=dev, dtype=torch.uint8), nbytes


def _xproj_ld(x):
    B, T, G = x.shape
    assert x.stride(2) == 1 and (B == 1 or x.stride(0) == T * x.stride(1)), (x.shape, x.stride())
    return x.stride(1) if T > 1 else max(x.stride(1), G)


class _GruFn(Function):
    """meta = dict(n, group[], col[], reverse[], lengths, group_width[], save) ; tensors = per instance
    (xproj, whh, bhh, h0-or-empty).  Instances may differ in B, T and H; instances of one output
    group share B and T.  Returns (*group_outs, *h_last)."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        ctx.set_materialize_grads(False)
        n = meta["n"]
        xs, ws, bs, h0s = tensors[0::5], tensors[1::5], tensors[2::5], tensors[3::5]  # [4::5]: b_ih, gradient routing only
        L.require_cuda(*xs)
        dev = xs[0].device
        BT = [(x.size(0), x.size(1)) for x in xs]
        Hs = [w.size(1) for w in ws]
        lengths = meta["lengths"]
        outs = []
        for g, wdt in enumerate(meta["group_width"]):
            Bg, Tg = BT[meta["group"].index(g)]
            outs.append(torch.empty(Bg, Tg, wdt, device=dev, dtype=torch.float32))
        hl = [torch.empty(BT[i][0], H, device=dev, dtype=torch.float32) for i, H in enumerate(Hs)]
        save = meta["save"]  # per instance
        gates = [torch.empty(BT[i][0], BT[i][1], 5 * H, device=dev, dtype=torch.float32) if save[i] else None
                 for i, H in enumerate(Hs)]
        ws = [w.contiguous() for w in ws]
        bs = [b.contiguous() for b in bs]
        h0s = [None if h.numel() == 0 else h.contiguous() for h in h0s]
        lib = L.load()
        fit = [_fits_persistent(BT[i][0], Hs[i]) for i in range(n)]
        ctx.fit = fit
        for i in range(n):
            if not fit[i]:
                o = outs[meta["group"][i]]
                assert (o.size(0), o.size(1)) == BT[i]
                _gru_steps_fwd(xs[i], ws[i], bs[i], h0s[i], lengths, bool(meta["reverse"][i]), o, meta["col"][i],
                               hl[i], gates[i])
        pers = [i for i in range(n) if fit[i]]
        for launch_ in _pack([(BT[i][0], Hs[i], BT[i][1]) for i in pers]):
            whole = [pers[k] for k in launch_]
            for entries in _schedule([(BT[i][0], Hs[i], BT[i][1]) for i in whole]):
                launch = [whole[j] for j, _, _, _ in entries]
                arr = (L.GruFwdInst * len(launch))()
                for k, (i, (_, t0, t1, rows)) in enumerate(zip(launch, entries)):
                    o = outs[meta["group"][i]]
                    assert (o.size(0), o.size(1)) == BT[i]
                    # a later segment continues from the state the previous one left in hl[i] (read and written
                    # by the same lane: in place)
                    state = h0s[i] if t0 == 0 else hl[i]
                    arr[k] = L.GruFwdInst(
                        xs[i].data_ptr(), _xproj_ld(xs[i]), ws[i].data_ptr(), bs[i].data_ptr(),
                        None if state is None else state.data_ptr(),
                        None if lengths is None else lengths.data_ptr(),
                        o.data_ptr() + 4 * meta["col"][i], o.size(2), hl[i].data_ptr(),
                        None if gates[i] is None else gates[i].data_ptr(), BT[i][0], t1 - t0, Hs[i],
                        int(meta["reverse"][i]), t0, BT[i][1], rows)
                wsb, nbytes = _workspace([BT[i][0] for i in launch], [Hs[i] for i in launch], dev)
                _note_persistent_error(wsb)
                steps = [t1 - t0 for _, t0, t1, _ in entries]
                fl = sum(6.0 * BT[i][0] * n_t * Hs[i] * Hs[i] for i, n_t in zip(launch, steps))
                by = sum(4.0 * BT[i][0] * n_t * Hs[i] * (9 if save[i] else 4) for i, n_t in zip(launch, steps))
                with _PersistentTurn(dev), _Span("gru_fwd_kernel", fl, by, max(steps), detail=" ".join(
                        "B%dT%dH%d" % (BT[i][0], n_t, Hs[i]) for i, n_t in zip(launch, steps))):
                    L.check(lib.dssvae_gru_fwd(arr, len(launch), L.ptr(wsb), nbytes, L.stream()), "dssvae_gru_fwd")
                if meta.get("check"):
                    L.check(lib.dssvae_gru_check(L.ptr(wsb), L.stream()), "dssvae_gru_fwd (persistent sync)")
        ctx.meta, ctx.Hs, ctx.BT = meta, Hs, BT
        ctx.bias_refs = (list(tensors[2::5]), list(tensors[4::5]))  # identify the parameters for _claim
        ctx.h0_none = [h is None for h in h0s]
        ctx.save_for_backward(*ws, *[h for h in h0s if h is not None], *[g for g in gates if g is not None])
        ctx.gate_slot = [sum(1 for g in gates[:i] if g is not None) if gates[i] is not None else -1 for i in range(n)]
        return (*outs, *hl)

    @staticmethod
    def backward(ctx, *grads):
        meta, Hs, BT = ctx.meta, ctx.Hs, ctx.BT
        n = meta["n"]
        ng = len(meta["group_width"])
        saved = list(ctx.saved_tensors)
        ws = saved[:n]
        n_h0 = sum(1 for x in ctx.h0_none if not x)
        gsaved = saved[n + n_h0:]
        gates = [gsaved[k] if k >= 0 else None for k in ctx.gate_slot]
        dev = ws[0].device
        douts = [None if g is None else g.contiguous() for g in grads[:ng]]
        dhl = [None if g is None else g.contiguous() for g in grads[ng:]]
        lengths = meta["lengths"]
        # instances that saved no history (forward-only heads) take no part in the backward launches
        live = [i for i in range(n) if gates[i] is not None and
                (douts[meta["group"][i]] is not None or dhl[i] is not None)]
        dgx, dgh, dh0 = [None] * n, [None] * n, [None] * n
        for i in live:
            dgx[i] = torch.empty(BT[i][0], BT[i][1], 3 * Hs[i], device=dev, dtype=torch.float32)
            dgh[i] = torch.empty(BT[i][0], BT[i][1], 3 * Hs[i], device=dev, dtype=torch.float32)
            dh0[i] = torch.empty(BT[i][0], Hs[i], device=dev, dtype=torch.float32)
        # bias gradients (column sums of dgx / dgh) are accumulated by the persistent kernel itself
        need_bx = [i in live and ctx.needs_input_grad[5 + 5 * i] for i in range(n)]
        need_bh = [i in live and ctx.needs_input_grad[3 + 5 * i] for i in range(n)]
        # (accumulated with atomics: a claimed slice of the flat gradient buffer is zero at this point, see _claim)
        bhh_refs, bih_refs = ctx.bias_refs

        def _bias_buf(ref, H):
            g = _claim(ref)
            return g if g is not None else torch.zeros(3 * H, device=dev, dtype=torch.float32)

        dbx = [_bias_buf(bih_refs[i], Hs[i]) if need_bx[i] else None for i in range(n)]
        dbh = [_bias_buf(bhh_refs[i], Hs[i]) if need_bh[i] else None for i in range(n)]
        lib = L.load()
        for i in live:
            if not ctx.fit[i]:
                _gru_steps_bwd(ws[i], gates[i], lengths, bool(meta["reverse"][i]), douts[meta["group"][i]],
                               meta["col"][i], dhl[i], dgx[i], dgh[i], dh0[i])
                rows = BT[i][0] * BT[i][1]
                if dbx[i] is not None:
                    colsum(dgx[i].view(rows, 3 * Hs[i]), out=dbx[i])
                if dbh[i] is not None:
                    colsum(dgh[i].view(rows, 3 * Hs[i]), out=dbh[i])
        live_p = [i for i in live if ctx.fit[i]]
        for launch_ in _pack([(BT[i][0], Hs[i], BT[i][1]) for i in live_p]):
            whole = [live_p[k] for k in launch_]
            # the forward's segments, last first: d(state) entering a segment is what the later one left in dh0[i]
            for entries in reversed(_schedule([(BT[i][0], Hs[i], BT[i][1]) for i in whole])):
                launch = [whole[j] for j, _, _, _ in entries]
                arr = (L.GruBwdInst * len(launch))()
                for k, (i, (_, t0, t1, rows)) in enumerate(zip(launch, entries)):
                    do = douts[meta["group"][i]]
                    dstate = dhl[i] if t1 >= BT[i][1] else dh0[i]
                    arr[k] = L.GruBwdInst(
                        ws[i].data_ptr(), None, None if lengths is None else lengths.data_ptr(), None, 0,
                        gates[i].data_ptr(), None if do is None else do.data_ptr() + 4 * meta["col"][i],
                        0 if do is None else do.size(2), None if dstate is None else dstate.data_ptr(),
                        dgx[i].data_ptr(), dgh[i].data_ptr(), dh0[i].data_ptr(), BT[i][0], t1 - t0,
                        Hs[i], int(meta["reverse"][i]), None if dbx[i] is None else dbx[i].data_ptr(),
                        None if dbh[i] is None else dbh[i].data_ptr(), t0, BT[i][1], rows)
                wsb, nbytes = _workspace([BT[i][0] for i in launch], [Hs[i] for i in launch], dev)
                _note_persistent_error(wsb)
                steps = [t1 - t0 for _, t0, t1, _ in entries]
                fl = sum(6.0 * BT[i][0] * n_t * Hs[i] * Hs[i] for i, n_t in zip(launch, steps))
                by = sum(4.0 * BT[i][0] * n_t * Hs[i] * 12 for i, n_t in zip(launch, steps))
                with _PersistentTurn(dev), _Span("gru_bwd_kernel", fl, by, max(steps), detail=" ".join(
                        "B%dT%dH%d" % (BT[i][0], n_t, Hs[i]) for i, n_t in zip(launch, steps))):
                    L.check(lib.dssvae_gru_bwd(arr, len(launch), L.ptr(wsb), nbytes, L.stream()), "dssvae_gru_bwd")
                if meta.get("check"):
                    L.check(lib.dssvae_gru_check(L.ptr(wsb), L.stream()), "dssvae_gru_bwd (persistent sync)")
        # recurrent weight gradients dW_hh = dgh^T . h_{t-1} of all instances in grouped launches
        iw = [i for i in range(n) if dgx[i] is not None and ctx.needs_input_grad[2 + 5 * i]]
        pairs = []
        for i in iw:
            rows = BT[i][0] * BT[i][1]
            pairs.append((dgh[i].view(rows, 3 * Hs[i]), gates[i].view(rows, 5 * Hs[i])[:, 4 * Hs[i]:]))
        dws = dict(zip(iw, _gemm_grouped("tn", pairs, outs_given=[_claim(ws[i]) for i in iw]))) if iw else {}
        out = [None]
        for i in range(n):
            if dgx[i] is None:
                out += [None, None, None, None, None]
                continue
            need = ctx.needs_input_grad[1 + 5 * i: 6 + 5 * i]
            out += [dgx[i] if need[0] else None, dws.get(i), dbh[i], dh0[i] if (need[3] and not ctx.h0_none[i]) else None,
                    dbx[i]]
        return tuple(out)


def gru_layer(insts, lengths=None, save=None, check=False):
    """Run independent GRU recurrences in as few persistent launches as fit the chip.

    insts: list of dicts {xproj [B,T,3H] (may be a column view), whh [3H,H], bhh [3H], h0 [B,H] or None,
    bih [3H] or None (NOT used in forward -- xproj already holds it: when given, the caller has computed xproj
    with a detached b_ih and the gradient of b_ih, the column sums of d(xproj), comes from the recurrence's
    backward kernel instead of a separate reduction pass), reverse: bool, group: int, col: int, save: bool (default: the `save` argument; False = forward-only
    instance, no history kept)}.  Instances of one group write disjoint column ranges
    [col, col+H) of that group's output [B,T,width].  lengths: int32 device tensor [B] or None.
    Returns (list of group outputs, list of per-instance final states)."""
    n = len(insts)
    if save is None:
        save = torch.is_grad_enabled()
    groups = sorted(set(i["group"] for i in insts))
    width = [0] * len(groups)
    for i in insts:
        g = groups.index(i["group"])
        width[g] = max(width[g], i["col"] + i["whh"].size(1))
    dev = insts[0]["xproj"].device
    empty = torch.empty(0, device=dev)
    meta = dict(n=n, group=[groups.index(i["group"]) for i in insts], col=[i["col"] for i in insts],
                reverse=[bool(i.get("reverse", False)) for i in insts], lengths=lengths, group_width=width,
                save=[bool(i.get("save", save)) for i in insts], check=check)
    flat = []
    for i in insts:
        flat += [i["xproj"], i["whh"], i["bhh"], empty if i.get("h0") is None else i["h0"],
                 empty if i.get("bih") is None else i["bih"]]
    res = _GruFn.apply(meta, *flat)
    return list(res[:len(groups)]), list(res[len(groups):])


# ----------------------------------------------------------------------------------------------
# LSTM (rnn_cell = "lstm" of the reference's RNNBase).  A recurrence whose W_hh slice (4 gates x 16 units x H) fits the
# persistent kernel's LDS budget runs through the same persistent kernels as the GRU (cell = LSTM: csrc/gru.hip);
# larger ones stepwise -- per timestep the recurrent product on the MFMA GEMM + one fused cell kernel
# (dssvae_lstm_step_*).  Both write the same saved history, so either backward reads either forward's.
# ----------------------------------------------------------------------------------------------
def _lstm_fits_persistent(B, H):
    if FORCE_STEPWISE_GRU:
        return False
    return _cells(B, H) <= n_cus() and 4 * ((H + 15) // 16) * 1024 <= GRU_LDS_BUDGET


class _LstmFn(Function):
    """(xproj [B,T,4H] incl. b_ih, whh [4H,H], bhh [4H], h0 [B,H], c0 [B,H], bih [4H] or empty -- gradient routing
    only) -> (out [B,T,H], h_T, c_T); gate order i, f, g, o; lengths: packed-sequence semantics; reverse: walks
    t = T-1 .. 0."""

    @staticmethod
    def forward(ctx, x, whh, bhh, h0, c0, bih, lengths, reverse):
        L.require_cuda(x, whh)
        ctx.set_materialize_grads(False)
        B, T, G = x.shape
        H = whh.size(1)
        assert G == 4 * H and whh.size(0) == 4 * H
        dev = x.device
        lib = L.load()
        save = torch.is_grad_enabled() or any(ctx.needs_input_grad[:6])
        whh, bhh = whh.contiguous(), bhh.contiguous()
        ldx = _xproj_ld(x)
        out = torch.empty(B, T, H, device=dev, dtype=torch.float32)
        gates = torch.empty(B, T, 7 * H, device=dev, dtype=torch.float32) if save else None
        h, c = h0.contiguous(), c0.contiguous()
        lp = None if lengths is None else lengths.data_ptr()
        ctx.persistent = _lstm_fits_persistent(B, H)
        if ctx.persistent:
            hl, cl = torch.empty_like(h), torch.empty_like(c)
            arr = (L.GruFwdInst * 1)()
            arr[0] = L.GruFwdInst(x.data_ptr(), ldx, whh.data_ptr(), bhh.data_ptr(), h.data_ptr(), lp, out.data_ptr(), H,
                                  hl.data_ptr(), None if gates is None else gates.data_ptr(), B, T, H, int(bool(reverse)),
                                  0, T, 0, L.CELL_LSTM, c.data_ptr(), cl.data_ptr())
            wsb, nbytes = _workspace([B], [H], dev)
            _note_persistent_error(wsb)
            with _PersistentTurn(dev), _Span("lstm_fwd_kernel", 8.0 * B * T * H * H,
                                             4.0 * B * T * H * (11 if save else 5), T, detail="B%dT%dH%d" % (B, T, H)):
                L.check(lib.dssvae_gru_fwd(arr, 1, L.ptr(wsb), nbytes, L.stream()), "dssvae_gru_fwd (lstm)")
            h, c = hl, cl
        gh = None if ctx.persistent else torch.empty(B, 4 * H, device=dev, dtype=torch.float32)
        for step in range(0 if ctx.persistent else T):
            t = T - 1 - step if reverse else step
            mm_nt(h, whh, out=gh)
            hn, cn = torch.empty_like(h), torch.empty_like(c)
            L.check(lib.dssvae_lstm_step_fwd(
                x.data_ptr() + 4 * t * ldx, T * ldx, L.ptr(gh), L.ptr(bhh), L.ptr(h), L.ptr(c), lp, t, B, H, L.ptr(hn),
                L.ptr(cn), out.data_ptr() + 4 * t * H, T * H,
                None if gates is None else gates.data_ptr() + 4 * t * 7 * H, T * 7 * H, L.stream()),
                "dssvae_lstm_step_fwd")
            h, c = hn, cn
        ctx.save_for_backward(whh, gates)
        ctx.lengths, ctx.reverse, ctx.dims = lengths, bool(reverse), (B, T, H)
        ctx.refs = (whh, bhh, bih)
        return out, h, c

    @staticmethod
    def backward(ctx, dout, dhT, dcT):
        whh, gates = ctx.saved_tensors
        B, T, H = ctx.dims
        if (dout is None and dhT is None and dcT is None) or gates is None:
            return (None,) * 8
        dev = whh.device
        lib = L.load()
        dout = None if dout is None else dout.contiguous()
        dg = torch.empty(B, T, 4 * H, device=dev, dtype=torch.float32)
        dh = None if dhT is None else dhT.contiguous()
        dc = None if dcT is None else dcT.contiguous()
        lp = None if ctx.lengths is None else ctx.lengths.data_ptr()
        need = ctx.needs_input_grad
        db = None
        if ctx.persistent:
            dh0 = torch.empty(B, H, device=dev, dtype=torch.float32)
            dc0 = torch.empty(B, H, device=dev, dtype=torch.float32)
            if need[2] or need[5]:  # accumulated by the kernel (atomics): starts at zero
                db = torch.zeros(4 * H, device=dev, dtype=torch.float32)
            arr = (L.GruBwdInst * 1)()
            arr[0] = L.GruBwdInst(whh.data_ptr(), None, lp, None, 0, gates.data_ptr(),
                                  None if dout is None else dout.data_ptr(), 0 if dout is None else H,
                                  None if dh is None else dh.data_ptr(), dg.data_ptr(), None, dh0.data_ptr(), B, T, H,
                                  int(ctx.reverse), None if db is None else db.data_ptr(), None, 0, T, 0, L.CELL_LSTM,
                                  None if dc is None else dc.data_ptr(), dc0.data_ptr())
            wsb, nbytes = _workspace([B], [H], dev)
            _note_persistent_error(wsb)
            with _PersistentTurn(dev), _Span("lstm_bwd_kernel", 8.0 * B * T * H * H, 4.0 * B * T * H * 12, T,
                                             detail="B%dT%dH%d" % (B, T, H)):
                L.check(lib.dssvae_gru_bwd(arr, 1, L.ptr(wsb), nbytes, L.stream()), "dssvae_gru_bwd (lstm)")
            dh, dc = dh0, dc0
        for step in range(0 if ctx.persistent else T):
            fstep = T - 1 - step
            t = T - 1 - fstep if ctx.reverse else fstep
            dhz = torch.empty(B, H, device=dev, dtype=torch.float32)
            dcp = torch.empty(B, H, device=dev, dtype=torch.float32)
            L.check(lib.dssvae_lstm_step_bwd(
                gates.data_ptr() + 4 * t * 7 * H, T * 7 * H, None if dout is None else dout.data_ptr() + 4 * t * H,
                0 if dout is None else T * H, L.ptr(dh), L.ptr(dc), lp, t, B, H, dg.data_ptr() + 4 * t * 4 * H, T * 4 * H,
                L.ptr(dhz), L.ptr(dcp), L.stream()), "dssvae_lstm_step_bwd")
            mm_nn(dg[:, t, :], whh, out=dhz, beta=1.0)
            dh, dc = dhz, dcp
        W_ref, bhh_ref, bih_ref = ctx.refs
        rows = B * T
        dg2 = dg.view(rows, 4 * H)
        dW = mm_tn(dg2, gates.view(rows, 7 * H)[:, 5 * H:6 * H], out=_claim(W_ref, need[1])) if need[1] else None
        if db is None and (need[2] or need[5]):
            db = colsum(dg2)
        return (dg if need[0] else None, dW, db if need[2] else None, dh if need[3] else None, dc if need[4] else None,
                db if (need[5] and bih_ref.numel()) else None, None, None)


def lstm_layer(xproj, whh, bhh, h0=None, c0=None, bih=None, lengths=None, reverse=False):
    """one LSTM recurrence over precomputed input projections -> (out [B,T,H], h_T, c_T)"""
    B, H = xproj.size(0), whh.size(1)
    z = None
    if h0 is None or c0 is None:
        z = torch.zeros(B, H, device=xproj.device, dtype=torch.float32)
    empty = torch.empty(0, device=xproj.device)
    return _LstmFn.apply(xproj, whh, bhh, z if h0 is None else h0, z if c0 is None else c0,
                         empty if bih is None else bih, lengths, bool(reverse))


def lstm_cell(gi, gh, c):
    """one LSTM cell step from gi = x W_ih^T + b_ih and gh = h W_hh^T + b_hh (greedy loop) -> (h', c')"""
    L.require_cuda(gi, gh, c)
    B, H = c.shape
    hn, cn = torch.empty_like(c), torch.empty_like(c)
    zero_b = torch.zeros(4 * H, device=c.device, dtype=torch.float32)
    gi, gh, c = gi.contiguous(), gh.contiguous(), c.contiguous()
    L.check(L.load().dssvae_lstm_step_fwd(L.ptr(gi), 4 * H, L.ptr(gh), L.ptr(zero_b), None, L.ptr(c), None, 0, B, H,
                                          L.ptr(hn), L.ptr(cn), None, 0, None, 0, L.stream()), "dssvae_lstm_step_fwd")
    return hn, cn


def gru_cell(gi, gh, h):
    L.require_cuda(gi, gh, h)
    out = torch.empty_like(h)
    L.check(L.load().dssvae_gru_cell(L.ptr(gi.contiguous()), L.ptr(gh.contiguous()), L.ptr(h.contiguous()), h.size(0),
                                     h.size(1), L.ptr(out), L.stream()), "dssvae_gru_cell")
    return out


def greedy_decode(table, layers, w_out, b_out, h0, sos_id, max_len, c0=None):
    """whole greedy loop in one native call (dssvae_greedy_decode / dssvae_greedy_decode_lstm when c0 is given):
    -> (tokens [max_len,B] int64, h_final [layers,B,H]) or (tokens, (h_final, c_final)).
    layers: list of (w_ih, w_hh, b_ih, b_hh)."""
    L.require_cuda(table, w_out, h0)
    n_layers, B, H = h0.shape
    E, V = table.size(1), w_out.size(0)
    G = 3 if c0 is None else 4
    assert len(layers) == n_layers <= L.DECODE_MAX_LAYERS and w_out.size(1) == H + E
    keep = [_f32(table), _f32(w_out)] + [None if b_out is None else b_out.contiguous()]
    prm = L.DecodeParams()
    prm.embedding, prm.ld_embedding, prm.n_embeddings = table.data_ptr(), table.stride(0), table.size(0)
    for l, ps in enumerate(layers):
        ps = [_f32(t).contiguous() for t in ps]
        keep += ps
        assert ps[0].shape == (G * H, E if l == 0 else H) and ps[1].shape == (G * H, H)
        prm.w_ih[l], prm.w_hh[l], prm.b_ih[l], prm.b_hh[l] = (t.data_ptr() for t in ps)
    prm.w_out, prm.ld_out = w_out.data_ptr(), w_out.stride(0)
    prm.b_out = None if keep[2] is None else keep[2].data_ptr()
    prm.E, prm.H, prm.V, prm.n_layers = E, H, V, n_layers
    S = _pick_split(B, V)
    lib = L.load()
    size_fn = lib.dssvae_greedy_decode_workspace_bytes if c0 is None else lib.dssvae_greedy_decode_lstm_workspace_bytes
    nbytes = size_fn(B, E, H, n_layers, S, V)
    if nbytes <= 0:
        raise RuntimeError("dssvae_greedy_decode_workspace_bytes rejected the shape")
    dev = h0.device
    wsb = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    tokens = torch.empty(max_len, B, device=dev, dtype=torch.int64)
    h_fin = torch.empty(n_layers, B, H, device=dev, dtype=torch.float32)
    h0 = _f32(h0).contiguous()
    if c0 is None:
        L.check(lib.dssvae_greedy_decode(C.byref(prm), L.ptr(h0), B, int(max_len), int(sos_id), S, L.ptr(tokens),
                                         L.ptr(h_fin), L.ptr(wsb), nbytes, L.stream()), "dssvae_greedy_decode")
        return tokens, h_fin
    assert c0.shape == h0.shape
    c0 = _f32(c0).contiguous()
    c_fin = torch.empty_like(h_fin)
    L.check(lib.dssvae_greedy_decode_lstm(C.byref(prm), L.ptr(h0), L.ptr(c0), B, int(max_len), int(sos_id), S,
                                          L.ptr(tokens), L.ptr(h_fin), L.ptr(c_fin), L.ptr(wsb), nbytes, L.stream()),
            "dssvae_greedy_decode_lstm")
    return tokens, (h_fin, c_fin)


# ----------------------------------------------------------------------------------------------
# flat-buffer optimiser pieces
# ----------------------------------------------------------------------------------------------
_SUMSQ_WS = {}


def sumsq(flat, out):
    """out[0] += sum(flat^2), in a fixed order (bit-identical on every data-parallel replica)"""
    L.require_cuda(flat, out)
    ws = _SUMSQ_WS.get(flat.device)
    if ws is None:
        ws = _SUMSQ_WS[flat.device] = torch.empty(L.SUMSQ_PARTIALS, device=flat.device)
    L.check(L.load().dssvae_sumsq(L.ptr(flat), flat.numel(), L.ptr(ws), L.ptr(out), L.stream()), "dssvae_sumsq")


def adam_step_dyn(p, g, m, v, hyper_ptr, beta1, beta2, eps, sumsq_t, max_norm, skip_flag=None):
    """Adam over one range with (lr, 1 - beta1^t, sqrt(1 - beta2^t)) read from three per-step device floats"""
    L.check(L.load().dssvae_adam_step_dyn(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), hyper_ptr, beta1, beta2,
                                          eps, L.ptr(sumsq_t), max_norm, L.ptr(skip_flag), L.stream()),
            "dssvae_adam_step_dyn")


def batch_rows(flat_ids, offsets, idx, width, boundary, entry, out=None, lengths_out=None):
    """rows of a minibatch cut from the device-resident id cache (dssvae_batch_rows) -> int64 [B, width]"""
    L.require_cuda(flat_ids, offsets, idx)
    B = idx.numel()
    if out is None:
        out = torch.empty(B, width, device=idx.device, dtype=torch.int64)
    assert out.shape == (B, width) and out.is_contiguous() and out.dtype == torch.int64
    L.check(L.load().dssvae_batch_rows(L.ptr(flat_ids), L.ptr(offsets), L.ptr(idx), B, width, int(boundary),
                                       entry.sos_id, entry.eos_id, entry.pad_id, L.ptr(out), L.ptr(lengths_out),
                                       L.stream()), "dssvae_batch_rows")
    return out


def keep_rows(lengths_dev, T, n_keep, out=None):
    """positions of the non-<pad> targets in the flattened [B, T] layout (dssvae_keep_rows) -> int64 [n_keep]"""
    L.require_cuda(lengths_dev)
    if out is None:
        out = torch.empty(n_keep, device=lengths_dev.device, dtype=torch.int64)
    L.check(L.load().dssvae_keep_rows(L.ptr(lengths_dev), lengths_dev.numel(), int(T), L.ptr(out), int(n_keep),
                                      L.stream()), "dssvae_keep_rows")
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, sumsq_t, max_norm, skip_flag=None):
    with _Span("adam_kernel", 0.0, 28.0 * p.numel()):
        L.check(L.load().dssvae_adam_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), lr, beta1, beta2, eps,
                                          step, L.ptr(sumsq_t), max_norm, None, L.ptr(skip_flag), L.stream()),
                "dssvae_adam_step")
