"""Kernel-level parity (through the C-ABI, via dss_vae_amd.ops) against fp32/fp64 CPU math and the
oracle's explicit GRU equations.  Tolerances: fp32 MFMA is an ordered fmaf chain, so results differ
from a CPU BLAS only by summation order -> 2e-5 relative on O(1) values; losses 1e-5."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def close(a, b, rtol=2e-5, atol=2e-5, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    # fp32 accumulation error scales with the magnitude of the sums: atol is relative to max|ref|
    atol = atol * max(1.0, b.abs().max().item() if b.numel() else 1.0)
    assert torch.allclose(a, b, rtol=rtol, atol=atol), (what, err)


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (5, 7, 3), (128, 128, 32), (130, 257, 70), (256, 450, 150),
                                   (300, 1800, 300), (64, 100, 600), (513, 64, 900)])
def test_gemm_all_layouts(M, N, K):
    from dss_vae_amd import ops
    x, w, dy = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(M, N, seed=3)
    b = rnd(N, seed=4)
    xd, wd, dyd, bd = (t.to(DEV) for t in (x, w, dy, b))
    close(ops.mm_nt(xd, wd, bd), x.double() @ w.double().t() + b.double(), what="nt")
    close(ops.mm_nn(dyd, wd), dy.double() @ w.double(), what="nn")
    close(ops.mm_tn(dyd, xd), dy.double().t() @ x.double(), what="tn")
    # accumulate + strided views (leading dimension != width)
    big = rnd(M, K + 5, seed=5).to(DEV)
    xv = big[:, 2:2 + K]
    out = torch.ones(M, N, device=DEV)
    ops.mm_nt(xv, wd, None, out=out, beta=1.0)
    close(out, xv.cpu().double() @ w.double().t() + 1.0, what="nt beta view")


def test_gemm_grouped_and_linear_many():
    """several independent products per launch (dssvae_gemm_f32_grouped), mixed shapes incl. the 3H = 450 /
    H = 150 extents of the default encoder (8-byte granules), and the autograd wrapper built on it"""
    from dss_vae_amd import ops
    shapes = [(7680 // 8, 450, 300), (960, 450, 300), (130, 257, 70), (64, 100, 600), (300, 1800, 300), (5, 7, 3)]
    xs = [rnd(M, K, seed=10 + i) for i, (M, N, K) in enumerate(shapes)]
    ws = [rnd(N, K, seed=30 + i) for i, (M, N, K) in enumerate(shapes)]
    bs = [rnd(N, seed=50 + i) for i, (M, N, K) in enumerate(shapes)]
    dys = [rnd(M, N, seed=70 + i) for i, (M, N, K) in enumerate(shapes)]
    xd, wd, bd, dyd = ([t.to(DEV) for t in ts] for ts in (xs, ws, bs, dys))
    for got, x, w, b in zip(ops._gemm_grouped("nt", list(zip(xd, wd)), biases=bd), xs, ws, bs):
        close(got, x.double() @ w.double().t() + b.double(), what="grouped nt")
    for got, dy, w in zip(ops._gemm_grouped("nn", list(zip(dyd, wd))), dys, ws):
        close(got, dy.double() @ w.double(), what="grouped nn")
    for got, dy, x in zip(ops._gemm_grouped("tn", list(zip(dyd, xd))), dys, xs):
        close(got, dy.double().t() @ x.double(), what="grouped tn")
    # autograd: shared input, one frozen weight
    leaves = [[t.clone().double().requires_grad_(True) for t in (xs[0], ws[0], bs[0])],
              [None, ws[1].clone().double().requires_grad_(True), bs[1].clone().double().requires_grad_(True)]]
    ref = [leaves[0][0] @ leaves[0][1].t() + leaves[0][2], leaves[0][0] @ leaves[1][1].t() + leaves[1][2]]
    (ref[0] * dys[0].double()).sum().backward(retain_graph=True)
    (ref[1] * dys[1].double()).sum().backward()
    x0 = xs[0].clone().to(DEV).requires_grad_(True)
    w0, b0 = ws[0].clone().to(DEV).requires_grad_(True), bs[0].clone().to(DEV).requires_grad_(True)
    w1, b1 = ws[1].clone().to(DEV).requires_grad_(True), bs[1].clone().to(DEV).requires_grad_(True)
    o0, o1 = ops.linear_many([(x0, w0, b0), (x0, w1, b1)])
    close(o0, ref[0], what="many out0")
    close(o1, ref[1], what="many out1")
    ((o0 * dyd[0]).sum() + (o1 * dyd[1]).sum()).backward()
    close(x0.grad, leaves[0][0].grad, rtol=1e-4, atol=1e-5, what="many dx (summed over both uses)")
    close(w0.grad, leaves[0][1].grad, rtol=1e-4, atol=1e-5, what="many dw0")
    close(w1.grad, leaves[1][1].grad, rtol=1e-4, atol=1e-5, what="many dw1")
    close(b0.grad, leaves[0][2].grad, rtol=1e-4, atol=1e-5, what="many db0")
    close(b1.grad, leaves[1][2].grad, rtol=1e-4, atol=1e-5, what="many db1")


@pytest.mark.parametrize("kind,shapes", [
    ("tn", [(450, 150, 7680)] * 4),                                      # the encoder's recurrent weight gradients
    ("tn", [(133, 70, 5000), (450, 300, 7681), (64, 900, 4000)]),        # narrow, narrow, wide + remainder; K tails
    ("nt", [(200, 140, 20000), (90, 1000, 9000)]),                       # with a bias: added by the piece that holds k = 0
    ("nn", [(257, 130, 24000)]),
])
def test_gemm_grouped_stream_k_form(kind, shapes):
    """grouped launches with few output tiles and long K run in the stream-K form (whole tiles as plain stores, pieces as
    atomics into the zero-filled output): against fp64, and against the tile-per-workgroup form of the same launch"""
    import os
    from dss_vae_amd import ops
    pairs, refs, biases = [], [], []
    for i, (M, N, K) in enumerate(shapes):
        if kind == "nt":
            a, b = rnd(M, K, seed=i), rnd(N, K, seed=50 + i)
            ref = a.double() @ b.double().t()
        elif kind == "nn":
            a, b = rnd(M, K, seed=i), rnd(K, N, seed=50 + i)
            ref = a.double() @ b.double()
        else:
            a, b = rnd(K, M, seed=i), rnd(K, N, seed=50 + i)
            ref = a.double().t() @ b.double()
        bias = rnd(N, seed=90 + i) if kind == "nt" else None
        pairs.append((a.to(DEV), b.to(DEV)))
        biases.append(None if bias is None else bias.to(DEV))
        refs.append(ref if bias is None else ref + bias.double())
    kw = dict(biases=biases) if kind == "nt" else {}
    outs = {}
    for sk in ("1", "0"):
        os.environ["DSSVAE_GROUP_SK"] = sk
        outs[sk] = [o.clone() for o in ops._gemm_grouped(kind, pairs, **kw)]
    os.environ.pop("DSSVAE_GROUP_SK", None)
    for got, other, ref in zip(outs["1"], outs["0"], refs):
        close(got, ref, rtol=1e-4, atol=2e-5, what="stream-K " + kind)
        close(other, ref, rtol=1e-4, atol=2e-5, what="tiles " + kind)


def test_gemm_row_gather():
    from dss_vae_amd import ops
    tab, w = rnd(50, 24, seed=1).to(DEV), rnd(36, 24, seed=2).to(DEV)
    ids = torch.randint(0, 50, (77,), generator=torch.Generator().manual_seed(0)).to(DEV)
    out = torch.empty(77, 36, device=DEV)
    ops._gemm(tab, 24, 0, w, 24, 0, out, 36, None, 77, 36, 24, a_rows=ids)
    close(out, tab[ids].cpu().double() @ w.cpu().double().t(), what="gather")


def ce_reference(xh, xe, W, b, tgt, ignore):
    x = xh if xe is None else torch.cat([xh, xe], 1)
    logits = x.double() @ W.double().t() + b.double()
    logp = torch.log_softmax(logits, -1)
    tgt2 = tgt.reshape(x.size(0), -1)
    tok = -logp.gather(1, tgt2) * (tgt2 != ignore)
    return tok.sum(1)


@pytest.mark.parametrize("M,Kh,Ke,N,T", [(7, 8, 4, 11, 1), (200, 60, 20, 300, 1), (300, 120, 20, 2000, 1),
                                         (37, 16, 0, 515, 9), (256, 600, 0, 3001, 32), (130, 30, 18, 777, 1)])
def test_linear_ce_forward_backward(M, Kh, Ke, N, T):
    from dss_vae_amd import ops
    g = torch.Generator().manual_seed(M + N)
    xh, W, b = rnd(M, Kh, seed=1), rnd(N, Kh + Ke, seed=2, scale=0.3), rnd(N, seed=3, scale=0.1)
    xe = rnd(M, Ke, seed=4) if Ke else None
    tgt = torch.randint(0, N, (M, T), generator=g)
    tgt[torch.rand(M, T, generator=g) < 0.3] = 0
    tgt = tgt.squeeze(1) if T == 1 else tgt
    gout = torch.rand(M, generator=g)
    leaves = [t.clone().double().requires_grad_(True) if t is not None else None for t in (xh, xe, W, b)]
    ref = ce_reference(*leaves, tgt, 0)
    (ref * gout.double()).sum().backward()
    dl = [t.clone().to(DEV).requires_grad_(True) if t is not None else None for t in (xh, xe, W, b)]
    out = ops.linear_ce(dl[0], dl[1], dl[2], dl[3], tgt.to(DEV), 0)
    close(out, ref, rtol=1e-5, atol=1e-4, what="nll")
    (out * gout.to(DEV)).sum().backward()
    for name, a, r in zip(("dxh", "dxe", "dW", "db"), dl, leaves):
        if a is not None:
            close(a.grad, r.grad, rtol=2e-4, atol=2e-5, what=name)


@pytest.mark.parametrize("M,Kh,Ke,N,T,frozen", [(7, 8, 4, 11, 1, False), (300, 120, 20, 2000, 1, False),
                                                (37, 16, 0, 515, 9, False), (256, 600, 0, 3001, 32, True),
                                                (130, 30, 18, 777, 1, False), (520, 64, 32, 1300, 1, True),
                                                (520, 64, 32, 1300, 1, "xe_only")])
def test_linear_ce_uniform_upstream_schedule(M, Kh, Ke, N, T, frozen, monkeypatch):
    """uniform_upstream=True: logits of a row chunk go to a slab in the forward pass, gradients for
    upstream 1 are produced there and scaled in backward; same numbers as the reference when every
    row gets the same upstream gradient.  `frozen`: projection weights without grad (an adversary
    head after the freeze), several row chunks forced by a small slab."""
    from dss_vae_amd import ops
    if frozen:
        monkeypatch.setattr(ops, "SLAB_BYTES", 128 * ((N + 3) // 4 * 4) * 4)
    g = torch.Generator().manual_seed(M + N)
    xh, W, b = rnd(M, Kh, seed=1), rnd(N, Kh + Ke, seed=2, scale=0.3), rnd(N, seed=3, scale=0.1)
    xe = rnd(M, Ke, seed=4) if Ke else None
    tgt = torch.randint(0, N, (M, T), generator=g)
    tgt[torch.rand(M, T, generator=g) < 0.3] = 0
    tgt = tgt.squeeze(1) if T == 1 else tgt
    scale = 0.37
    leaves = [t.clone().double().requires_grad_(True) if t is not None else None for t in (xh, xe, W, b)]
    ref = ce_reference(*leaves, tgt, 0)
    (ref.sum() * scale).backward()
    dl = [t.clone().to(DEV).requires_grad_(True) if t is not None else None for t in (xh, xe, W, b)]
    if frozen == "xe_only":  # only the embedding panel wants a gradient
        dl[0].requires_grad_(False)
    elif frozen:
        dl[2].requires_grad_(False), dl[3].requires_grad_(False)
        if dl[1] is not None:  # a frozen adversary's embedding panel: only the h panel of dX is produced
            dl[1].requires_grad_(False)
    out = ops.linear_ce(dl[0], dl[1], dl[2], dl[3], tgt.to(DEV), 0, uniform_upstream=True)
    close(out, ref, rtol=1e-5, atol=1e-4, what="nll")
    (out.sum() * scale).backward()
    for name, a, r in zip(("dxh", "dxe", "dW", "db"), dl, leaves):
        if a is not None and a.requires_grad:
            close(a.grad, r.grad, rtol=2e-4, atol=2e-5, what=name)
        elif a is not None:
            assert a.grad is None


def test_linear_ce_small_slab_chunks(monkeypatch):
    """backward must give the same result when the dlogits slab forces many row chunks."""
    from dss_vae_amd import ops
    monkeypatch.setattr(ops, "SLAB_BYTES", 128 * 404 * 4)
    test_linear_ce_forward_backward(300, 120, 20, 401, 1)


def test_linear_argmax():
    from dss_vae_amd import ops
    xh, xe, W, b = rnd(50, 40, seed=1), rnd(50, 12, seed=2), rnd(3000, 52, seed=3), rnd(3000, seed=4)
    ref = (torch.cat([xh, xe], 1).double() @ W.double().t() + b.double()).argmax(1)
    got = ops.linear_argmax(xh.to(DEV), xe.to(DEV), W.to(DEV), b.to(DEV))
    assert torch.equal(got.cpu(), ref)


def test_embed_and_unk_replace_and_latent():
    from dss_vae_amd import ops
    import oracle
    tab = rnd(40, 300, seed=1)
    ids = torch.randint(0, 40, (6, 9), generator=torch.Generator().manual_seed(1))
    mask = (torch.rand(6 * 9, 300) > 0.3).float() / 0.7
    t = tab.clone().to(DEV).requires_grad_(True)
    out = ops.embed(t, ids.to(DEV), mask.to(DEV))
    close(out, tab[ids] * mask.view(6, 9, 300), what="embed")
    gout = rnd(6, 9, 300, seed=2)
    out.backward(gout.to(DEV))
    ref = torch.zeros(40, 300).index_add_(0, ids.reshape(-1), (gout * mask.view(6, 9, 300)).reshape(-1, 300))
    close(t.grad, ref, what="embed grad")
    # word dropout with an injected uniform draw is bit-exact vs the oracle
    ent = oracle.OVocabEntry("w%d" % i for i in range(30))
    ids = torch.randint(0, 34, (5, 12), generator=torch.Generator().manual_seed(2))
    r = torch.rand(5, 12, generator=torch.Generator().manual_seed(3))
    assert torch.equal(ops.unk_replace(ids.to(DEV), 0.4, ent, rand=r).cpu(), oracle.word_dropout(ids, 0.4, ent, r))
    assert ops.unk_replace(ids.to(DEV), 0.0, ent) is not None
    frac = (ops.unk_replace(torch.full((1000, 50), 7, device=DEV), 0.25, ent, seed=9) == 3).float().mean().item()
    assert abs(frac - 0.25) < 0.01
    # reparameterisation + KL, forward and backward
    mean, logv, eps = rnd(33, 100, seed=5), rnd(33, 100, seed=6), rnd(33, 100, seed=7)
    m64, v64 = mean.double().requires_grad_(True), logv.double().requires_grad_(True)
    z_ref = eps.double() * torch.exp(0.5 * v64) + m64
    kl_ref = -0.5 * torch.sum(1 + v64 - m64.pow(2) - v64.exp())
    gz = rnd(33, 100, seed=8)
    ((z_ref * gz.double()).sum() + 0.37 * kl_ref).backward()
    md, vd = mean.clone().to(DEV).requires_grad_(True), logv.clone().to(DEV).requires_grad_(True)
    z, kl = ops.reparam_kl(md, vd, eps.to(DEV))
    close(z, z_ref, what="z")
    close(kl, kl_ref, rtol=1e-5, what="kl")
    ((z * gz.to(DEV)).sum() + 0.37 * kl).backward()
    close(md.grad, m64.grad, what="dmean")
    close(vd.grad, v64.grad, what="dlogv")
    z2, _ = ops.reparam_kl(md, vd, None)
    close(z2, mean, what="z eval")
    mk = ops.dropout_mask((1000, 300), 0.3, 5, DEV)
    assert abs((mk == 0).float().mean().item() - 0.3) < 0.01 and abs(mk.max().item() - 1 / 0.7) < 1e-6


def gru_reference(x, h0, wih, whh, bih, bhh, lengths, reverse):
    import oracle
    return oracle.gru_loop(x, h0, wih, whh, bih, bhh, lengths=lengths, reverse=reverse)


@pytest.mark.parametrize("B,T,H,I,packed", [(3, 4, 8, 5, False), (70, 6, 30, 12, True), (64, 9, 150, 20, True),
                                            (130, 5, 300, 16, False), (256, 4, 600, 8, False),
                                            (1030, 3, 300, 8, True)])
@pytest.mark.parametrize("stepwise", [False, True, "waves8", "nosplit"])
def test_persistent_gru_forward_backward(B, T, H, I, packed, stepwise, monkeypatch):
    """two instances (forward + reverse direction) in one persistent launch -- or, stepwise=True, down the
    per-timestep GEMM + gate-kernel path that oversized recurrences take; "waves8": the persistent kernel
    with 128-row workgroups of 8 waves; "nosplit": the layouts without the K split inside a workgroup (what a grid
    that is not resident in 16-wave workgroups falls back to); the default (False) takes the 16-wave workgroups:
    4 row groups x 4 K parts, or 8 x 2 at B = 1030 -- vs explicit equations"""
    from dss_vae_amd import ops
    if stepwise == "waves8":
        monkeypatch.setenv("DSSVAE_GRU_WAVES", "8")
        stepwise = False
    if stepwise == "nosplit":
        monkeypatch.setenv("DSSVAE_GRU_KSPLIT", "0")
        stepwise = False
    monkeypatch.setattr(ops, "FORCE_STEPWISE_GRU", stepwise)
    g = torch.Generator().manual_seed(B + H)
    lengths = None
    if packed:
        lengths = sorted([int(v) for v in torch.randint(1, T + 1, (B,), generator=g)], reverse=True)
        lengths[0] = T
    x = rnd(B, T, I, seed=1)
    params, refs = [], []
    for d in range(2):
        p = [rnd(3 * H, I, seed=10 + d, scale=0.3), rnd(3 * H, H, seed=20 + d, scale=0.3),
             rnd(3 * H, seed=30 + d, scale=0.3), rnd(3 * H, seed=40 + d, scale=0.3), rnd(B, H, seed=50 + d)]
        params.append(p)
    gout = rnd(B, T, 2 * H, seed=3)
    ghl = [rnd(B, H, seed=60 + d) for d in range(2)]
    # reference in fp64
    leaves = [[t.clone().double().requires_grad_(True) for t in p] for p in params]
    x64 = x.clone().double().requires_grad_(True)
    outs, hls = [], []
    for d in range(2):
        wih, whh, bih, bhh, h0 = leaves[d]
        o, h = gru_reference(x64, h0, wih, whh, bih, bhh, lengths, reverse=bool(d))
        outs.append(o)
        hls.append(h)
    ref_out = torch.cat(outs, -1)
    ((ref_out * gout.double()).sum() + sum((h * gg.double()).sum() for h, gg in zip(hls, ghl))).backward()
    # device
    dl = [[t.clone().to(DEV).requires_grad_(True) for t in p] for p in params]
    xd = x.clone().to(DEV).requires_grad_(True)
    len_d = None if lengths is None else torch.tensor(lengths, dtype=torch.int32, device=DEV)
    insts = []
    for d in range(2):
        wih, whh, bih, bhh, h0 = dl[d]
        if d == 0:
            insts.append(dict(xproj=ops.linear(xd, wih, bih), whh=whh, bhh=bhh, h0=h0, reverse=False, group=0, col=0))
        else:  # b_ih's gradient routed through the recurrence's backward (column sums taken in the kernel)
            insts.append(dict(xproj=ops.linear(xd, wih, bih.detach()), whh=whh, bhh=bhh, bih=bih, h0=h0, reverse=True,
                              group=0, col=H))
    (out,), hl = ops.gru_layer(insts, lengths=len_d, check=True)
    close(out, ref_out, what="out")
    for d in range(2):
        close(hl[d], hls[d], what="h_last%d" % d)
    ((out * gout.to(DEV)).sum() + sum((h * gg.to(DEV)).sum() for h, gg in zip(hl, ghl))).backward()
    close(xd.grad, x64.grad, rtol=1e-4, atol=5e-5, what="dx")
    for d in range(2):
        for name, a, r in zip(("dwih", "dwhh", "dbih", "dbhh", "dh0"), dl[d], leaves[d]):
            close(a.grad, r.grad, rtol=1e-4, atol=1e-4, what="%s[%d]" % (name, d))


@pytest.mark.parametrize("reverse", [False, True])
def test_persistent_gru_launch_cut_in_time(reverse, monkeypatch):
    """recurrences of different lengths in one gru_layer call: the launch is cut where the short ones end and the
    long ones continue in a launch of their own from the handed-over state (ops._segments); results and gradients
    equal the explicit equations, and equal the uncut launch"""
    from dss_vae_amd import ops
    import oracle
    B, shapes = 70, [(40, 30), (12, 50), (40, 16), (13, 30)]  # (T, H) per recurrence
    monkeypatch.setattr(ops, "GRU_SEGMENT_MIN", 8)
    # the T = 13 recurrence ends one step into the second segment: too short a piece for a cut of its own
    assert [(a, b) for a, b, _ in ops._segments([t for t, _ in shapes])] == [(0, 12), (12, 40)]
    ps = [[rnd(B, T, 3 * H, seed=i), rnd(3 * H, H, seed=10 + i, scale=0.3), rnd(3 * H, seed=20 + i, scale=0.3),
           rnd(B, H, seed=30 + i)] for i, (T, H) in enumerate(shapes)]
    gos = [rnd(B, T, H, seed=40 + i) for i, (T, H) in enumerate(shapes)]
    ghs = [rnd(B, H, seed=50 + i) for i, (T, H) in enumerate(shapes)]

    def run(seg_min):
        monkeypatch.setattr(ops, "GRU_SEGMENT_MIN", seg_min)
        dl = [[t.clone().to(DEV).requires_grad_(True) for t in p] for p in ps]
        insts = [dict(xproj=x, whh=whh, bhh=bhh, h0=h0, reverse=reverse, group=i, col=0)
                 for i, (x, whh, bhh, h0) in enumerate(dl)]
        outs, hl = ops.gru_layer(insts, check=True)
        sum((o * g.to(DEV)).sum() + (h * gh.to(DEV)).sum() for o, h, g, gh in zip(outs, hl, gos, ghs)).backward()
        return outs, hl, dl

    outs, hl, dl = run(8)
    outs_u, hl_u, dl_u = run(0)
    for i, (T, H) in enumerate(shapes):
        leaves = [t.clone().double().requires_grad_(True) for t in ps[i]]
        x, whh, bhh, h0 = leaves
        o, h = oracle.gru_loop(x, h0, torch.eye(3 * H, dtype=torch.float64), whh, torch.zeros(3 * H, dtype=torch.float64),
                               bhh, reverse=reverse)
        ((o * gos[i].double()).sum() + (h * ghs[i].double()).sum()).backward()
        close(outs[i], o, what="out%d" % i)
        close(hl[i], h, what="h_last%d" % i)
        for name, a, r, u in zip(("dxproj", "dwhh", "dbhh", "dh0"), dl[i], leaves, dl_u[i]):
            close(a.grad, r.grad, rtol=1e-4, atol=1e-4, what="%s[%d]" % (name, i))
            close(a.grad, u.grad, rtol=1e-5, atol=1e-5, what="%s[%d] cut vs uncut" % (name, i))
        close(outs[i], outs_u[i], rtol=1e-6, atol=1e-6, what="out%d cut vs uncut" % i)


def test_persistent_gru_schedule_defers_short_recurrences_and_mixes_cell_heights(monkeypatch):
    """BASELINE configs[1]'s decoder layer -- [H600 T31 | 2 x H300 T91 | 2 x H300 T31] at batch 256 -- in one gru_layer
    call: the schedule (ops._schedule) runs the H600 recurrence in 64-row cells beside the first 31 steps of the long
    pair in 128-row cells, and the short pair WHOLE in the second launch beside the long pair's tails; every output
    and gradient equals the plain time-cut plan's (one cell height per launch, every recurrence from step 0)"""
    from dss_vae_amd import ops
    B, shapes = 256, [(31, 600), (91, 300), (91, 300), (31, 300), (31, 300)]
    plan = ops._schedule([(B, H, T) for T, H in shapes])
    assert [sorted(j for j, _, _, _ in e) for e in plan] == [[0, 1, 2], [1, 2, 3, 4]], plan
    assert {j: r for j, _, _, r in plan[0]} == {0: 64, 1: 128, 2: 128}
    assert {j: (t0, t1, r) for j, t0, t1, r in plan[1]} == {1: (31, 91, 64), 2: (31, 91, 64), 3: (0, 31, 128),
                                                             4: (0, 31, 128)}
    ps = [[rnd(B, T, 3 * H, seed=i), rnd(3 * H, H, seed=10 + i, scale=1.0 / H ** 0.5), rnd(3 * H, seed=20 + i, scale=0.3),
           rnd(B, H, seed=30 + i)] for i, (T, H) in enumerate(shapes)]
    gos = [rnd(B, T, H, seed=40 + i) for i, (T, H) in enumerate(shapes)]
    ghs = [rnd(B, H, seed=50 + i) for i, (T, H) in enumerate(shapes)]

    def run(scheduled):
        monkeypatch.setattr(ops, "GRU_SCHEDULE", scheduled)
        dl = [[t.clone().to(DEV).requires_grad_(True) for t in p] for p in ps]
        insts = [dict(xproj=x, whh=whh, bhh=bhh, h0=h0, group=i, col=0) for i, (x, whh, bhh, h0) in enumerate(dl)]
        outs, hl = ops.gru_layer(insts, check=True)
        sum((o * g.to(DEV)).sum() + (h * gh.to(DEV)).sum() for o, h, g, gh in zip(outs, hl, gos, ghs)).backward()
        ops.check_persistent_errors()
        return outs, hl, dl

    outs, hl, dl = run(True)
    outs_p, hl_p, dl_p = run(False)
    for i in range(len(shapes)):
        close(outs[i], outs_p[i], rtol=2e-5, atol=2e-5, what="out%d" % i)
        close(hl[i], hl_p[i], rtol=2e-5, atol=2e-5, what="h_last%d" % i)
        for name, a, u in zip(("dxproj", "dwhh", "dbhh", "dh0"), dl[i], dl_p[i]):
            close(a.grad, u.grad, rtol=1e-4, atol=1e-4, what="%s[%d] scheduled vs time cuts only" % (name, i))


def test_persistent_gru_grid_one_over_the_residency_bound_is_refused():
    """every workgroup of a persistent launch spins on its peers, so the library admits a grid only up to what the
    chip holds at once (the occupancy answer x CUs: dssvae_gru_resident_workgroups): a grid of exactly that many
    workgroups runs, one more returns DSSVAE_ETOOBIG before anything is launched"""
    import ctypes as C
    from dss_vae_amd import lib as L
    lib = L.load()
    n = C.c_int(0)
    H = 600  # a 114 KB slice: one workgroup per CU, and no room for the K-split layouts' LDS beside it
    L.check(lib.dssvae_gru_resident_workgroups(H, 128, 0, C.byref(n)), "resident")
    n = n.value
    ns = (H + 15) // 16
    assert n >= ns
    full, rest = divmod(n, ns)

    def launch(extra):
        # `full` 128-row batch blocks of H = 600 (ns workgroups each) + one 128-row block of a narrower recurrence
        shapes = [(128 * full, H)]
        if rest + extra > 0:
            shapes.append((128, 16 * (rest + extra)))
        T = 2
        keep, insts = [], (L.GruFwdInst * len(shapes))()
        for i, (B, h) in enumerate(shapes):
            x = rnd(B, T, 3 * h, seed=i).to(DEV)
            whh, bhh = rnd(3 * h, h, seed=9 + i, scale=0.05).to(DEV), rnd(3 * h, seed=5).to(DEV)
            out = torch.empty(B, T, h, device=DEV)
            keep += [x, whh, bhh, out]
            it = insts[i]
            it.xproj, it.ldx, it.whh, it.bhh, it.out, it.ldo = x.data_ptr(), 3 * h, whh.data_ptr(), bhh.data_ptr(), \
                out.data_ptr(), h
            it.B, it.T, it.H, it.rows = B, T, h, 128
        Bs = (C.c_int32 * len(shapes))(*[b for b, _ in shapes])
        Hs = (C.c_int32 * len(shapes))(*[h for _, h in shapes])
        nbytes = lib.dssvae_gru_workspace_bytes(len(shapes), Bs, Hs)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
        rc = lib.dssvae_gru_fwd(insts, len(shapes), ws.data_ptr(), nbytes, L.stream())
        if rc == 0:
            L.check(lib.dssvae_gru_check(ws.data_ptr(), L.stream()), "gru_check")
            import oracle
            x, whh, bhh, out = keep[0].cpu(), keep[1].cpu(), keep[2].cpu(), keep[3]
            o, _ = oracle.gru_loop(x[:4], torch.zeros(4, H), torch.eye(3 * H), whh, torch.zeros(3 * H), bhh)
            close(out[:4], o, rtol=1e-4, atol=1e-4, what="rows of the full grid")
        return rc

    assert launch(0) == 0
    assert launch(1) == L.E_TOOBIG


def test_gru_cell_and_no_h0():
    from dss_vae_amd import ops
    B, H = 50, 600
    gi, gh, h = rnd(B, 3 * H, seed=1), rnd(B, 3 * H, seed=2), rnd(B, H, seed=3)
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    close(ops.gru_cell(gi.to(DEV), gh.to(DEV), h.to(DEV)), (1 - z) * n + z * h, what="cell")
    # h0 = None means zeros; inference mode saves no gates
    import oracle
    x, whh, bhh = rnd(5, 3, 3 * 20, seed=4), rnd(60, 20, seed=5, scale=0.3), rnd(60, seed=6)
    with torch.no_grad():
        (out,), (hl,) = ops.gru_layer([dict(xproj=x.to(DEV), whh=whh.to(DEV), bhh=bhh.to(DEV), h0=None, group=0,
                                            col=0)], save=False, check=True)
    eye = torch.eye(60)
    o, hT = oracle.gru_loop(x, torch.zeros(5, 20), eye, whh, torch.zeros(60), bhh)
    close(out, o, what="out no h0")
    close(hl, hT, what="hl no h0")


def test_oversized_gru_takes_the_stepwise_path():
    """H = 1024 does not fit the persistent kernel's LDS budget (3 * 64 slices * 1 KiB > 150 KiB)"""
    from dss_vae_amd import ops
    assert not ops._fits_persistent(8, 1024) and not ops._fits_persistent(2048, 512) and ops._fits_persistent(256, 600)
    B, T, H = 8, 3, 1024
    x, whh, bhh, h0 = rnd(B, T, 3 * H, seed=4), rnd(3 * H, H, seed=5, scale=0.03), rnd(3 * H, seed=6), rnd(B, H, seed=7)
    with torch.no_grad():
        (out,), (hl,) = ops.gru_layer([dict(xproj=x.to(DEV), whh=whh.to(DEV), bhh=bhh.to(DEV), h0=h0.to(DEV), group=0,
                                            col=0)], save=False)
    import oracle
    o, hT = oracle.gru_loop(x, h0, torch.eye(3 * H), whh, torch.zeros(3 * H), bhh)
    close(out, o, rtol=1e-4, atol=1e-4, what="out")
    close(hl, hT, rtol=1e-4, atol=1e-4, what="hl")


@pytest.mark.parametrize("B,E,H,V,layers,T", [
    (5, 12, 20, 97, 1, 6), (50, 300, 600, 3001, 2, 9),
    (7, 10, 18, 53, 2, 5),         # extents not multiples of 4: every step product through the tiled GEMM
    (50, 300, 1024, 1000, 2, 4),   # fused cell step with 16-row workgroups (the 32-row image is beyond LDS)
    (200, 64, 128, 500, 1, 4),     # decode batch beyond the skinny kernels' 128 rows
    (20, 64, 2048, 300, 1, 3),     # [x | h] image beyond LDS altogether
])
def test_native_greedy_decode_loop(B, E, H, V, layers, T):
    """dssvae_greedy_decode (whole loop, one call) vs an explicit fp64 loop of the reference's step"""
    from dss_vae_amd import ops
    table = rnd(V, E, seed=1, scale=0.5)
    ps = [(rnd(3 * H, E if l == 0 else H, seed=10 + l, scale=0.2), rnd(3 * H, H, seed=20 + l, scale=0.2),
           rnd(3 * H, seed=30 + l, scale=0.2), rnd(3 * H, seed=40 + l, scale=0.2)) for l in range(layers)]
    wo, bo, h0 = rnd(V, H + E, seed=2, scale=0.5), rnd(V, seed=3), rnd(layers, B, H, seed=4)
    tok = torch.full((B,), 1, dtype=torch.int64)
    h = [h0[l].double() for l in range(layers)]
    want = []
    for _ in range(T):
        e = table.double()[tok]
        x = e
        for l, (wih, whh, bih, bhh) in enumerate(ps):
            gi, gh = x @ wih.double().t() + bih.double(), h[l] @ whh.double().t() + bhh.double()
            r = torch.sigmoid(gi[:, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            h[l] = (1 - z) * n + z * h[l]
            x = h[l]
        tok = (torch.cat((x, e), 1) @ wo.double().t() + bo.double()).argmax(1)
        want.append(tok)
    got, hf = ops.greedy_decode(table.to(DEV), [tuple(t.to(DEV) for t in p) for p in ps], wo.to(DEV), bo.to(DEV),
                                h0.to(DEV), 1, T)
    assert torch.equal(got.cpu(), torch.stack(want, 0))
    close(hf, torch.stack(h, 0), rtol=1e-4, atol=1e-4, what="h_final")


@pytest.mark.parametrize("M,K1,K2,N", [(1, 4, 0, 5), (50, 600, 300, 3001), (33, 300, 0, 1800), (128, 20, 12, 97),
                                      (64, 600, 300, 30000), (100, 800, 300, 20000),   # 64-row arg-max (2 k passes; 1 and 2 row chunks)
                                      # ... with one or two rows behind the last full tile on the vector ALUs, or fewer tiles
                                      (50, 600, 300, 30000), (49, 600, 300, 30000), (33, 300, 0, 30000), (34, 600, 300, 30000),
                                      (40, 600, 300, 30000), (51, 300, 300, 30000)])
def test_skinny_linear_and_argmax(M, K1, K2, N):
    """C-ABI skinny kernels (decode-batch sized M) vs fp64 torch"""
    from dss_vae_amd import lib as L
    import ctypes as C
    lib = L.load()
    x1, x2 = rnd(M, K1, seed=1), (rnd(M, K2, seed=2) if K2 else None)
    w, b = rnd(N, K1 + K2, seed=3, scale=0.3), rnd(N, seed=4)
    xx = x1 if x2 is None else torch.cat((x1, x2), 1)
    want = xx.double() @ w.double().t() + b.double()
    d1, d2, dw, db = x1.to(DEV), None if x2 is None else x2.to(DEV), w.to(DEV), b.to(DEV)
    assert lib.dssvae_skinny_supported(M, K1, K2, K1, K2, K1 + K2) == 1
    assert lib.dssvae_skinny_supported(129, K1, K2, K1, K2, K1 + K2) == 0
    out = torch.full((M, N), float("nan"), device=DEV)
    L.check(lib.dssvae_skinny_linear(L.ptr(d1), K1, K1, L.ptr(d2), K2, K2, L.ptr(dw), K1 + K2, L.ptr(db), M, N,
                                     L.ptr(out), N, L.stream()), "skinny_linear")
    close(out, want, rtol=1e-4, atol=1e-4, what="skinny linear")
    S = lib.dssvae_skinny_argmax_splits(M, N)
    assert S >= 1
    pm, ps = torch.empty(M, S, device=DEV), torch.empty(M, S, device=DEV)
    pa = torch.empty(M, S, device=DEV, dtype=torch.int64)
    L.check(lib.dssvae_skinny_argmax(L.ptr(d1), K1, K1, L.ptr(d2), K2, K2, L.ptr(dw), K1 + K2, L.ptr(db), M, N,
                                     L.ptr(pm), L.ptr(ps), L.ptr(pa), L.stream()), "skinny_argmax")
    arg = torch.empty(M, device=DEV, dtype=torch.int64)
    L.check(lib.dssvae_lse_combine(L.ptr(pm), L.ptr(ps), L.ptr(pa), M, S, None, L.ptr(arg), L.stream()), "combine")
    assert torch.equal(arg.cpu(), want.argmax(1))
    close(pm.max(1).values, want.max(1).values, rtol=1e-4, atol=1e-4, what="max")
    # out of range shapes are refused, not mangled
    assert lib.dssvae_skinny_linear(L.ptr(d1), K1, K1, None, 0, 0, L.ptr(dw), K1 + K2, None, 129, N, L.ptr(out), N,
                                    L.stream()) == -2


@pytest.mark.parametrize("M,In,H", [(1, 4, 4), (50, 300, 512), (33, 12, 20), (128, 512, 100), (7, 100, 36),
                                    (50, 768, 768), (100, 300, 1000), (128, 64, 1024)])  # last: 32-row workgroups
def test_skinny_gru_step(M, In, H):
    """fused decode cell step (one launch) vs nn.GRUCell in fp64; refused shapes answer ETOOBIG"""
    from dss_vae_amd import lib as L
    lib = L.load()
    x, h = rnd(M, In, seed=1), rnd(M, H, seed=2)
    wih, whh = rnd(3 * H, In, seed=3, scale=0.3), rnd(3 * H, H, seed=4, scale=0.3)
    bih, bhh = rnd(3 * H, seed=5), rnd(3 * H, seed=6)
    cell = torch.nn.GRUCell(In, H).double()
    with torch.no_grad():
        cell.weight_ih.copy_(wih), cell.weight_hh.copy_(whh), cell.bias_ih.copy_(bih), cell.bias_hh.copy_(bhh)
        want = cell(x.double(), h.double())
    d = [t.to(DEV) for t in (x, h, wih, whh, bih, bhh)]
    out = torch.full((M, H), float("nan"), device=DEV)
    L.check(lib.dssvae_skinny_gru_step(L.ptr(d[0]), In, In, L.ptr(d[1]), H, H, L.ptr(d[2]), In, L.ptr(d[3]), H,
                                       L.ptr(d[4]), L.ptr(d[5]), M, L.ptr(out), H, L.stream()), "skinny_gru_step")
    close(out, want, rtol=1e-4, atol=1e-5, what="fused cell step")
    args = [L.ptr(d[0]), In, In, L.ptr(d[1]), H, H, L.ptr(d[2]), In, L.ptr(d[3]), H, L.ptr(d[4]), L.ptr(d[5])]
    assert lib.dssvae_skinny_gru_step(*args, 129, L.ptr(out), H, L.stream()) == -2
    assert lib.dssvae_skinny_gru_step(*args, M, L.ptr(d[1]), H, L.stream()) == -1  # h_new must not alias h_prev
    # an [x | h] image beyond the LDS budget even at 16 rows per workgroup (extents only: nothing is launched)
    assert lib.dssvae_skinny_gru_step(L.ptr(d[0]), 2048, 2048, L.ptr(d[1]), 2048, 2048, L.ptr(d[2]), 2048, L.ptr(d[3]),
                                      2048, L.ptr(d[4]), L.ptr(d[5]), M, L.ptr(out), 2048, L.stream()) == -2


@pytest.mark.parametrize("M,In,H,S", [(1, 4, 4, 1), (50, 300, 600, 235), (33, 12, 20, 70), (128, 64, 1024, 300), (50, 300, 304, 64)])
def test_skinny_gru_step_picks_its_tokens(M, In, H, S):
    """the greedy loop's first cell launch of a step: tokens from the previous step's arg-max partials (best split, ties
    to the smaller split), their embedding rows as x, tokens and rows left behind; vs torch + nn.GRUCell in fp64"""
    from dss_vae_amd import lib as L
    lib = L.load()
    V = 500
    g = torch.Generator().manual_seed(M + S)
    pm = torch.randn(M, S, generator=g)
    if S > 3:
        pm[:, S // 2] = pm.max(dim=1).values          # a tie per row: the smaller split must win
        pm[0, :] = -3.2e38                            # nothing above the floor: the first split's column
    pa = torch.randint(0, V, (M, S), generator=g)
    table, h = rnd(V, In, seed=1), rnd(M, H, seed=2)
    wih, whh = rnd(3 * H, In, seed=3, scale=0.3), rnd(3 * H, H, seed=4, scale=0.3)
    bih, bhh = rnd(3 * H, seed=5), rnd(3 * H, seed=6)
    best = torch.zeros(M, dtype=torch.long)
    for m in range(M):
        mx = pm[m].max()
        best[m] = int((pm[m] == mx).nonzero()[0]) if mx > -3.0e38 else 0
    tok = pa[torch.arange(M), best]
    cell = torch.nn.GRUCell(In, H).double()
    with torch.no_grad():
        cell.weight_ih.copy_(wih), cell.weight_hh.copy_(whh), cell.bias_ih.copy_(bih), cell.bias_hh.copy_(bhh)
        want = cell(table[tok].double(), h.double())
    d = [t.to(DEV) for t in (pm, pa, table, h, wih, whh, bih, bhh)]
    out = torch.full((M, H), float("nan"), device=DEV)
    tok_out = torch.full((M,), -1, dtype=torch.long, device=DEV)
    x_out = torch.full((M, In), float("nan"), device=DEV)
    L.check(lib.dssvae_skinny_gru_step_pick(L.ptr(d[0]), L.ptr(d[1]), S, L.ptr(tok_out), L.ptr(d[2]), In, In, L.ptr(x_out),
                                            In, L.ptr(d[3]), H, H, L.ptr(d[4]), In, L.ptr(d[5]), H, L.ptr(d[6]),
                                            L.ptr(d[7]), M, L.ptr(out), H, L.stream()), "skinny_gru_step_pick")
    assert torch.equal(tok_out.cpu(), tok)
    assert torch.equal(x_out.cpu(), table[tok])
    close(out, want, rtol=1e-4, atol=1e-5, what="picking cell step")


@pytest.mark.parametrize("rows,cols,nj,beta", [(300, 1000, 8, 0.0), (50, 260, 4, 0.0), (130, 128, 1, 1.0),
                                                (2050, 5004, 8, 1.0), (7, 4, 3, 0.0)])
def test_ce_wgrad_remainder_columns(rows, cols, nj, beta):
    """dssvae_ce_wgrad_cols: C[v, j] (+)= sum_m slab[m, v] * scale[v >> 7][m] * X[m, j] against fp64; X and C are
    column ranges of wider matrices, as in the fused CE chain (the last columns of [h | e | 1] / of dW)"""
    from dss_vae_amd import lib as L
    lib = L.load()
    ldl, ldx, ldc, tiles = cols + 4, nj + 12, nj + 20, (cols + 127) // 128
    slab, scale = rnd(rows, ldl, seed=1), rnd(tiles, rows + 3, seed=2).abs()
    X, C0 = rnd(rows, ldx, seed=3), rnd(cols, ldc, seed=4)
    sc_full = scale[:, :rows].t().repeat_interleave(128, dim=1)[:, :cols]          # [rows, cols]
    want = (slab[:, :cols].double() * sc_full.double()).t() @ X[:, 12:12 + nj].double()
    if beta:
        want = want + C0[:, 20:20 + nj].double()
    d_slab, d_scale, d_X, d_C = slab.to(DEV), scale.to(DEV), X.to(DEV), C0.to(DEV).clone()
    import ctypes as C
    L.check(lib.dssvae_ce_wgrad_cols(L.ptr(d_slab), ldl, rows, cols, L.ptr(d_scale), rows + 3,
                                     C.c_void_p(d_X.data_ptr() + 4 * 12), ldx, nj, C.c_void_p(d_C.data_ptr() + 4 * 20), ldc,
                                     beta, L.stream()), "dssvae_ce_wgrad_cols")
    close(d_C[:, 20:20 + nj], want, rtol=1e-4, atol=1e-4 * max(1.0, rows ** 0.5), what="remainder columns")
    assert torch.equal(d_C[:, :20].cpu(), C0[:, :20])    # nothing outside the nj columns is touched
    assert lib.dssvae_ce_wgrad_cols(L.ptr(d_slab), ldl, rows, cols, L.ptr(d_scale), rows + 3, L.ptr(d_X), ldx, 9,
                                    L.ptr(d_C), ldc, 0.0, L.stream()) == -2


def _lstm_cell64(x, h, c, wih, whh, bih, bhh):
    cell = torch.nn.LSTMCell(x.size(1), h.size(1)).double()
    with torch.no_grad():
        cell.weight_ih.copy_(wih), cell.weight_hh.copy_(whh), cell.bias_ih.copy_(bih), cell.bias_hh.copy_(bhh)
        return cell(x.double(), (h.double(), c.double()))


@pytest.mark.parametrize("M,In,H", [(1, 4, 4), (50, 300, 600), (33, 12, 20), (128, 512, 100), (7, 100, 36),
                                    (50, 768, 768), (100, 300, 1000), (128, 64, 1024), (50, 16, 8)])
def test_skinny_lstm_step(M, In, H):
    """fused decode cell step for nn.LSTM decoders (one launch; c updated in place) vs nn.LSTMCell in fp64"""
    from dss_vae_amd import lib as L
    lib = L.load()
    x, h, c = rnd(M, In, seed=1), rnd(M, H, seed=2), rnd(M, H, seed=7)
    wih, whh = rnd(4 * H, In, seed=3, scale=0.3), rnd(4 * H, H, seed=4, scale=0.3)
    bih, bhh = rnd(4 * H, seed=5), rnd(4 * H, seed=6)
    want_h, want_c = _lstm_cell64(x, h, c, wih, whh, bih, bhh)
    d = [t.to(DEV) for t in (x, h, wih, whh, bih, bhh)]
    cd = c.to(DEV)
    out = torch.full((M, H), float("nan"), device=DEV)
    L.check(lib.dssvae_skinny_lstm_step(L.ptr(d[0]), In, In, L.ptr(d[1]), H, H, L.ptr(cd), L.ptr(cd), H, L.ptr(d[2]), In,
                                        L.ptr(d[3]), H, L.ptr(d[4]), L.ptr(d[5]), M, L.ptr(out), H, L.stream()),
            "skinny_lstm_step")
    close(out, want_h, rtol=1e-4, atol=1e-5, what="fused LSTM step: h")
    close(cd, want_c, rtol=1e-4, atol=1e-5, what="fused LSTM step: c (in place)")
    args = [L.ptr(d[0]), In, In, L.ptr(d[1]), H, H, L.ptr(cd), L.ptr(cd), H, L.ptr(d[2]), In, L.ptr(d[3]), H, L.ptr(d[4]),
            L.ptr(d[5])]
    assert lib.dssvae_skinny_lstm_step(*args, 129, L.ptr(out), H, L.stream()) == -2
    assert lib.dssvae_skinny_lstm_step(*args, M, L.ptr(d[1]), H, L.stream()) == -1  # h_new must not alias h_prev
    args[6] = None
    assert lib.dssvae_skinny_lstm_step(*args, M, L.ptr(out), H, L.stream()) == -1   # no cell state


@pytest.mark.parametrize("M,In,H,S", [(1, 4, 4, 1), (50, 300, 600, 235), (33, 12, 20, 70), (128, 64, 1024, 300)])
def test_skinny_lstm_step_picks_its_tokens(M, In, H, S):
    """the LSTM loop's first cell launch of a step: tokens from the previous step's arg-max partials, their embedding
    rows as x, tokens and rows left behind; vs torch + nn.LSTMCell in fp64"""
    from dss_vae_amd import lib as L
    lib = L.load()
    V = 500
    g = torch.Generator().manual_seed(M + S)
    pm = torch.randn(M, S, generator=g)
    if S > 3:
        pm[:, S // 2] = pm.max(dim=1).values          # a tie per row: the smaller split must win
    pa = torch.randint(0, V, (M, S), generator=g)
    table, h, c = rnd(V, In, seed=1), rnd(M, H, seed=2), rnd(M, H, seed=7)
    wih, whh = rnd(4 * H, In, seed=3, scale=0.3), rnd(4 * H, H, seed=4, scale=0.3)
    bih, bhh = rnd(4 * H, seed=5), rnd(4 * H, seed=6)
    best = torch.tensor([int((pm[m] == pm[m].max()).nonzero()[0]) for m in range(M)])
    tok = pa[torch.arange(M), best]
    want_h, want_c = _lstm_cell64(table[tok], h, c, wih, whh, bih, bhh)
    d = [t.to(DEV) for t in (pm, pa, table, h, wih, whh, bih, bhh)]
    cd, cn = c.to(DEV), torch.full((M, H), float("nan"), device=DEV)
    out = torch.full((M, H), float("nan"), device=DEV)
    tok_out = torch.full((M,), -1, dtype=torch.long, device=DEV)
    x_out = torch.full((M, In), float("nan"), device=DEV)
    L.check(lib.dssvae_skinny_lstm_step_pick(L.ptr(d[0]), L.ptr(d[1]), S, L.ptr(tok_out), L.ptr(d[2]), In, In,
                                             L.ptr(x_out), In, L.ptr(d[3]), H, H, L.ptr(cd), L.ptr(cn), H, L.ptr(d[4]), In,
                                             L.ptr(d[5]), H, L.ptr(d[6]), L.ptr(d[7]), M, L.ptr(out), H, L.stream()),
            "skinny_lstm_step_pick")
    assert torch.equal(tok_out.cpu(), tok)
    assert torch.equal(x_out.cpu(), table[tok])
    close(out, want_h, rtol=1e-4, atol=1e-5, what="picking LSTM step: h")
    close(cn, want_c, rtol=1e-4, atol=1e-5, what="picking LSTM step: c")
    assert torch.equal(cd.cpu(), c)   # (c_new given its own buffer: c_prev untouched)


@pytest.mark.parametrize("B,E,H,V,layers,T", [
    (5, 12, 20, 97, 1, 6), (50, 300, 600, 3001, 2, 9),
    (7, 10, 18, 53, 2, 5),         # extents not multiples of 4: two products + dssvae_lstm_step_fwd per layer
    (50, 300, 1024, 1000, 2, 4),   # fused cell step with 16-row workgroups
    (200, 64, 128, 500, 1, 4),     # decode batch beyond the skinny kernels' 128 rows
])
def test_native_lstm_greedy_decode_loop(B, E, H, V, layers, T):
    """dssvae_greedy_decode_lstm (whole loop, one call) vs an explicit fp64 loop of the reference's step on nn.LSTM's
    cell, and vs the loop enqueued step by step from the host (RNNDecoder._greedy_lstm_stepwise's kernels)"""
    from dss_vae_amd import ops
    table = rnd(V, E, seed=1, scale=0.5)
    ps = [(rnd(4 * H, E if l == 0 else H, seed=10 + l, scale=0.2), rnd(4 * H, H, seed=20 + l, scale=0.2),
           rnd(4 * H, seed=30 + l, scale=0.2), rnd(4 * H, seed=40 + l, scale=0.2)) for l in range(layers)]
    wo, bo = rnd(V, H + E, seed=2, scale=0.5), rnd(V, seed=3)
    h0, c0 = rnd(layers, B, H, seed=4), rnd(layers, B, H, seed=5)
    tok = torch.full((B,), 1, dtype=torch.int64)
    h, c = [h0[l].double() for l in range(layers)], [c0[l].double() for l in range(layers)]
    want = []
    for _ in range(T):
        e = table.double()[tok]
        x = e
        for l, (wih, whh, bih, bhh) in enumerate(ps):
            gt = x @ wih.double().t() + bih.double() + h[l] @ whh.double().t() + bhh.double()
            i, f, gg, o = gt.split(H, 1)
            c[l] = torch.sigmoid(f) * c[l] + torch.sigmoid(i) * torch.tanh(gg)
            h[l] = torch.sigmoid(o) * torch.tanh(c[l])
            x = h[l]
        tok = (torch.cat((x, e), 1) @ wo.double().t() + bo.double()).argmax(1)
        want.append(tok)
    got, (hf, cf) = ops.greedy_decode(table.to(DEV), [tuple(t.to(DEV) for t in p) for p in ps], wo.to(DEV), bo.to(DEV),
                                      h0.to(DEV), 1, T, c0=c0.to(DEV))
    assert torch.equal(got.cpu(), torch.stack(want, 0))
    close(hf, torch.stack(h, 0), rtol=1e-4, atol=1e-4, what="h_final")
    close(cf, torch.stack(c, 0), rtol=1e-4, atol=1e-4, what="c_final")


def test_weighted_sums_loss_assembly():
    """out_j = sum_k coef[j][k] * sum(x_k) against fp64; gradients broadcast per input; an input whose live outputs
    all carry a zero coefficient gets None (not zeros); an output that received no gradient contributes nothing"""
    from dss_vae_amd import ops
    xs = [rnd(256, seed=1), rnd(1, seed=2).reshape(()), rnd(1000, seed=3), rnd(7, 3, seed=4), rnd(1, seed=5)]
    coef = [[1.0, 0.5, 0.0, -2.0, 0.0], [0.0, 0.0, 0.25, 0.0, 0.0], [-1.0 / 256, 3.0, 0.0, 0.0, 0.0]]
    xd = [x.to(DEV).requires_grad_(True) for x in xs]
    outs = ops.weighted_sums(coef, xd)
    sums = torch.stack([x.double().sum() for x in xs])
    ref = torch.tensor(coef, dtype=torch.float64) @ sums
    close(torch.stack(list(outs)), ref, what="weighted sums")
    assert all(o.dim() == 0 for o in outs)
    (outs[0] * 2.0 + outs[2]).backward()      # outs[1] (the only row that reads x_2) stays without a gradient
    g = [2.0 * coef[0][k] + coef[2][k] for k in range(5)]
    for k in (0, 1, 3):
        close(xd[k].grad, torch.full(xs[k].shape, g[k], dtype=torch.float64), what="grad %d" % k)
    assert xd[2].grad is None and xd[4].grad is None
    # the C-ABI's limits
    with pytest.raises(RuntimeError):
        ops.weighted_sums([[1.0] * 17], [x.to(DEV) for x in [rnd(3, seed=i) for i in range(17)]])


def test_sumsq_and_adam():
    from dss_vae_amd import ops
    n = 100003
    p, g = rnd(n, seed=1), rnd(n, seed=2, scale=3.0)
    ref_p = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=5e-4, betas=(0.9, 0.995))
    pd, m, v = p.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        gs = g * step
        ref_p.grad = gs.clone()
        gn = torch.nn.utils.clip_grad_norm_([ref_p], 5.0)
        opt.step()
        gd = gs.to(DEV)
        ss = torch.zeros(1, device=DEV)
        ops.sumsq(gd, ss)
        assert abs(ss.sqrt().item() - gn.item()) / gn.item() < 1e-5
        ops.adam_step(pd, gd, m, v, 5e-4, 0.9, 0.995, 1e-8, step, ss, 5.0)
    close(pd, ref_p, rtol=1e-5, atol=1e-6, what="adam")


def test_adam_update_is_dropped_when_the_error_word_is_set():
    """a step whose persistent recurrence timed out must not reach the parameters: the update kernel reads the
    sticky error word on device"""
    from dss_vae_amd import ops
    n = 4099
    p, g = rnd(n, seed=1).to(DEV), rnd(n, seed=2).to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    flag = torch.ones(1, device=DEV, dtype=torch.int32)
    before = p.clone()
    ops.adam_step(p, g, m, v, 5e-4, 0.9, 0.995, 1e-8, 1, None, 0.0, flag)
    assert torch.equal(p, before) and not m.any() and not v.any()
    flag.zero_()
    ops.adam_step(p, g, m, v, 5e-4, 0.9, 0.995, 1e-8, 1, None, 0.0, flag)
    assert not torch.equal(p, before) and m.any()


@pytest.mark.parametrize("uniform", [False, True])
def test_linear_ce_sentence_groups_and_fused_dropout(uniform):
    """group = T: the loss kernel sums the rows of a sentence itself -- same values and gradients as summing the
    per-row losses; ops.dropout equals x * dropout_mask for the same seed, forward and backward"""
    from dss_vae_amd import ops
    B, T, Kh, Ke, N = 6, 5, 24, 8, 52
    xh, xe = rnd(B * T, Kh, seed=1), rnd(B * T, Ke, seed=2)
    W, b = rnd(N, Kh + Ke, seed=3, scale=0.3), rnd(N, seed=4)
    tgt = torch.randint(0, N, (B * T,), generator=torch.Generator().manual_seed(5))
    res = []
    for group in (1, T):
        leaves = [t.clone().to(DEV).requires_grad_(True) for t in (xh, xe, W, b)]
        out = ops.linear_ce(leaves[0], leaves[1], leaves[2], leaves[3], tgt.to(DEV), 0, uniform_upstream=uniform,
                            group=group)
        sent = out.view(B, T).sum(1) if group == 1 else out
        assert sent.shape == (B,)
        (sent.sum() * 0.25).backward()
        res.append((sent.detach().cpu(), [t.grad.detach().cpu() for t in leaves]))
    close(res[1][0], res[0][0], rtol=1e-5, atol=1e-5, what="sentence sums")
    for a, c in zip(res[1][1], res[0][1]):
        close(a, c, rtol=1e-5, atol=1e-6, what="grad")
    x = rnd(7, 33, seed=8).to(DEV).requires_grad_(True)
    y = ops.dropout(x, 0.3, 1234)
    m = ops.dropout_mask((7, 33), 0.3, 1234, DEV)
    assert torch.equal(y, x.detach() * m)
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad, m)
