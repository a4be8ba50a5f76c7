#!/usr/bin/env python
"""Kernel micro-benchmarks on the MI355X (development aid): times single C-ABI launches at the
shapes of the DSS-VAE step with HIP events.  usage: python tools/kbench.py [gemm|ce|gru|all]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dss_vae_amd import ops  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def gemm_cases():
    dev = "cuda"
    cases = [
        # (kind, M, N, K, note)
        ("nt", 4096, 4096, 4096, "square"),
        ("nt", 7936, 1800, 300, "dec L0 xproj"),
        ("nt", 7936, 1800, 600, "dec L1 xproj"),
        ("nt", 23296, 900, 300, "tree xproj"),
        ("nt", 7680, 450, 300, "enc xproj"),
        ("nn", 7936, 900, 4096, "CE dX, column chunk"),
        ("nn", 1024, 900, 30000, "CE dX, row chunk"),
        ("nn", 256, 600, 30000, "BoW dX"),
        ("tn", 4096, 900, 7936, "CE dW, column chunk (dL^T X)"),
        ("tn", 30000, 900, 1024, "CE dW, row chunk"),
        ("tn", 1800, 600, 7936, "GRU dWhh"),
        ("tn", 1800, 300, 7936, "GRU dWih"),
        ("nn", 7936, 600, 1800, "GRU dX"),
    ]
    for kind, M, N, K, note in cases:
        if kind == "nt":
            a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
            fn = lambda: ops.mm_nt(a, b)
        elif kind == "nn":
            a, b = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
            fn = lambda: ops.mm_nn(a, b)
        else:
            a, b = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
            fn = lambda: ops.mm_tn(a, b)
        ms = timeit(fn)
        print("gemm %s M=%5d N=%5d K=%5d  %8.3f ms  %6.1f TFLOP/s  (%s)" % (kind, M, N, K, ms,
                                                                            2.0 * M * N * K / ms / 1e9, note), flush=True)


def ce_cases():
    dev = "cuda"
    for M, Kh, Ke, N, T, note in [(7936, 600, 300, 30000, 1, "main decoder"), (7936, 300, 300, 30000, 1, "sent adversary"),
                                  (23296, 300, 300, 300, 1, "tree head"), (256, 600, 0, 30000, 32, "BoW head")]:
        xh = (torch.randn(M, Kh, device=dev) * 0.3).requires_grad_()
        xe = (torch.randn(M, Ke, device=dev) * 0.3).requires_grad_() if Ke else None
        W = (torch.randn(N, Kh + Ke, device=dev) * 0.05).requires_grad_()
        b = torch.zeros(N, device=dev, requires_grad=True)
        tgt = torch.randint(1, N, (M, T), device=dev)

        def fwd():
            with torch.no_grad():
                return ops.linear_ce(xh, xe, W, b, tgt, 0)

        def fwdbwd():
            for t in (xh, xe, W, b):
                if t is not None:
                    t.grad = None
            ops.linear_ce(xh, xe, W, b, tgt, 0).sum().backward()

        f = timeit(fwd, iters=5)
        fb = timeit(fwdbwd, iters=5)
        fl = 2.0 * M * N * (Kh + Ke)
        print("ce M=%5d K=%d+%d N=%5d T=%2d  fwd %7.3f ms (%5.1f TF)  fwd+bwd %7.3f ms (%5.1f TF at 4x)  (%s)"
              % (M, Kh, Ke, N, T, f, fl / f / 1e9, fb, 4 * fl / fb / 1e9, note), flush=True)


def gru_cases():
    dev = "cuda"
    for B, T, H, n, note in [(256, 31, 600, 1, "main dec layer"), (256, 91, 300, 2, "tree heads layer"),
                             (256, 30, 150, 4, "encoder layer"), (256, 31, 300, 2, "sentence adversaries")]:
        insts = []
        for k in range(n):
            insts.append(dict(xproj=torch.randn(B, T, 3 * H, device=dev, requires_grad=True),
                              whh=(torch.randn(3 * H, H, device=dev) * 0.05).requires_grad_(),
                              bhh=torch.zeros(3 * H, device=dev, requires_grad=True),
                              h0=torch.randn(B, H, device=dev) * 0.1, group=k, col=0))

        def fwd():
            with torch.no_grad():
                ops.gru_layer(insts, save=False)

        def fwdbwd():
            outs, _ = ops.gru_layer(insts, save=True)
            sum(o.sum() for o in outs).backward()

        f = timeit(fwd, iters=5)
        fb = timeit(fwdbwd, iters=5)
        print("gru B=%d T=%d H=%d x%d  fwd %7.3f ms (%5.2f us/step)  fwd+bwd(+wgrad) %7.3f ms  (%s)"
              % (B, T, H, n, f, 1e3 * f / T, fb, note), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("gemm", "all"):
        gemm_cases()
    if what in ("ce", "all"):
        ce_cases()
    if what in ("gru", "all"):
        gru_cases()
