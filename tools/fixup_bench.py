#!/usr/bin/env python
"""development aid: the one-hot fix-up of the fused CE chain (dssvae_ce_onehot_fixup), its two corrections timed apart.
usage (GPU box): python tools/fixup_bench.py
(measured: dX 10 us; dW 30 us uniform targets, 50 us Zipf, 60 us for the 300-symbol tree vocabulary under 23 296 rows.  A dW
form without per-row atomics for small vocabularies -- a workgroup per (vocabulary row, 256 columns, 4096-row batch) scanning
the targets: 23 us alone -- left the step where it was, and was not kept.)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dss_vae_amd import lib as L
lib = L.load()
dev = "cuda:0"


def run(M, V, Kx, Kw, zipf):
    g = torch.Generator().manual_seed(0)
    if zipf:
        p = 1.0 / torch.arange(1, V + 1, dtype=torch.float64)
        tgt = torch.multinomial(p / p.sum(), M, replacement=True, generator=g)
    else:
        tgt = torch.randint(1, V, (M,), generator=g)
    tgt = tgt.to(dev)
    W, dX = torch.rand(V, Kx, device=dev), torch.rand(M, Kx, device=dev)
    Xc, dW = torch.rand(M, Kw, device=dev), torch.rand(V, Kw, device=dev)

    def t(dx, dw):
        def f():
            L.check(lib.dssvae_ce_onehot_fixup(L.ptr(tgt), 0, None, M, L.ptr(W) if dx else None, Kx, Kx,
                                               L.ptr(dX) if dx else None, Kx if dx else 0, L.ptr(Xc) if dw else None,
                                               Kw if dw else 0, Kw, L.ptr(dW) if dw else None, Kw, L.stream()), "fixup")
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    print("M=%d V=%d Kx=%d Kw=%d %s: dX only %.1f us, dW only %.1f us, both %.1f us" %
          (M, V, Kx, Kw, "zipf" if zipf else "uniform", t(True, False), t(False, True), t(True, True)))


run(7936, 30000, 600, 904, False)
run(7936, 30000, 600, 904, True)
run(23296, 300, 300, 604, False)

