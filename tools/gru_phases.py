#!/usr/bin/env python3
"""Where a step of the persistent GRU forward goes (development tool; needs the profiling build of the library:
    hipcc ... -DGRU_PROFILE -c gru.hip  ->  dss_vae_amd/lib/libdssvae_prof.so, run with DSSVAE_LIB pointing at it).
Prints, per sampled workgroup, the mean time between the lead wave's phase stamps over the steady-state steps:
    top -> counter matched -> own products done -> partial sums added -> gate math done -> state store drained
    -> signalled -> (history stores issued) next top.
usage: DSSVAE_LIB=.../libdssvae_prof.so python tools/gru_phases.py [--bwd] N_RECURRENCES B T H [B,T,H ...]
(further B,T,H triples add recurrences of other shapes to the same launch; --bwd stamps the backward kernel instead:
    top -> saved gates loaded, gate gradients done -> stores issued + published -> history stores issued
    -> counter matched -> own products done -> partial sums added, next top)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dss_vae_amd import lib as L, ops  # noqa: E402

NAMES = ["await", "load+mfma", "reduce", "gates", "store drain", "signal", "history"]
NAMES_BWD = ["saved gates + math", "stores + publish", "history", "await", "load+mfma", "reduce", "loop"]


def main():
    global NAMES
    argv = sys.argv[1:]
    bwd = "--bwd" in argv
    argv = [a for a in argv if a != "--bwd"]
    n, B, T, H = (int(a) for a in argv[:4])
    shapes = [(B, T, H)] * n + [tuple(int(v) for v in a.split(",")) for a in argv[4:]]
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    insts = []
    for i, (b, t, h) in enumerate(shapes):
        insts.append(dict(xproj=torch.randn(b, t, 3 * h, generator=g).to(dev).requires_grad_(bwd),
                          whh=(torch.randn(3 * h, h, generator=g) * 0.05).to(dev).requires_grad_(bwd),
                          bhh=torch.randn(3 * h, generator=g).to(dev).requires_grad_(bwd),
                          h0=torch.randn(b, h, generator=g).to(dev).requires_grad_(bwd), group=i, col=0))
    lib = L.load()
    lib.dssvae_gru_profile_read.argtypes = [C.c_void_p, C.c_int64]
    buf = np.zeros((4, 128, 8), dtype=np.uint64)
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    if bwd:
        NAMES = NAMES_BWD
        for rep in range(4):
            outs, _ = ops.gru_layer(insts, save=True)
            gos = [torch.ones_like(o) for o in outs]
            torch.cuda.synchronize()
            t0.record()
            torch.autograd.backward(outs, gos)
            t1.record()
            torch.cuda.synchronize()
    else:
        with torch.no_grad():
            for rep in range(3):
                ops.gru_layer(insts, save=True)
            torch.cuda.synchronize()
            t0.record()
            ops.gru_layer(insts, save=True)
            t1.record()
            torch.cuda.synchronize()
    print("%s launch(es) %.1f us = %.2f us/step of the longest" % ("backward" if bwd else "forward",
          t0.elapsed_time(t1) * 1e3, t0.elapsed_time(t1) * 1e3 / T))
    rc = lib.dssvae_gru_profile_read(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
    simd = np.zeros((4, 16), dtype=np.uint32)
    lib.dssvae_gru_profile_simd.argtypes = [C.c_void_p]
    assert lib.dssvae_gru_profile_simd(simd.ctypes.data) == 0
    print("SIMD of waves 0..15 in the sampled workgroups:", " | ".join("".join(str(int(v)) for v in r) for r in simd))
    part = np.zeros((4, 128, 4), dtype=np.uint64)
    lib.dssvae_gru_profile_part.argtypes = [C.c_void_p]
    assert lib.dssvae_gru_profile_part(part.ctypes.data) == 0
    steps = min(min(t for _, t, _ in shapes), 128)
    for w in range(4):
        s = buf[w, 4:steps - 1].astype(np.int64)
        nxt = buf[w, 5:steps, 0].astype(np.int64)
        if s[:, 0].min() == 0:
            continue
        d = np.concatenate([np.diff(s[:, :7], axis=1), (nxt - s[:, 6])[:, None]], axis=1) / 100.0  # 100 MHz -> us
        print("wg sample %d: step %.2f us | " % (w, (nxt - s[:, 0]).mean() / 100.0) +
              "  ".join("%s %.2f" % (nm, v) for nm, v in zip(NAMES, d.mean(0))))
        if w == 0 and os.environ.get("RAW"):
            for st in range(10, 16):
                row = buf[0, st].astype(np.int64)
                base = row[1]
                print("      step %d lead:" % st, " ".join("%+.2f" % ((v - base) / 100.0) for v in row[:8]), "| parts signalled:",
                      " ".join("%+.2f" % ((int(part[pw, st, 3]) - base) / 100.0) for pw in range(1, 4)))
        for pw in range(1, 4 if not bwd else 0):
            pp = part[pw, 4:steps - 1].astype(np.int64)
            if w == 0 and pp[:, 0].min() > 0:
                # non-lead waves of row group 0 of workgroup 0, relative to the lead's 'counter matched' stamp
                rel = (pp - s[:, 1:2]) / 100.0
                print("      part wave %d: top %+.2f  released %+.2f  products %+.2f  signalled %+.2f  (us after the "
                      "lead's counter match; lead: products %+.2f)" % (pw, *rel.mean(0), ((s[:, 2] - s[:, 1]) / 100.0).mean()))


if __name__ == "__main__":
    main()
