#!/usr/bin/env python3
"""LSTM recurrence, persistent kernels vs the stepwise path (development tool): fwd + bwd time of one
ops.lstm_layer call.  usage: python tools/lstm_bench.py [B T H]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dss_vae_amd import ops  # noqa: E402


def run(B, T, H, stepwise):
    ops.FORCE_STEPWISE_GRU = stepwise
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, 4 * H, generator=g).cuda().requires_grad_(True)
    whh = (torch.randn(4 * H, H, generator=g) * 0.05).cuda().requires_grad_(True)
    bhh = torch.randn(4 * H, generator=g).cuda().requires_grad_(True)
    h0, c0 = torch.randn(B, H, generator=g).cuda(), torch.randn(B, H, generator=g).cuda()
    go = torch.randn(B, T, H, generator=g).cuda()
    times = []
    for rep in range(6):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        out, h, c = ops.lstm_layer(x, whh, bhh, h0=h0, c0=c0)
        e[1].record()
        (out * go).sum().backward()
        e[2].record()
        torch.cuda.synchronize()
        times.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
    f, b = min(t[0] for t in times[2:]), min(t[1] for t in times[2:])
    print("B%d T%d H%d %s: forward %.3f ms (%.1f us/step)  backward %.3f ms (%.1f us/step incl. the weight-gradient GEMM)"
          % (B, T, H, "stepwise  " if stepwise else "persistent", f, f * 1e3 / T, b, b * 1e3 / T))


if __name__ == "__main__":
    B, T, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 30, 300)
    run(B, T, H, False)
    run(B, T, H, True)
