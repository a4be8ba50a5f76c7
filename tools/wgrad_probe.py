#!/usr/bin/env python
"""development aid: the CE weight-gradient product (slab^T . [h | e | 1]) at N = 896 / 900 / 904 and the dgrad at
N = 256 / 300 / 320 / 600 / 640: what the remainder columns cost"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from dss_vae_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402


def main():
    rows, V = 7936, 30000
    slab = torch.rand(rows, V, device="cuda")
    tab = torch.rand((V + 127) // 128, rows, device="cuda")
    for N in (896, 900, 904):
        x = torch.randn(rows, N, device="cuda")
        out = torch.empty(V, N, device="cuda")
        ms = timeit(lambda: ops._gemm_dlogits(slab, V, rows, V, tab, 1, x, N, out, N, N, 0.0), iters=10, warm=3)
        print("wgrad N=%d  %.3f ms  %.1f TFLOP/s" % (N, ms, 2.0 * rows * V * N / ms / 1e9), flush=True)
    for N in (256, 300, 320, 600, 640):
        w = torch.randn(V, N, device="cuda") * 0.05
        out = torch.empty(rows, N, device="cuda")
        ms = timeit(lambda: ops._gemm_dlogits(slab, V, rows, V, tab, 0, w, N, out, N, N, 0.0), iters=10, warm=3)
        print("dgrad N=%d  %.3f ms  %.1f TFLOP/s" % (N, ms, 2.0 * rows * V * N / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
