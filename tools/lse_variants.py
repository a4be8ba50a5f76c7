#!/usr/bin/env python
"""development aid: the LSE kernel (+ slab) on the headline shapes under the library named by DSSVAE_LIB"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from dss_vae_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402


def main():
    M, N = 7936, 30000
    for K in (600, 900):
        x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.05, torch.randn(N, device="cuda")
        slab = torch.empty(M, N, device="cuda")
        ms = timeit(lambda: ops._lse(x, K, K, None, 0, 0, w, K, b, M, N, logits=slab), iters=10, warm=3)
        print("%-24s LSE+slab M=%d N=%d K=%d  %.3f ms  %.1f TFLOP/s" % (
            os.path.basename(os.environ.get("DSSVAE_LIB", "product")), M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
