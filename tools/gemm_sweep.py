#!/usr/bin/env python
"""development aid: time the step's dominant GEMM shapes under the tile / split-K knobs of gemm.hip
(DSSVAE_GEMM_TILE, DSSVAE_GEMM_SLOTS).  usage: python tools/gemm_sweep.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dss_vae_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402

SHAPES = [("nn", 1024, 600, 30000), ("nn", 1024, 900, 30000), ("tn", 30000, 900, 1024), ("tn", 30000, 600, 1024),
          ("nt", 23296, 900, 300), ("tn", 900, 300, 23296), ("tn", 1800, 600, 7936), ("tn", 450, 300, 7680),
          ("nn", 7680, 300, 450), ("nn", 23296, 300, 900)]


def main():
    dev = "cuda"
    for kind, M, N, K in SHAPES:
        if kind == "nt":
            a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
            fn = lambda: ops.mm_nt(a, b)
        elif kind == "nn":
            a, b = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
            fn = lambda: ops.mm_nn(a, b)
        else:
            a, b = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
            fn = lambda: ops.mm_tn(a, b)
        res = []
        for tile in ("0", "64", "128"):
            for slots in ("2", "3", "4"):
                os.environ["DSSVAE_GEMM_TILE"], os.environ["DSSVAE_GEMM_SLOTS"] = tile, slots
                ms = timeit(fn, iters=8, warm=2)
                res.append((tile, slots, ms, 2.0 * M * N * K / ms / 1e9))
        print("%s M=%d N=%d K=%d: " % (kind, M, N, K) + "  ".join("t%s/s%s %.3fms %.0fTF" % r for r in res), flush=True)


if __name__ == "__main__":
    main()
