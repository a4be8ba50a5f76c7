#!/usr/bin/env python
"""development aid: the grouped launches of the headline step, each timed over the k-tiles-per-workgroup knob
(DSSVAE_GROUP_KTPS; 0 = the library's own choice) and the tile-width knob (DSSVAE_GEMM_TILE)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dss_vae_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402

CASES = [
    ("tn", [(1800, 600, 7936), (900, 300, 23296)]),
    ("tn", [(1800, 300, 7936), (900, 300, 23296)]),
    ("tn", [(450, 300, 7680)] * 4),
    ("tn", [(450, 150, 7680)] * 4),
    ("nt", [(7936, 1800, 600), (23296, 900, 300), (23296, 900, 300), (7936, 900, 300), (7936, 900, 300)]),
    ("nt", [(7680, 450, 300)] * 4),
    ("nn", [(7936, 600, 1800), (23296, 300, 900), (23296, 300, 900), (7936, 300, 900), (7936, 300, 900)]),
    ("nn", [(7680, 300, 450)] * 4),
    ("nn", [(23296, 300, 900)]),
]


def operands(kind, M, N, K):
    r = lambda *s: torch.randn(*s, device="cuda")
    if kind == "nt":
        return r(M, K), r(N, K)
    if kind == "nn":
        return r(M, K), r(K, N)
    return r(K, M), r(K, N)


def main():
    ktps_list = [int(a) for a in sys.argv[1:]] or [0, 8, 12, 16, 24, 32, 48, 64, 96]
    for kind, shapes in CASES:
        pairs = [operands(kind, *s) for s in shapes]
        fl = sum(2.0 * m * n * k for m, n, k in shapes)
        row = []
        for tile in (0, 64):
            for sk in (1, 0):
                os.environ["DSSVAE_GROUP_TILE"] = str(tile)
                os.environ["DSSVAE_GROUP_SK"] = str(sk)
                for kt in (ktps_list if sk == 0 else [0]):
                    os.environ["DSSVAE_GROUP_KTPS"] = str(kt)
                    ms = timeit(lambda: ops._gemm_grouped(kind, pairs), iters=20, warm=3)
                    row.append("t%d sk%d k%d:%.3f(%.0f)" % (tile, sk, kt, ms, fl / ms / 1e9))
        os.environ["DSSVAE_GROUP_TILE"] = "0"
        os.environ["DSSVAE_GROUP_SK"] = "1"
        os.environ["DSSVAE_GROUP_KTPS"] = "0"
        print(kind, "x%d" % len(shapes), "%dx%dx%d" % shapes[-1], "  ms(TFLOP/s):", " ".join(row), flush=True)


if __name__ == "__main__":
    main()
