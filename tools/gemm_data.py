#!/usr/bin/env python
"""GEMM throughput vs operand data (power / clock sensitivity check)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dss_vae_amd import ops
from kbench import timeit
M = N = K = 4096
for name, mk in [("zeros", lambda *s: torch.zeros(*s, device="cuda")), ("uniform(-1,1)", lambda *s: torch.rand(*s, device="cuda") * 2 - 1),
                 ("randn", lambda *s: torch.randn(*s, device="cuda")), ("randn*0.08", lambda *s: torch.randn(*s, device="cuda") * 0.08)]:
    a, b = mk(M, K), mk(N, K)
    ms = timeit(lambda: ops.mm_nt(a, b), iters=20, warm=5)
    print("%-14s %.3f ms %.1f TFLOP/s" % (name, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
