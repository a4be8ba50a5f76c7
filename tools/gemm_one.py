#!/usr/bin/env python
"""one shape, a few launches: target for rocprofv3 --pmc passes.  usage: gemm_one.py kind M N K"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dss_vae_amd import ops
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = "cuda"
if kind == "nt":
    a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev); fn = lambda: ops.mm_nt(a, b)
elif kind == "nn":
    a, b = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev); fn = lambda: ops.mm_nn(a, b)
else:
    a, b = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev); fn = lambda: ops.mm_tn(a, b)
for _ in range(5):
    fn()
torch.cuda.synchronize()
