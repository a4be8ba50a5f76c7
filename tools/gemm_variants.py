#!/usr/bin/env python
"""development aid: one GEMM shape timed under several builds of the library (DSSVAE_LIB), one subprocess each.
usage: gemm_variants.py kind M N K lib1.so lib2.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from dss_vae_amd import ops
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kbench import timeit
    kind, M, N, K = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    dev = "cuda"
    if os.environ.get("GEMM_DATA") == "zeros":     # operands that toggle no bits: is the rate limited by power?
        torch.randn = lambda *a, **k: torch.zeros(*a, **k)
    if os.environ.get("GEMM_DATA") == "ones":
        torch.randn = lambda *a, **k: torch.ones(*a, **k)
    if kind == "nt":
        a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev); fn = lambda: ops.mm_nt(a, b)
    elif kind == "nn":
        a, b = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev); fn = lambda: ops.mm_nn(a, b)
    else:
        a, b = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev); fn = lambda: ops.mm_tn(a, b)
    ms = timeit(fn, iters=20, warm=3)
    print("%-40s %s M=%d N=%d K=%d  %.3f ms  %.1f TFLOP/s" % (os.path.basename(os.environ.get("DSSVAE_LIB", "product")), kind, M, N, K, ms,
                                                       2.0 * M * N * K / ms / 1e9), flush=True)
else:
    shape = sys.argv[1:5]
    for lib in sys.argv[5:]:
        env = dict(os.environ)
        if lib != "product":
            env["DSSVAE_LIB"] = os.path.join(ROOT, lib)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + shape, env=env, check=False)
