"""CPU oracle for the DSS-VAE hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32) restatement of the reference's
algorithm for the training / generation hot path (SURVEY.md section 8a, rows
A1-A15).  It exists so that the HIP path can be checked against it; it is never
the thing that is shipped or measured:

* only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
  ``cpu_baseline`` leg may import it;
* the product package ``dss_vae_amd`` never imports it and has no CPU
  fallback: it raises when the HIP library is missing.

Pinning: ``tests/golden/*.pt`` hold inputs, injected noise and outputs captured
from the *imported reference implementation* (``tests/golden/make_golden.py``,
run once in the build container with PYTHONPATH=/root/reference, torch
2.10.0 CPU).  ``tests/test_oracle_golden.py`` checks every function here
against those vectors, so parity of this oracle is PINNED (not "unpinned").
"""
from .dssvae_oracle import *  # noqa: F401,F403
