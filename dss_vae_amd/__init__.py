"""dss_vae_amd -- MI355X-native (gfx950) hot path for DSS-VAE training and generation.

Layout: ``csrc/`` hand-written HIP kernels + the C-ABI of ``include/dssvae_hip.h``;
``lib.py`` ctypes binding; ``ops.py`` autograd wrappers; the remaining sub-packages mirror the
reference's module interface (``models``, ``vae``, ``encoder``, ``decoder``, ``networks``,
``utils``, ``structs``) on top of those ops.  There is no CPU fallback.
"""
__version__ = "0.1.0"
