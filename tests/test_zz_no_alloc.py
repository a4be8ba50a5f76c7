"""dssvae_greedy_decode_noalloc (workspace in the library's own per-GPU device array, no allocation) against
dssvae_greedy_decode (caller-owned workspace), through the C-ABI: identical tokens and final states on two
calls in a row, the second with a smaller decode batch (the array keeps the first call's contents)."""
import ctypes as C
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(DEV)


def _decode_both(lib, L, B, E, H, V, layers, T, seed):
    table = _rnd(V, E, seed=seed + 1, scale=0.5)
    ps = [(_rnd(3 * H, E if l == 0 else H, seed=seed + 10 + l, scale=0.2), _rnd(3 * H, H, seed=seed + 20 + l, scale=0.2),
           _rnd(3 * H, seed=seed + 30 + l, scale=0.2), _rnd(3 * H, seed=seed + 40 + l, scale=0.2)) for l in range(layers)]
    wo, bo, h0 = _rnd(V, H + E, seed=seed + 2, scale=0.5), _rnd(V, seed=seed + 3), _rnd(layers, B, H, seed=seed + 4)
    prm = L.DecodeParams()
    prm.embedding, prm.ld_embedding, prm.n_embeddings = table.data_ptr(), E, V
    for l, p in enumerate(ps):
        prm.w_ih[l], prm.w_hh[l], prm.b_ih[l], prm.b_hh[l] = (t.data_ptr() for t in p)
    prm.w_out, prm.ld_out, prm.b_out = wo.data_ptr(), H + E, bo.data_ptr()
    prm.E, prm.H, prm.V, prm.n_layers = E, H, V, layers
    S = 4
    nbytes = lib.dssvae_greedy_decode_workspace_bytes(B, E, H, layers, S, V)
    assert 0 < nbytes <= L.DECODE_NOALLOC_BYTES
    ws = torch.empty(nbytes, device=DEV, dtype=torch.uint8)
    want_tok = torch.full((T, B), -1, device=DEV, dtype=torch.int64)
    want_h = torch.full((layers, B, H), float("nan"), device=DEV)
    L.check(lib.dssvae_greedy_decode(C.byref(prm), L.ptr(h0), B, T, 1, S, L.ptr(want_tok), L.ptr(want_h), L.ptr(ws),
                                     nbytes, L.stream()), "dssvae_greedy_decode")
    got_tok = torch.full((T, B), -1, device=DEV, dtype=torch.int64)
    got_h = torch.full((layers, B, H), float("nan"), device=DEV)
    L.check(lib.dssvae_greedy_decode_noalloc(C.byref(prm), L.ptr(h0), B, T, 1, S, L.ptr(got_tok), L.ptr(got_h),
                                             L.stream()), "dssvae_greedy_decode_noalloc")
    torch.cuda.synchronize()
    assert int(want_tok.min()) >= 0 and int(want_tok.max()) < V
    assert torch.equal(got_tok, want_tok)
    assert torch.equal(got_h, want_h)  # same kernels, same order: bit-identical
    return prm, h0, (table, ps, wo, bo)


def test_noalloc_greedy_decode_matches_workspace_entry_point_twice():
    from dss_vae_amd import lib as L
    lib = L.load()
    # first call: the default decoder shape at eval_bs 50; second: a smaller batch and other extents, so every
    # region of the array starts with stale contents of the first call
    _decode_both(lib, L, B=50, E=300, H=600, V=3001, layers=2, T=9, seed=100)
    _decode_both(lib, L, B=7, E=12, H=20, V=97, layers=1, T=6, seed=200)
    # a shape beyond the array is rejected before anything is launched
    prm, h0, keep = _decode_both(lib, L, B=5, E=12, H=20, V=97, layers=1, T=3, seed=300)
    big_B = 1 << 16
    assert lib.dssvae_greedy_decode_workspace_bytes(big_B, 12, 20, 1, 4, 97) > L.DECODE_NOALLOC_BYTES
    tok = torch.empty(3, big_B, device=DEV, dtype=torch.int64)
    rc = lib.dssvae_greedy_decode_noalloc(C.byref(prm), L.ptr(h0), big_B, 3, 1, 4, L.ptr(tok), None, L.stream())
    assert rc == L.E_TOOBIG
