#!/usr/bin/env python3
"""The shader clock during the generation loop (development; needs -DSK_PROFILE build of skinny.hip as DSSVAE_LIB)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dss_vae_amd import lib as L, ops  # noqa: E402

g = torch.Generator().manual_seed(0)
B, E, H, V, T = 50, 300, 600, 30000, 35
table = torch.randn(V, E, generator=g).cuda()
layers = [tuple(t.cuda() for t in (torch.randn(3 * H, E if l == 0 else H, generator=g) * 0.05, torch.randn(3 * H, H, generator=g) * 0.05,
                                   torch.randn(3 * H, generator=g) * 0.1, torch.randn(3 * H, generator=g) * 0.1)) for l in range(2)]
wo, bo = (torch.randn(V, H + E, generator=g) * 0.05).cuda(), torch.randn(V, generator=g).cuda()
h0 = torch.randn(2, B, H, generator=g).cuda()
for rep in range(20):
    ops.greedy_decode(table, layers, wo, bo, h0, 1, T)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 2)()
lib = L.load()
assert lib.dssvae_skinny_profile_read(buf) == 0
print("arg-max kernel, workgroup 0: %d shader cycles in %.2f us -> %.2f GHz" % (buf[0], buf[1] / 100.0, buf[0] / (buf[1] * 10.0)))
S = lib.dssvae_skinny_argmax_splits(B, V)
pm, ps = torch.empty(B, S, device="cuda"), torch.empty(B, S, device="cuda")
pa = torch.empty(B, S, device="cuda", dtype=torch.int64)
xh, xe = torch.randn(B, H, generator=g).cuda(), torch.randn(B, E, generator=g).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for rep in range(3):
    ev[0].record()
    for _ in range(50):
        lib.dssvae_skinny_argmax(L.ptr(xh), H, H, L.ptr(xe), E, E, L.ptr(wo), H + E, L.ptr(bo), B, V, L.ptr(pm), L.ptr(ps),
                                 L.ptr(pa), L.stream())
    ev[1].record()
    torch.cuda.synchronize()
print("arg-max launch back to back: %.1f us" % (ev[0].elapsed_time(ev[1]) * 1e3 / 50))

# the decode cell kernel: phases of workgroup 0 / wave 0 and the launch back to back
# (layer 1 of the decoder: In = E, layer 2: In = H)
xin, hp = torch.randn(B, H, generator=g).cuda(), torch.randn(B, H, generator=g).cuda()
wih1 = (torch.randn(3 * H, E, generator=g) * 0.05).cuda()
hn = torch.empty(B, H, device="cuda")
for In, wih in ((E, wih1), (H, layers[1][0])):
    x = xin[:, :In].contiguous()
    for rep in range(3):
        ev[0].record()
        for _ in range(50):
            L.check(lib.dssvae_skinny_gru_step(L.ptr(x), In, In, L.ptr(hp), H, H, L.ptr(wih), In, L.ptr(layers[1][1]), H,
                                               L.ptr(layers[1][2]), L.ptr(layers[1][3]), B, L.ptr(hn), H, L.stream()), "gru_step")
        ev[1].record()
        torch.cuda.synchronize()
    st = (C.c_ulonglong * 8)()
    assert lib.dssvae_skinny_gru_profile_read(st) == 0
    d = [(st[i + 1] - st[i]) / 100.0 for i in range(5)]
    print("cell kernel In=%d H=%d: launch back to back %.2f us; workgroup 0: image + weight loads issued, image stored %.2f | barrier %.2f | "
          "products %.2f | bias loads + barrier %.2f | cell + store %.2f us" % (In, H, ev[0].elapsed_time(ev[1]) * 1e3 / 50, *d))
