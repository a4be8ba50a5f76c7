#!/usr/bin/env python
"""development aid: the tree decoder's greedy decode (E 300, H 600, 2 layers, V 300, 100 steps): one decode of B rows,
one merged decode of 2B rows, and two decodes of B rows as parallel branches of one captured HIP graph"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dss_vae_amd import ops  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    E, H, V, layers = 300, 600, 300, 2
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.2).cuda()
    table = r(V, E)
    ps = [(r(3 * H, E if l == 0 else H), r(3 * H, H), r(3 * H), r(3 * H)) for l in range(layers)]
    wo, bo = r(V, H + E), r(V)
    B = 50
    ha, hb = r(layers, B, H), r(layers, B, H)
    hab = torch.cat((ha, hb), 1).contiguous()
    one = timed(lambda: ops.greedy_decode(table, ps, wo, bo, ha, 1, T))
    merged = timed(lambda: ops.greedy_decode(table, ps, wo, bo, hab, 1, T))
    print("one decode of %d rows: %.2f ms (%.1f us/step); merged %d rows: %.2f ms (%.1f us/step)"
          % (B, one * 1e3, one / T * 1e6, 2 * B, merged * 1e3, merged / T * 1e6), flush=True)
    # two branches of one graph
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    out = {}
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(graph, stream=cap):
            side.wait_stream(cap)
            with torch.cuda.stream(side):
                out["b"] = ops.greedy_decode(table, ps, wo, bo, hb, 1, T)[0]
            out["a"] = ops.greedy_decode(table, ps, wo, bo, ha, 1, T)[0]
            cap.wait_stream(side)
    both = timed(graph.replay)
    want_a = ops.greedy_decode(table, ps, wo, bo, ha, 1, T)[0]
    want_b = ops.greedy_decode(table, ps, wo, bo, hb, 1, T)[0]
    graph.replay()
    torch.cuda.synchronize()
    print("two decodes as branches of one graph: %.2f ms; tokens equal: %s %s"
          % (both * 1e3, torch.equal(out["a"], want_a), torch.equal(out["b"], want_b)), flush=True)
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g1, stream=cap):
            out["m"] = ops.greedy_decode(table, ps, wo, bo, hab, 1, T)[0]
    print("merged decode as a graph: %.2f ms" % (timed(g1.replay) * 1e3), flush=True)


if __name__ == "__main__":
    main()
