#!/usr/bin/env python
"""Host-contention rehearsal for N ranks on one GPU: the configs[1] training step with the whole
process -- every thread, the HIP runtime's included -- pinned to `--cores` host cores (2 = one rank's share of a
16-core host at N = 8), against the same step unpinned.  Prints one JSON line: ms/step and host enqueue ms/step for
the graph-replayed step (the product's default) and for eager launches, pinned and unpinned."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pin_process(cores):
    """every existing thread of this process onto `cores` (threads started later inherit it)"""
    for tid in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(tid), cores)
        except (OSError, ValueError):
            pass


def build(batch=256, vocab_size=30000, tree_vocab=300, seed=0):
    import torch
    from dss_vae_amd import synthetic
    from dss_vae_amd.models import MODEL_CLS
    from dss_vae_amd.structs.id_cache import IdCache
    from dss_vae_amd.train import FlatAdam, build_model_params_
    vocab = synthetic.vocab(vocab_size, tree_vocab)
    exs = synthetic.examples(4 * batch, vocab_size, tree_vocab, fixed=True, seed=5)
    cache = IdCache.from_examples(exs, vocab).cuda()
    model = MODEL_CLS["SyntaxVAE2"](args=synthetic.model_args(), vocab=vocab)
    build_model_params_(model, seed=seed)
    model.cuda().train()
    opt = FlatAdam(model, lr=0.0005, betas=(0.9, 0.995), clip_grad=5.0)
    return model, opt, cache


def measure(model, opt, cache, graph, steps=20, warmup=4, batch=256, it0=6000):
    """-> (ms per step, host enqueue ms per step): the real loop's work per step -- cut the batch from the id cache,
    run the step -- with one device sync around the timed region"""
    import numpy as np
    import torch
    from dss_vae_amd import ops
    from dss_vae_amd.train import GraphedTrainer
    runner = GraphedTrainer(model, opt, enabled=graph)
    idx = [np.arange(k * batch, (k + 1) * batch, dtype=np.int64) for k in range(4)]
    it = it0
    for s in range(warmup + (4 if graph else 0)):
        it += 1
        runner(cache.batch(idx[s % 4]), it)
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for s in range(steps):
        it += 1
        h0 = time.perf_counter()
        runner(cache.batch(idx[s % 4]), it)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.check_persistent_errors()
    if graph:
        assert runner.replays >= steps, (runner.replays, runner.eager)
    return 1e3 * dt / steps, 1e3 * host / steps


def rehearse(cores=2, steps=20, modes=("graph", "eager")):
    allowed = sorted(os.sched_getaffinity(0))
    model, opt, cache = build()
    out = {"cores_allowed": len(allowed), "cores_pinned": cores, "steps": steps}
    for mode in modes:
        free = measure(model, opt, cache, mode == "graph", steps)
        pin_process(set(allowed[:cores]))
        try:
            pinned = measure(model, opt, cache, mode == "graph", steps)
        finally:
            pin_process(set(allowed))
        out[mode] = {"unpinned_ms_per_step": free[0], "unpinned_host_enqueue_ms": free[1],
                     "pinned_ms_per_step": pinned[0], "pinned_host_enqueue_ms": pinned[1],
                     "pinned_over_unpinned": pinned[0] / free[0]}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cores", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    print(json.dumps(rehearse(a.cores, a.steps)))
