#!/usr/bin/env python
"""development probe: the id cache's device-cut batches against DeviceBatch, and the captured training step
(train.GraphedTrainer) against the eager one -- losses over a few steps and host time per step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from dss_vae_amd import ops, synthetic
from dss_vae_amd.models import MODEL_CLS
from dss_vae_amd.structs.id_cache import IdCache
from dss_vae_amd.train import FlatAdam, GraphedTrainer, build_model_params_, train_iteration
from dss_vae_amd.vae.base_vae import DeviceBatch


def main():
    fixed = "--ragged" not in sys.argv
    model_name = "SyntaxVAE2"
    for a in sys.argv[1:]:
        if a.startswith("--model="):
            model_name = a.split("=")[1]
    V, Vt, B = 30000, 300, 256
    vocab = synthetic.vocab(V, Vt)
    exs = synthetic.examples(4 * B, V, Vt, fixed=fixed, seed=5)
    cache = IdCache.from_examples(exs, vocab).cuda()
    # 1. device-cut batch == host-tensorised batch
    idx = np.arange(B, dtype=np.int64)
    idx = idx[np.argsort(-cache.src_len[idx], kind="stable")]
    a = cache.batch(idx)
    b = DeviceBatch([exs[i] for i in idx], vocab)
    for name in ("src", "tgt_var", "syn_var"):
        ta, tb = getattr(a, name), getattr(b, name)
        assert torch.equal(ta, tb), name
        ka, kb = getattr(ta, "_keep_rows", None), getattr(tb, "_keep_rows", None)
        assert (ka is None) == (kb is None), name
        if ka is not None:
            assert torch.equal(ka, kb), name
    assert list(a.length) == list(b.length) and torch.equal(a.length.dev, b.length.dev)
    assert torch.equal(a.tree_length.dev, b.tree_length.dev)
    print("id cache batch == DeviceBatch, signature", a.signature(), flush=True)

    over = dict(dec_rd=0.0, dec_ed=0.0, rnn_drop=0.0, unk_rate=0.0, x0=6010, k=0.5) if "--det" in sys.argv else {}
    margs = synthetic.model_args(model_select=model_name, task_type=model_name, **over)

    def build():
        m = MODEL_CLS[model_name](args=margs, vocab=vocab)
        build_model_params_(m, seed=0)
        m.cuda().train()
        return m, FlatAdam(m, lr=0.0005, betas=(0.9, 0.995), clip_grad=5.0)

    batches = [cache.batch(np.sort(np.arange(k * B, (k + 1) * B))[np.argsort(
        -cache.src_len[k * B:(k + 1) * B], kind="stable")] + 0) for k in range(4)]
    if "--no-cache" in sys.argv:
        batches = [DeviceBatch(exs[k * B:(k + 1) * B], vocab) for k in range(4)]
    n_steps = 12
    keys = ("Loss", "KL Item", "NLL Loss", "ADV Loss", "MUL Loss")
    results = {}
    modes = [m for m in ("eager", "graph") if "--only=" + m in sys.argv] or ["eager", "graph"]
    for mode in modes:
        torch.manual_seed(11)
        from dss_vae_amd.networks.rnn_base import GLOBAL_SEEDS
        GLOBAL_SEEDS.state = 12345
        model, opt = build()
        runner = GraphedTrainer(model, opt, enabled=(mode == "graph"))
        it = 6000
        rows = []
        host = []
        for s in range(n_steps):
            it += 1
            bt = batches[s % 4] if fixed else batches[0]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ret = runner(bt, it)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
            rows.append([float(torch.mean(ret[k]).item()) for k in keys if k in ret] + [ret.get("KL Weight")])
        ops.check_persistent_errors()
        results[mode] = rows
        print(mode, "replays", runner.replays, "eager", runner.eager, "captures", runner.captures)
        for s, (h, r) in enumerate(zip(host, rows)):
            print("  step %2d host %7.2f ms total %7.2f ms  %s" % (s, h[0], h[1], " ".join("%.6f" % x for x in r)))
        print("  params checksum %.9e steps max %d" % (float(opt.flat.double().abs().sum().item()), opt.t), flush=True)
        del model, opt, runner
        torch.cuda.empty_cache()
    if "--det" in sys.argv and len(results) == 2:
        worst = 0.0
        for ra, rb in zip(results["eager"], results["graph"]):
            for x, y in zip(ra, rb):
                worst = max(worst, abs(x - y) / max(1e-6, abs(x)))
        print("worst relative difference eager vs graph: %.3e" % worst)


if __name__ == "__main__":
    main()
