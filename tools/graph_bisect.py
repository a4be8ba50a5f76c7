#!/usr/bin/env python
"""development probe: which op of the training step a HIP graph capture chokes on (stages from small to large;
the first that fails ends the process)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from dss_vae_amd import ops, synthetic
from dss_vae_amd.models import MODEL_CLS
from dss_vae_amd.train import FlatAdam, build_model_params_, train_iteration
from dss_vae_amd.vae.base_vae import DeviceBatch


def stage(name, fn, warm=True):
    print("stage", name, "...", flush=True)
    if warm:
        fn()
        torch.cuda.synchronize()
    ops._PERSISTENT_DONE.clear()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    ops._PERSISTENT_DONE.clear()
    print("   captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("   replayed", flush=True)
    return g, out


def main():
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    dev = torch.device("cuda")
    x = torch.randn(512, 300, device=dev)
    w = torch.randn(900, 300, device=dev, requires_grad=True)
    b = torch.zeros(900, device=dev, requires_grad=True)
    ids = torch.randint(4, 1000, (256, 32), device=dev)
    vocab = synthetic.vocab(30000, 300)
    stages = []
    stages.append(("torch", lambda: torch.randn(16, 16, device=dev) * 2))

    def s_dyn():
        ops.DYN = ops.StepScalars(dev)
        try:
            y = ops.dropout(x, 0.3, 5)
            r = ops.DYN.value("unk_rate", 0.25)
            z = ops.unk_replace(ids, r, vocab.src, seed=3)
            ops.DYN.upload()
        finally:
            ops.DYN = None
        return y, z
    stages.append(("dropout+unk dyn", s_dyn))
    stages.append(("gemm fwd", lambda: ops.mm_nt(x, w.detach(), b.detach())))

    def s_lin():
        y = ops.linear(x, w, b)
        y.sum().backward()
        return w.grad
    stages.append(("linear fwd+bwd (torch sum)", s_lin))

    def s_ws():
        parts = [torch.ones(256, device=dev, requires_grad=True) for _ in range(3)]
        outs = ops.weighted_sums([[1.0, 2.0, 3.0], [0.5, 0.0, 0.0]], parts)
        outs[0].backward()
        return outs
    stages.append(("weighted sums", s_ws))

    xp = torch.randn(256, 31, 900, device=dev)
    whh = (torch.randn(900, 300, device=dev) * 0.05).requires_grad_()
    bhh = torch.zeros(900, device=dev, requires_grad=True)

    def s_gru():
        (o,), (h,) = ops.gru_layer([dict(xproj=xp, whh=whh, bhh=bhh, h0=None, group=0, col=0)])
        return o
    stages.append(("gru fwd", s_gru))

    def s_gru_b():
        (o,), (h,) = ops.gru_layer([dict(xproj=xp, whh=whh, bhh=bhh, h0=None, group=0, col=0)])
        o.sum().backward()
        return whh.grad
    stages.append(("gru fwd+bwd", s_gru_b))

    W = (torch.randn(30000, 600, device=dev) * 0.05).requires_grad_()
    bo = torch.zeros(30000, device=dev, requires_grad=True)
    xh = torch.randn(1024, 600, device=dev, requires_grad=True)
    tg = torch.randint(0, 30000, (1024,), device=dev)

    def s_ce():
        nll = ops.linear_ce(xh, None, W, bo, tg, 0, uniform_upstream=True, group=1)
        nll.sum().backward()
        return nll
    stages.append(("linear_ce fwd+bwd", s_ce))

    margs = synthetic.model_args()
    model = MODEL_CLS["SyntaxVAE2"](args=margs, vocab=vocab)
    build_model_params_(model, seed=0)
    model.cuda().train()
    opt = FlatAdam(model, lr=0.0005, betas=(0.9, 0.995), clip_grad=5.0)
    batch = DeviceBatch(synthetic.examples(64, 30000, 300, seed=3), vocab)
    train_iteration(model, opt, batch, 6001)
    train_iteration(model, opt, batch, 6002)

    def s_fwd():
        with torch.no_grad():
            return model.get_loss(batch, 6003)["Loss"]
    stages.append(("model forward no_grad", s_fwd))

    def s_fb():
        opt.zero_grad("main")
        ret = model.get_loss(batch, 6003)
        torch.mean(ret["Loss"]).backward()
        return ret["Loss"]
    stages.append(("model forward+backward", s_fb))
    stages.append(("optimizer step", lambda: opt.step()))
    stages.append(("train_iteration", lambda: train_iteration(model, opt, batch, 6004)["Loss"]))
    def s_dyn_iter(bt):
        def run():
            ops.DYN = ops.StepScalars(dev)
            ops.DYN.adam = []
            steps = list(opt.steps)
            try:
                return train_iteration(model, opt, bt, 6005)["Loss"]
            finally:
                ops.DYN = None
                opt.steps = steps
        return run
    stages.append(("dyn train_iteration B64", s_dyn_iter(batch)))
    stages.append(("dyn train_iteration B64 clone", s_dyn_iter(batch.clone())))
    big = DeviceBatch(synthetic.examples(256, 30000, 300, seed=4), vocab)
    stages.append(("big train_iteration B256", lambda: train_iteration(model, opt, big, 6004)["Loss"]))
    stages.append(("dyn big train_iteration B256", s_dyn_iter(big)))
    stages.append(("full ret big train_iteration B256", lambda: train_iteration(model, opt, big, 6004)))
    if "--trainer" in sys.argv:
        from dss_vae_amd.train import GraphedTrainer
        variant = [a for a in sys.argv if a.startswith("--variant=")]
        variant = variant[0].split("=")[1] if variant else ""
        runner = GraphedTrainer(model, opt)
        bt = big if "big" in variant else batch
        if "manual" in variant:
            rec = dict(static=bt.clone() if "clone" in variant else bt, dyn=ops.StepScalars(dev))
            if "sync" in variant:
                ops.persistent_error_word(dev)
                torch.cuda.synchronize()
            ops._PERSISTENT_DONE.clear()
            steps = list(opt.steps)
            ops.DYN = rec["dyn"]
            g = torch.cuda.CUDAGraph()
            print("manual capture", variant, flush=True)
            with torch.cuda.graph(g, pool=None):
                rec["ret"] = train_iteration(model, opt, rec["static"], 6010, False, False)
            ops.DYN = None
            opt.steps = steps
            print("manual captured", flush=True)
            g.replay()
            torch.cuda.synchronize()
            print("manual replayed", flush=True)
        else:
            for it in range(6010, 6016):
                print("trainer step", it, flush=True)
                ret = runner(bt, it)
                torch.cuda.synchronize()
                print("   ok", float(torch.mean(ret["Loss"]).item()), runner.replays, runner.eager, flush=True)
        return
    keep = []
    for name, fn in stages:
        if only and not any(o in name for o in only):
            continue
        keep.append(stage(name, fn, warm="--cold" not in sys.argv))
    ops.check_persistent_errors()
    print("all stages passed", flush=True)


if __name__ == "__main__":
    main()
