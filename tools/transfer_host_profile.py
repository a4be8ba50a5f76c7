#!/usr/bin/env python
"""development aid: host profile (cProfile) of a syntax-transfer pair batch (eval_adv, batch 50, default dims).
usage (GPU box): python tools/transfer_host_profile.py"""
import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dss_vae_amd import synthetic
from dss_vae_amd.models import MODEL_CLS
from dss_vae_amd.train import build_model_params_
margs = synthetic.model_args(model_select="SyntaxVAE2", task_type="SyntaxVAE2")
vocab = synthetic.vocab(30000, 300)
m = MODEL_CLS["SyntaxVAE2"](args=margs, vocab=vocab); build_model_params_(m, seed=0); m.cuda().eval()
exs = synthetic.examples(50, 30000, 300, src_len=30, tree_len=90, fixed=False, seed=99)
ref = synthetic.examples(50, 30000, 300, src_len=30, tree_len=90, fixed=False, seed=199)
with torch.no_grad():
    for _ in range(5):
        m.eval_adv(exs, ref)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m.eval_adv(exs, ref)
    torch.cuda.synchronize()
    print("eval_adv %.2f ms per batch of 50 pairs" % ((time.perf_counter() - t0) / 20 * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        m.eval_adv(exs, ref)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
