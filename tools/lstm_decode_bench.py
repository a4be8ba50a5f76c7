#!/usr/bin/env python
"""Greedy generation with an nn.LSTM decoder: the native loop (dssvae_greedy_decode_lstm) against the loop enqueued
step by step from the host, and the GRU loop at the same shape for scale.  usage: lstm_decode_bench.py [B] [T]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from dss_vae_amd.decoder import RNNDecoder  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = "cuda:0"
torch.manual_seed(0)
for cell in ("lstm", "gru"):
    dec = RNNDecoder(vocab=30000, max_len=T, input_size=300, hidden_size=600, eos_id=2, sos_id=1, pad_id=0, n_layers=1,
                     rnn_cell=cell, embed_droprate=0.0, rnn_droprate=0.0, use_attention=False).to(dev).eval()
    h0 = torch.randn(1, B, 600, device=dev)
    hid = (h0, torch.randn(1, B, 600, device=dev)) if cell == "lstm" else h0
    paths = [("native", lambda: dec.greedy_tokens(hid, T))]
    if cell == "lstm":
        paths.append(("stepwise", lambda: dec._greedy_lstm_stepwise(hid, T)))
    toks = {}
    for name, fn in paths:
        with torch.no_grad():
            for _ in range(3):
                toks[name] = fn()[0]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        print("%-5s %-9s B=%d T=%d: %.3f ms per decode, %.1f us per step, %.0f sentences/s" %
              (cell, name, B, T, dt * 1e3, dt * 1e6 / T, B / dt))
    if len(toks) == 2:
        print("      identical tokens:", bool(torch.equal(toks["native"], toks["stepwise"])))
