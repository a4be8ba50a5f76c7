#!/usr/bin/env python3
"""Golden vectors for rnn_cell='lstm'  from the REFERENCE's own RNNEncoder / RNNDecoder
(imported, unmodified; they wrap torch.nn.LSTM -- networks/rnn_base.py:32-36).  Run in the build container:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_lstm.py

Writes tests/golden/lstm_encdec.pt: state_dicts, inputs and the reference's outputs / gradients / greedy
tokens (data only; torch 2.10 CPU)."""
import os
import sys
import types

import torch

REF = os.environ.get("DSSVAE_REFERENCE", "/root/reference")
if REF not in sys.path:
    sys.path.insert(0, REF)
for name in ("nltk", "nltk.tree"):
    m = types.ModuleType(name)
    m.Tree = object
    sys.modules.setdefault(name, m)

from dss_vae.decoder.rnn_decoder import RNNDecoder  # noqa: E402
from dss_vae.encoder.rnn_encoder import RNNEncoder  # noqa: E402


def main():
    torch.manual_seed(11)
    V, E, H, L, B, T = 57, 20, 18, 2, 5, 9          # extents that are not tile multiples
    enc = RNNEncoder(vocab_size=V, max_len=T, input_size=E, hidden_size=H, embed_droprate=0.0, rnn_droprate=0.0,
                     n_layers=L, bidirectional=True, rnn_cell="lstm", variable_lengths=True)
    for p in enc.parameters():
        p.data.uniform_(-0.3, 0.3)
    lengths = [9, 7, 7, 4, 2]
    ids = torch.randint(4, V, (B, T))
    for b, n in enumerate(lengths):
        ids[b, n:] = 0
    out, (hn, cn) = enc(ids, lengths)
    w_out, w_h, w_c = torch.randn_like(out), torch.randn_like(hn), torch.randn_like(cn)
    ((out * w_out).sum() + (hn * w_h).sum() + (cn * w_c).sum()).backward()
    g = {"V": V, "E": E, "H": H, "L": L, "B": B, "T": T, "torch": str(torch.__version__),
         "enc_state": {k: v.detach().clone() for k, v in enc.state_dict().items()},
         "enc_ids": ids, "enc_lengths": lengths, "enc_out": out.detach(), "enc_hn": hn.detach(), "enc_cn": cn.detach(),
         "enc_w_out": w_out, "enc_w_h": w_h, "enc_w_c": w_c,
         "enc_grads": {k: p.grad.detach().clone() for k, p in enc.named_parameters()}}

    D = 2 * H
    dec = RNNDecoder(vocab=V, max_len=7, input_size=E, hidden_size=D, eos_id=2, sos_id=1, pad_id=0, n_layers=L,
                     rnn_cell="lstm", embed_droprate=0.0, rnn_droprate=0.0, use_attention=False)
    for p in dec.parameters():
        p.data.uniform_(-0.3, 0.3)
    dec.eval()  # the embedding block's dropout is 0 anyway
    tgt = torch.randint(4, V, (B, 8))
    tgt[:, 0] = 1
    for b, n in enumerate([8, 7, 6, 6, 4]):
        tgt[b, n - 1] = 2
        tgt[b, n:] = 0
    h0 = (torch.randn(L, B, D) * 0.5).requires_grad_(True)
    c0 = (torch.randn(L, B, D) * 0.5).requires_grad_(True)
    score = dec.score(inputs=tgt, encoder_outputs=None, encoder_hidden=(h0, c0))
    w_s = torch.randn(B)
    (score * w_s).sum().backward()
    with torch.no_grad():
        _, _, ret, _ = dec.forward(inputs=None, encoder_hidden=(h0.detach(), c0.detach()), encoder_outputs=None)
        tokens = torch.stack([t.view(-1) for t in ret["sequence"]], 0)
    g.update({"dec_state": {k: v.detach().clone() for k, v in dec.state_dict().items()}, "dec_tgt": tgt,
              "dec_h0": h0.detach().clone(), "dec_c0": c0.detach().clone(), "dec_score": score.detach(), "dec_w": w_s,
              "dec_grads": {k: p.grad.detach().clone() for k, p in dec.named_parameters()},
              "dec_dh0": h0.grad.clone(), "dec_dc0": c0.grad.clone(), "dec_greedy": tokens,
              "dec_lengths": [int(x) for x in ret["length"]]})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lstm_encdec.pt")
    torch.save(g, path)
    print("wrote", path, "score", score.tolist(), "greedy", tuple(tokens.shape))


if __name__ == "__main__":
    main()
