#!/usr/bin/env python3
"""Golden vectors for beam search from the REFERENCE's TopKDecoder (decoder/beam_decoder.py:47-350)
over its own RNNDecoder, both imported.  Run in the build container:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_beam.py

As shipped the reference's beam search cannot run on this image; two things are supplied around it (neither changes
a line of its algorithm, and both are what its code evidently assumes):
  * `TopKDecoder.forward` calls `self.rnn.step(...)`, a method its RNNDecoder does not have: the decoder handed in here
    is a subclass whose `step` is `forward_step` (decoder/rnn_decoder.py:61-73), the only single-step method it has;
  * it computes beam back-pointers as `candidates / self.V` on int64 tensors (beam_decoder.py:165, 282), which was
    integer division in the torch it was written for and yields floats (then an index_select dtype error) since
    torch 1.5: for the duration of the call `Tensor.__truediv__` on an integer tensor by a Python int truncates.

Writes tests/golden/beam_search.pt: per case the decoder's state_dict, the initial state, k, the step count and the
reference's metadata (data only; torch 2.10 CPU).  A case is kept only if every score the search stored is finite
(all beams of a sentence ended = every candidate is -inf = torch.topk's tie order decides, which no fixture can pin)."""
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

from dss_vae.decoder.beam_decoder import TopKDecoder  # noqa: E402
from dss_vae.decoder.rnn_decoder import RNNDecoder  # noqa: E402


class _Decoder(RNNDecoder):
    def step(self, input_var, hidden, encoder_outputs, function):
        return self.forward_step(input_var, hidden, encoder_outputs, function)


class _IntegerDivision(object):
    def __enter__(self):
        self.orig = orig = torch.Tensor.__truediv__

        def div(a, b):
            if isinstance(b, int) and not a.is_floating_point():
                return torch.div(a, b, rounding_mode="trunc")
            return orig(a, b)

        torch.Tensor.__truediv__ = div

    def __exit__(self, *exc):
        torch.Tensor.__truediv__ = self.orig


def case(seed, V, E, H, L, B, k, T, cell, eos_bias):
    torch.manual_seed(seed)
    dec = _Decoder(vocab=V, max_len=T, input_size=E, hidden_size=H, eos_id=2, sos_id=1, pad_id=0, n_layers=L,
                   rnn_cell=cell, embed_droprate=0.0, rnn_droprate=0.0, use_attention=False)
    for p in dec.parameters():
        p.data.uniform_(-0.4, 0.4)
    dec.out.bias.data[2] += eos_bias  # beams do end inside the horizon
    dec.eval()
    h0 = torch.randn(L, B, H) * 0.7
    hidden = (h0, torch.randn(L, B, H) * 0.7) if cell == "lstm" else h0
    beam = TopKDecoder(dec, k)
    seen = {}
    inner = beam._backtrack

    def spy(nw_output, nw_hidden, predecessors, symbols, scores, b, hidden_size):
        seen["scores"] = torch.stack([s.view(-1) for s in scores], 0)
        seen["predecessors"] = torch.stack([s.view(-1) for s in predecessors], 0)
        seen["symbols"] = torch.stack([s.view(-1) for s in symbols], 0)
        return inner(nw_output, nw_hidden, predecessors, symbols, scores, b, hidden_size)

    beam._backtrack = spy
    with torch.no_grad(), _IntegerDivision():
        outs, dec_hidden, meta = beam.forward(inputs=None, encoder_hidden=hidden, encoder_outputs=None, dmts=T)
    if not torch.isfinite(seen["scores"]).all():
        return None
    lstm = cell == "lstm"
    g = {"V": V, "E": E, "H": H, "L": L, "B": B, "k": k, "T": T, "cell": cell,
         "state": {n: v.detach().clone() for n, v in dec.state_dict().items()},
         "h0": h0, "c0": hidden[1] if lstm else None,
         "step_scores": seen["scores"], "step_predecessors": seen["predecessors"], "step_symbols": seen["symbols"],
         "score": meta["score"].clone(), "topk_length": [[int(x) for x in row] for row in meta["topk_length"]],
         "topk_sequence": torch.stack([s.squeeze(-1) for s in meta["topk_sequence"]], 0),  # [T,B,k]
         "output": torch.stack(meta["output"], 0),  # [T,B,k,V]
         "length": [int(x) for x in meta["length"]],
         "sequence": torch.stack([s.squeeze(-1) for s in meta["sequence"]], 0),  # [T,k]: sentence 0's beams (beam_decoder.py:217)
         "decoder_outputs": torch.stack(outs, 0)}  # [T,B,V]
    if lstm:
        g["h_t"] = torch.stack([s[0] for s in meta["h_t"]], 0)  # [T,L,B,k,H]
        g["c_t"] = torch.stack([s[1] for s in meta["h_t"]], 0)
        g["h_n"], g["c_n"] = dec_hidden[0].clone(), dec_hidden[1].clone()  # [L,B,H]
    else:
        g["h_t"] = torch.stack(meta["h_t"], 0)
        g["h_n"] = dec_hidden.clone()
    n_eos = int((g["topk_sequence"] == 2).sum())
    print("case", cell, "L", L, "B", B, "k", k, "T", T, "eos symbols", n_eos, "lengths", g["topk_length"])
    return g


def main():
    specs = [  # B = 1 is the reference's only call site (models/seq2seq.py:203-216); the others pin the batch layout
        dict(seed=3, V=57, E=20, H=36, L=1, B=1, k=5, T=9, cell="gru", eos_bias=0.5),
        dict(seed=5, V=83, E=24, H=40, L=2, B=3, k=4, T=8, cell="gru", eos_bias=0.3),
        dict(seed=7, V=61, E=20, H=28, L=1, B=2, k=3, T=7, cell="lstm", eos_bias=0.9),
        dict(seed=9, V=45, E=16, H=32, L=2, B=1, k=6, T=10, cell="lstm", eos_bias=0.6),
    ]
    cases = []
    for s in specs:
        g = case(**s)
        if g is None:
            raise SystemExit("case %r stores a non-finite score: pick another seed / bias" % (s,))
        cases.append(g)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "beam_search.pt")
    torch.save({"torch": str(torch.__version__), "cases": cases}, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
