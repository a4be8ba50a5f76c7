"""Beam search (decoder/beam_decoder.py:47-350) on the MI355X against golden vectors from the
reference's TopKDecoder over its own RNNDecoder (tests/golden/make_golden_beam.py): identical k-best token sequences,
lengths and back-pointers; scores, per-step log-probability rows and states within fp32 tolerance (1e-4 relative:
the fixture is torch CPU fp32, the device accumulates the projections in another order)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from parity_util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def _decoder(g):
    from dss_vae_amd.decoder import RNNDecoder
    dec = RNNDecoder(vocab=g["V"], max_len=g["T"], input_size=g["E"], hidden_size=g["H"], eos_id=2, sos_id=1, pad_id=0,
                     n_layers=g["L"], rnn_cell=g["cell"], embed_droprate=0.0, rnn_droprate=0.0, use_attention=False)
    dec.load_state_dict(g["state"], strict=True)
    return dec.to(DEV).eval()


def _hidden(g):
    h0 = g["h0"].to(DEV)
    return (h0, g["c0"].to(DEV)) if g["cell"] == "lstm" else h0


@pytest.mark.parametrize("i", range(4))
def test_beam_search_matches_the_reference(i):
    from dss_vae_amd.decoder import TopKDecoder
    g = load_golden("beam_search")["cases"][i]
    B, k, T, V, H, L = g["B"], g["k"], g["T"], g["V"], g["H"], g["L"]
    lstm = g["cell"] == "lstm"
    beam = TopKDecoder(_decoder(g), k)
    outs, dec_hidden, meta = beam.forward(inputs=None, encoder_hidden=_hidden(g), encoder_outputs=None, dmts=T)

    seq = torch.stack([s.squeeze(-1) for s in meta["topk_sequence"]], 0)
    assert seq.shape == (T, B, k) and torch.equal(seq.cpu(), g["topk_sequence"])
    assert meta["topk_length"] == g["topk_length"] and [int(x) for x in meta["length"]] == g["length"]
    assert torch.equal(torch.stack([s.squeeze(-1) for s in meta["sequence"]], 0).cpu(), g["sequence"])
    assert meta["score"].shape == (B, k) and rel_err(meta["score"], g["score"]) < TOL

    out = torch.stack(meta["output"], 0)
    assert out.shape == (T, B, k, V) and rel_err(out, g["output"]) < TOL
    assert rel_err(torch.stack(outs, 0), g["decoder_outputs"]) < TOL
    if lstm:
        h_t = torch.stack([s[0] for s in meta["h_t"]], 0)
        c_t = torch.stack([s[1] for s in meta["h_t"]], 0)
        assert h_t.shape == (T, L, B, k, H) and rel_err(h_t, g["h_t"]) < TOL and rel_err(c_t, g["c_t"]) < TOL
        assert rel_err(dec_hidden[0], g["h_n"]) < TOL and rel_err(dec_hidden[1], g["c_n"]) < TOL
    else:
        h_t = torch.stack(meta["h_t"], 0)
        assert h_t.shape == (T, L, B, k, H) and rel_err(h_t, g["h_t"]) < TOL
        assert dec_hidden.shape == (L, B, H) and rel_err(dec_hidden, g["h_n"]) < TOL


def test_beam_search_entry_point_and_one_beam_is_greedy():
    """beam_search() (the call models/seq2seq.py:212 makes) returns the metadata; a beam of one is the greedy decode
    until its first </s> (afterwards the single hypothesis is masked to -inf and the tie order decides)"""
    from dss_vae_amd.decoder import TopKDecoder
    g = load_golden("beam_search")["cases"][1]
    dec = _decoder(g)
    beam = TopKDecoder(dec, 3)
    meta = beam.beam_search(encoder_hidden=_hidden(g), encoder_outputs=None, decode_max_time_step=6, beam_size=1)
    assert beam.k == 1 and len(meta["topk_sequence"]) == 6
    seq = torch.stack([s.view(-1) for s in meta["topk_sequence"]], 0)  # [T, B]
    greedy, _ = dec.greedy_tokens(_hidden(g), max_length=6)
    for b in range(g["B"]):
        n = meta["topk_length"][b][0]
        assert torch.equal(seq[:n, b], greedy[:n, b])


def test_sentence_major_inflation_decodes_each_sentence_as_if_alone():
    """with `sentence_major` the beams of sentence b start from sentence b's state: a batch of three gives what three
    searches of one sentence give (with the reference's tiling only B = 1 has that property)"""
    from dss_vae_amd.decoder import TopKDecoder
    g = load_golden("beam_search")["cases"][1]
    dec = _decoder(g)
    beam = TopKDecoder(dec, g["k"])
    beam.sentence_major = True
    h0 = _hidden(g)
    _, _, meta = beam.forward(encoder_hidden=h0, dmts=g["T"])
    seq = torch.stack([s.squeeze(-1) for s in meta["topk_sequence"]], 0)
    for b in range(g["B"]):
        _, _, one = beam.forward(encoder_hidden=h0[:, b:b + 1].contiguous(), dmts=g["T"])
        assert torch.equal(torch.stack([s.squeeze(-1) for s in one["topk_sequence"]], 0)[:, 0], seq[:, b])
        assert one["topk_length"][0] == meta["topk_length"][b]
        assert rel_err(one["score"][0], meta["score"][b]) < 1e-5


def test_other_output_function_takes_the_general_path():
    from dss_vae_amd.decoder import TopKDecoder
    g = load_golden("beam_search")["cases"][0]
    beam = TopKDecoder(_decoder(g), g["k"])

    def fn(x, dim=-1):
        return F.log_softmax(x, dim=dim)

    _, _, a = beam.forward(encoder_hidden=_hidden(g), dmts=g["T"])
    _, _, b = beam.forward(encoder_hidden=_hidden(g), dmts=g["T"], function=fn)
    assert torch.equal(torch.cat(a["topk_sequence"], -1), torch.cat(b["topk_sequence"], -1))
    assert rel_err(a["score"], b["score"]) < 1e-5
