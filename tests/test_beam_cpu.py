"""Host half of beam search: `backtrack_plan` is a numpy restatement of the reference's
TopKDecoder._backtrack (decoder/beam_decoder.py:236-341) as an index plan.  Pinned here, without a GPU, on what the
reference's own search stored (back-pointers, symbols, step scores: tests/golden/make_golden_beam.py) against what its
_backtrack returned for them: scores, lengths, the k sequences, and the gathered per-step outputs."""
import numpy as np
import pytest
import torch

from conftest import load_golden

CASES = list(range(4))


@pytest.mark.parametrize("i", CASES)
def test_backtrack_plan_matches_the_reference(i):
    from dss_vae_amd.decoder.beam_decoder import backtrack_plan
    g = load_golden("beam_search")["cases"][i]
    B, k, T = g["B"], g["k"], g["T"]
    pred, sym, sc = (g["step_predecessors"].numpy(), g["step_symbols"].numpy(), g["step_scores"].numpy())
    plan = backtrack_plan(pred, sym, sc, B, k, eos=2)
    assert np.array_equal(plan["score"], g["score"].numpy())
    assert plan["length"] == g["topk_length"]
    seq = np.take_along_axis(sym, plan["src"], axis=1).reshape(T, B, k)
    assert np.array_equal(seq, g["topk_sequence"].numpy())
    # every place that holds an ended hypothesis points h_n at the step where it emitted </s>
    for col in range(B * k):
        t, row = int(plan["hn_t"][col]), int(plan["hn_row"][col])
        assert (t < 0) == (row < 0)
        if t >= 0:
            assert sym[t, row] == 2 and plan["length"][col // k][col % k] == t + 1


def test_inflate_is_the_reference_tiling():
    from dss_vae_amd.decoder.beam_decoder import _inflate
    a = torch.tensor([[1, 2], [3, 4]])
    assert _inflate(a, 2, 1).tolist() == [[1, 2, 1, 2], [3, 4, 3, 4]]  # the docstring examples of beam_decoder.py:22-39
    assert _inflate(a, 2, 0).tolist() == [[1, 2], [3, 4], [1, 2], [3, 4]]


def test_plan_without_any_ended_hypothesis_is_a_plain_backtrace():
    from dss_vae_amd.decoder.beam_decoder import backtrack_plan
    rng = np.random.RandomState(0)
    B, k, T = 2, 3, 5
    pos = np.repeat(np.arange(B) * k, k)
    pred = rng.randint(0, k, size=(T, B * k)) + pos
    sym = rng.randint(3, 20, size=(T, B * k))
    sc = rng.randn(T, B * k).astype(np.float32)
    plan = backtrack_plan(pred, sym, sc, B, k, eos=2)
    assert (plan["hn_t"] == -1).all() and plan["length"] == [[T] * k] * B
    for b in range(B):
        order = np.argsort(-sc[T - 1, b * k:(b + 1) * k], kind="stable")
        for rank, j in enumerate(order):
            row = b * k + j
            for t in range(T - 1, -1, -1):
                assert plan["src"][t, b * k + rank] == row
                row = pred[t, row]


class _TorchCells(object):
    """torch CPU stand-in for RNNDecoder.step_rows (the cell math of nn.GRU / nn.LSTM + the projection), so that the
    search's own bookkeeping -- candidate selection, state reordering, the single host hand-over, the gathers -- is
    checked end to end without a GPU.  The HIP step itself is checked in tests/test_beam_gpu.py."""

    def __init__(self, g):
        self.g, self.s = g, g["state"]
        self.hidden_size, self.vocab_size, self.sos_id, self.eos_id, self.max_length = g["H"], g["V"], 1, 2, g["T"]

    def init_state(self, h):
        return h

    def valid_args(self, inputs, encoder_hidden, encoder_outputs, function, teacher_forcing_ratio):
        h = encoder_hidden[0] if isinstance(encoder_hidden, tuple) else encoder_hidden
        return None, h.size(1), self.max_length

    def step_rows(self, tok, h, c=None, out=None, function=None):
        s, H = self.s, self.g["H"]
        emb = s["embedding.embeddings.weight"][tok]
        x, hn, cn = emb, [], []
        for l in range(self.g["L"]):
            gi = x @ s["rnn.weight_ih_l%d" % l].t() + s["rnn.bias_ih_l%d" % l]
            gh = h[l] @ s["rnn.weight_hh_l%d" % l].t() + s["rnn.bias_hh_l%d" % l]
            if c is not None:
                i, f, gg, o = (gi + gh).split(H, 1)
                cl = torch.sigmoid(f) * c[l] + torch.sigmoid(i) * torch.tanh(gg)
                x = torch.sigmoid(o) * torch.tanh(cl)
                cn.append(cl)
            else:
                r = torch.sigmoid(gi[:, :H] + gh[:, :H])
                z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
                n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
                x = (1 - z) * n + z * h[l]
            hn.append(x)
        logp = torch.log_softmax(torch.cat((x, emb), 1) @ s["out.weight"].t() + s["out.bias"], dim=-1)
        return out.copy_(logp), hn, (cn if c is not None else None)


@pytest.mark.parametrize("i", CASES)
def test_search_bookkeeping_end_to_end_on_torch_cells(i):
    from dss_vae_amd.decoder.beam_decoder import TopKDecoder
    g = load_golden("beam_search")["cases"][i]
    B, k, T = g["B"], g["k"], g["T"]
    lstm = g["cell"] == "lstm"
    beam = TopKDecoder(_TorchCells(g), k)
    hidden = (g["h0"], g["c0"]) if lstm else g["h0"]
    outs, dec_hidden, meta = beam.forward(encoder_hidden=hidden, dmts=T)
    assert torch.equal(torch.stack([s.squeeze(-1) for s in meta["topk_sequence"]], 0), g["topk_sequence"])
    assert meta["topk_length"] == g["topk_length"] and meta["length"] == g["length"]
    assert torch.equal(torch.stack([s.squeeze(-1) for s in meta["sequence"]], 0), g["sequence"])
    close = lambda a, b: torch.allclose(a, b, rtol=1e-4, atol=1e-5)
    assert close(meta["score"], g["score"]) and close(torch.stack(meta["output"], 0), g["output"])
    assert close(torch.stack(outs, 0), g["decoder_outputs"])
    if lstm:
        assert close(torch.stack([s[0] for s in meta["h_t"]], 0), g["h_t"])
        assert close(torch.stack([s[1] for s in meta["h_t"]], 0), g["c_t"])
        assert close(dec_hidden[0], g["h_n"]) and close(dec_hidden[1], g["c_n"])
    else:
        assert close(torch.stack(meta["h_t"], 0), g["h_t"]) and close(dec_hidden, g["h_n"])
