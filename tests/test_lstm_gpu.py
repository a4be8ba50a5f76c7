"""rnn_cell='lstm' (networks/rnn_base.py:32-36) on the MI355X against golden vectors produced by the
reference's own RNNEncoder / RNNDecoder over torch.nn.LSTM (tests/golden/make_golden_lstm.py): packed bidirectional
2-layer encoder forward + backward, teacher-forced decoder score + backward (incl. the gradients of (h0, c0)), and
the greedy decode (identical tokens) -- through the persistent kernels (cell = LSTM) and through the stepwise path
that oversized recurrences take; the persistent kernels also against torch.nn.LSTM in fp64 at sizes that exercise
several batch blocks, the K split and packed lengths."""
import pytest
import torch

from conftest import load_golden
from parity_util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(params=["persistent", "stepwise"])
def lstm_path(request, monkeypatch):
    from dss_vae_amd import ops
    monkeypatch.setattr(ops, "FORCE_STEPWISE_GRU", request.param == "stepwise")
    return request.param


def test_lstm_encoder_matches_the_reference(lstm_path):
    from dss_vae_amd.encoder import RNNEncoder
    g = load_golden("lstm_encdec")
    enc = RNNEncoder(vocab_size=g["V"], max_len=g["T"], input_size=g["E"], hidden_size=g["H"], embed_droprate=0.0,
                     rnn_droprate=0.0, n_layers=g["L"], bidirectional=True, rnn_cell="lstm", variable_lengths=True)
    enc.load_state_dict(g["enc_state"], strict=True)
    enc.to(DEV).train()
    out, (hn, cn) = enc(g["enc_ids"].to(DEV), g["enc_lengths"])
    for name, got, want in (("out", out, g["enc_out"]), ("h_n", hn, g["enc_hn"]), ("c_n", cn, g["enc_cn"])):
        assert got.shape == want.shape and rel_err(got, want) < 1e-5, (name, rel_err(got, want))
    ((out * g["enc_w_out"].to(DEV)).sum() + (hn * g["enc_w_h"].to(DEV)).sum() + (cn * g["enc_w_c"].to(DEV)).sum()).backward()
    for k, p in enc.named_parameters():
        assert p.grad is not None and rel_err(p.grad, g["enc_grads"][k]) < 2e-4, (k, rel_err(p.grad, g["enc_grads"][k]))


def test_lstm_decoder_score_and_greedy_match_the_reference(lstm_path):
    from dss_vae_amd.decoder import RNNDecoder
    g = load_golden("lstm_encdec")
    D = 2 * g["H"]
    dec = RNNDecoder(vocab=g["V"], max_len=7, input_size=g["E"], hidden_size=D, eos_id=2, sos_id=1, pad_id=0,
                     n_layers=g["L"], rnn_cell="lstm", embed_droprate=0.0, rnn_droprate=0.0, use_attention=False)
    dec.load_state_dict(g["dec_state"], strict=True)
    dec.to(DEV).eval()
    h0 = g["dec_h0"].to(DEV).requires_grad_(True)
    c0 = g["dec_c0"].to(DEV).requires_grad_(True)
    score = dec.score(inputs=g["dec_tgt"].to(DEV), encoder_outputs=None, encoder_hidden=(h0, c0))
    assert rel_err(score, g["dec_score"]) < 1e-5, rel_err(score, g["dec_score"])
    (score * g["dec_w"].to(DEV)).sum().backward()
    for k, p in dec.named_parameters():
        assert p.grad is not None and rel_err(p.grad, g["dec_grads"][k]) < 2e-4, (k, rel_err(p.grad, g["dec_grads"][k]))
    assert rel_err(h0.grad, g["dec_dh0"]) < 2e-4 and rel_err(c0.grad, g["dec_dc0"]) < 2e-4
    with torch.no_grad():
        _, hidden, ret, _ = dec.forward(inputs=None, encoder_hidden=(h0.detach(), c0.detach()), encoder_outputs=None)
    tokens = torch.stack([t.view(-1) for t in ret["sequence"]], 0)
    assert torch.equal(tokens.cpu(), g["dec_greedy"])
    assert [int(x) for x in ret["length"]] == g["dec_lengths"]
    assert isinstance(hidden, tuple) and hidden[0].shape == (g["L"], g["B"], D)


def test_rnn_cell_selection():
    from dss_vae_amd.encoder import RNNEncoder
    with pytest.raises(NotImplementedError):
        RNNEncoder(vocab_size=10, max_len=5, input_size=4, hidden_size=4, rnn_cell="rnn")
    with pytest.raises(ValueError):
        RNNEncoder(vocab_size=10, max_len=5, input_size=4, hidden_size=4, rnn_cell="qrnn")


@pytest.mark.parametrize("B,T,H,packed,reverse", [(130, 12, 300, True, False), (70, 9, 150, False, True),
                                                  (300, 6, 96, True, True), (5, 7, 20, False, False)])
def test_persistent_lstm_against_torch(B, T, H, packed, reverse):
    """one LSTM recurrence over given input projections through the persistent kernels vs the cell equations of
    torch.nn.LSTM (gate order i, f, g, o) in fp64 on the host: outputs, final (h, c), and every gradient"""
    from dss_vae_amd import ops
    assert ops._lstm_fits_persistent(B, H)
    g = torch.Generator().manual_seed(B + H)
    x = torch.randn(B, T, 4 * H, generator=g)
    whh = torch.randn(4 * H, H, generator=g) * (0.5 / H ** 0.5)
    bhh, h0, c0 = torch.randn(4 * H, generator=g) * 0.3, torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)
    lengths = None
    if packed:
        lengths = sorted([int(v) for v in torch.randint(1, T + 1, (B,), generator=g)], reverse=True)
        lengths[0] = T
    w_o, w_h, w_c = torch.randn(B, T, H, generator=g), torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)

    leaves = [t.clone().double().requires_grad_(True) for t in (x, whh, bhh, h0, c0)]
    x64, W, b, h, c = leaves
    out = torch.zeros(B, T, H, dtype=torch.float64)
    lens = None if lengths is None else torch.tensor(lengths)
    for step in range(T):
        t = T - 1 - step if reverse else step
        gt = x64[:, t] + h @ W.t() + b
        i, f, gg, o = gt.split(H, 1)
        cn = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        hn = torch.sigmoid(o) * torch.tanh(cn)
        if lens is not None:
            live = (t < lens).unsqueeze(1)
            out[:, t] = torch.where(live, hn, torch.zeros_like(hn))
            h, c = torch.where(live, hn, h), torch.where(live, cn, c)
        else:
            out[:, t] = hn
            h, c = hn, cn
    ((out * w_o.double()).sum() + (h * w_h.double()).sum() + (c * w_c.double()).sum()).backward()

    dl = [t.clone().to(DEV).requires_grad_(True) for t in (x, whh, bhh, h0, c0)]
    len_d = None if lengths is None else torch.tensor(lengths, dtype=torch.int32, device=DEV)
    o_d, h_d, c_d = ops.lstm_layer(dl[0], dl[1], dl[2], h0=dl[3], c0=dl[4], lengths=len_d, reverse=reverse)
    ops.check_persistent_errors()
    for name, got, want in (("out", o_d, out), ("h_T", h_d, h), ("c_T", c_d, c)):
        assert rel_err(got, want.float()) < 2e-5, (name, rel_err(got, want.float()))
    ((o_d * w_o.to(DEV)).sum() + (h_d * w_h.to(DEV)).sum() + (c_d * w_c.to(DEV)).sum()).backward()
    for name, a, r in zip(("dxproj", "dwhh", "dbhh", "dh0", "dc0"), dl, leaves):
        assert rel_err(a.grad, r.grad.float()) < 1e-4, (name, rel_err(a.grad, r.grad.float()))
