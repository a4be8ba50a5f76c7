"""rnn_cell='lstm' (networks/rnn_base.py:32-36) on the MI355X against golden vectors produced by the
reference's own RNNEncoder / RNNDecoder over torch.nn.LSTM (tests/golden/make_golden_lstm.py): packed bidirectional
2-layer encoder forward + backward, teacher-forced decoder score + backward (incl. the gradients of (h0, c0)), and
the greedy decode (identical tokens)."""
import pytest
import torch

from conftest import load_golden
from parity_util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_lstm_encoder_matches_the_reference():
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


def test_lstm_decoder_score_and_greedy_match_the_reference():
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
