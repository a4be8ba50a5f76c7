"""Edge cases of the hot path on the MI355X against the CPU oracle: batch of one, one-token sentences,
ragged batches whose shortest sentence has length 1, vocabulary not a multiple of any tile."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle  # noqa: E402
from parity_util import build_pair, compare_loss_dicts, make_args, product_examples, rel_err  # noqa: E402

pytestmark = pytest.mark.gpu

TINY = dict(embed_size=20, hidden_size=24, enc_embed_dim=20, enc_hidden_dim=24, dec_embed_dim=20, dec_hidden_dim=24,
            latent_size=10, src_max_time_step=7, tgt_max_time_step=9, dec_ed=0.0, dec_rd=0.0, tgt_wd=False)


def _examples(lengths, n_src, n_tgt, seed=0):
    g = torch.Generator().manual_seed(seed)
    exs = []
    for L in lengths:
        src = ["w%d" % int(i) for i in torch.randint(0, n_src - 4, (L,), generator=g)]
        tgt = ["t%d" % int(i) for i in torch.randint(0, n_tgt - 4, (max(1, 2 * L),), generator=g)]
        exs.append(oracle.OExample(src, tgt))
    return exs


@pytest.mark.parametrize("model", ["SyntaxVAE2", "SyntaxVAE", "DisentangleVAE", "VanillaVAE"])
@pytest.mark.parametrize("lengths", [[1], [5], [6, 3, 1, 1], [2, 2, 2]])
def test_small_and_ragged_batches(model, lengths):
    n_src, n_tgt = 37, 11
    om, pm = build_pair(model, make_args(**TINY), n_src, n_tgt, seed=4, train=True)
    exs = _examples(lengths, n_src, n_tgt, seed=len(lengths))
    B = len(exs)
    g = torch.Generator().manual_seed(1)
    noise = {"eps_syn": torch.randn(B, 10, generator=g), "eps_sem": torch.randn(B, 10, generator=g),
             "eps": torch.randn(B, 10, generator=g)}
    om.noise, pm.noise = dict(noise), dict(noise)
    want = om.get_loss(exs, 6000)
    got = pm.get_loss(product_examples(exs), 6000)
    compare_loss_dicts(got, want, what="%s %s" % (model, lengths))
    torch.mean(want["Loss"]).backward()
    torch.mean(got["Loss"]).backward()
    og = dict(om.named_parameters())
    for n, p in pm.named_parameters():
        if og[n].grad is None:
            assert p.grad is None, n
        else:
            assert p.grad is not None and rel_err(p.grad, og[n].grad) < 5e-3, (n, rel_err(p.grad, og[n].grad))
    om.eval(), pm.eval()
    with torch.no_grad():
        t_ref = om.predict_tokens(exs)
        ret = pm.latent_for_init(pm.hidden_to_latent(pm.encode_to_hidden(product_examples(exs)), False))
        t_hip = pm.decode_tokens(ret).cpu()
    assert t_hip.shape == (7, B) and torch.equal(t_hip, t_ref)
    res = pm.predict(product_examples(exs))
    assert len(res) == B and all(len(r[0][0]) == r[1][0] <= 7 for r in res)


def test_generating_over_generates_to_a_multiple_of_the_batch():
    """BaseVAE.generating (base_vae.py:69-85): ceil(n / bs) batches of bs sentences; n < bs -> n"""
    _, pm = build_pair("SyntaxVAE2", make_args(**TINY), 37, 11, seed=4, train=False)
    with torch.no_grad():
        assert len(pm.generating(7, batch_size=3)) == 9
        assert len(pm.generating(2, batch_size=50)) == 2
        assert len(pm.generating(6, batch_size=3)) == 6
