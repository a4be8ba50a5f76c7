"""BASELINE.json configs[1] at FULL size on the MI355X (SyntaxVAE2, V=30000, V_t=300, B=256, src 30 /
tree 90, E=H=300, Z=100), where the CPU oracle is too slow to run the whole batch:
  * size-independent property: every loss term is a per-sentence sum divided by B, so
    B*term(batch) == B1*term(first part) + B2*term(second part), and the same holds for the gradient
    of mean(Loss) -- checked between ONE 256-sentence pass and its two 128-sentence halves
    (different row-chunk / tile / split-K schedules inside the kernels);
  * the oracle itself on a 16-sentence slice of the same batch with the full 30k vocabulary
    (losses 1e-4 relative, gradients, identical greedy tokens)."""
import pytest
import torch

from parity_util import build_pair, compare_loss_dicts, make_args, make_examples, product_examples, rel_err

pytestmark = pytest.mark.gpu

N_SRC, N_TGT, B = 30000, 300, 256
SUM_TERMS = ("KL Loss", "NLL Loss", "MUL Loss", "ADV Loss", "Loss", "SYN KL Loss", "SEM KL Loss")


def _noise(n, seed):
    gen = torch.Generator().manual_seed(seed)
    return {"eps_syn": torch.randn(n, 100, generator=gen), "eps_sem": torch.randn(n, 100, generator=gen)}


def _loss_and_grads(pm, exs, noise):
    for p in pm.parameters():
        p.requires_grad = True
        p.grad = None
    pm.noise = dict(noise)
    ret = pm.get_loss(product_examples(exs), 6000)
    torch.mean(ret["Loss"]).backward()
    grads = {n: p.grad.detach().clone() for n, p in pm.named_parameters() if p.grad is not None}
    return {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in ret.items()}, grads


def test_batch_additivity_at_full_size():
    args = make_args(dec_ed=0.0, dec_rd=0.0, tgt_wd=False)
    _, pm = build_pair("SyntaxVAE2", args, N_SRC, N_TGT, seed=2, train=True)
    exs = make_examples(B, N_SRC, N_TGT, src_len=30, tree_len=90, fixed=True, seed=1234)
    noise = _noise(B, 9)
    full, g_full = _loss_and_grads(pm, exs, noise)
    parts = []
    for lo, hi in ((0, B // 2), (B // 2, B)):
        sub = {k: v[lo:hi].contiguous() for k, v in noise.items()}
        parts.append(_loss_and_grads(pm, exs[lo:hi], sub))
    for k in SUM_TERMS:
        want = sum(float(torch.as_tensor(r[k]).double().mean()) * (B // 2) for r, _ in parts) / B
        got = float(torch.as_tensor(full[k]).double().mean())
        assert abs(got - want) <= 1e-4 * abs(want), (k, got, want)
    sq_full = sq_sum = 0.0
    assert set(g_full) == set(parts[0][1]) == set(parts[1][1])
    for n, g in g_full.items():
        s = (parts[0][1][n].double() * (B // 2) + parts[1][1][n].double() * (B // 2)) / B
        assert rel_err(g, s) < 2e-3, (n, rel_err(g, s))
        sq_full += float((g.double() ** 2).sum())
        sq_sum += float((s ** 2).sum())
    assert abs(sq_full ** 0.5 - sq_sum ** 0.5) / sq_sum ** 0.5 < 1e-4


def test_oracle_parity_on_a_slice_with_the_full_vocabulary():
    n = 16
    args = make_args(dec_ed=0.0, dec_rd=0.0, tgt_wd=False)
    om, pm = build_pair("SyntaxVAE2", args, N_SRC, N_TGT, seed=2, train=True)
    exs = make_examples(B, N_SRC, N_TGT, src_len=30, tree_len=90, fixed=False, seed=77)[::B // n][:n]
    noise = _noise(n, 10)
    om.noise, pm.noise = dict(noise), dict(noise)
    want = om.get_loss(exs, 6000)
    got = pm.get_loss(product_examples(exs), 6000)
    compare_loss_dicts(got, want, what="full-vocabulary slice")
    torch.mean(want["Loss"]).backward()
    torch.mean(got["Loss"]).backward()
    og = dict(om.named_parameters())
    sq_ref = sq_hip = 0.0
    for name, p in pm.named_parameters():
        if og[name].grad is None:
            assert p.grad is None, name
            continue
        assert rel_err(p.grad, og[name].grad) < 5e-3, (name, rel_err(p.grad, og[name].grad))
        sq_ref += float((og[name].grad.double() ** 2).sum())
        sq_hip += float((p.grad.double() ** 2).sum())
    assert abs(sq_hip ** 0.5 - sq_ref ** 0.5) / sq_ref ** 0.5 < 1e-4
    om.eval(), pm.eval()
    with torch.no_grad():
        t_ref = om.predict_tokens(exs)
        ret = pm.latent_for_init(pm.hidden_to_latent(pm.encode_to_hidden(product_examples(exs)), False))
        assert torch.equal(pm.decode_tokens(ret).cpu(), t_ref)
