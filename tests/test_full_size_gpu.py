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


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3], the stress configuration: hidden 1024 (-> per-direction encoder width 512, main decoder
# D = 2048, task decoders 1024 under mapper_type: link), latent 256, vocab 30k, batch 1024 per GPU.  The main and
# task decoders are beyond the persistent kernel's LDS budget and run the stepwise GRU (per-step MFMA GEMM + fused
# gate kernel, forward and backward); the encoders (512) stay persistent; the flat optimiser holds 381 M parameters.
# ---------------------------------------------------------------------------------------------------------------
def _stress_args(**over):
    from dss_vae_amd import synthetic
    kw = dict(synthetic.STRESS_OVERRIDES)
    kw.update(dec_ed=0.0, dec_rd=0.0, tgt_wd=False)
    kw.update(over)
    return make_args(**kw)


def _noise_z(n, z, seed):
    gen = torch.Generator().manual_seed(seed)
    return {"eps_syn": torch.randn(n, z, generator=gen), "eps_sem": torch.randn(n, z, generator=gen)}


def test_stress_config_batch_additivity():
    """one 1024-sentence pass of the stress model against its two 512-sentence halves: every loss term and the
    gradient of mean(Loss) add up (different row chunks, tile plans, split-K and stepwise-GRU GEMM shapes)"""
    from dss_vae_amd import ops
    from dss_vae_amd.models import MODEL_CLS
    from dss_vae_amd.train import build_model_params_
    from parity_util import product_vocab
    BS = 1024
    args = _stress_args()
    pm = MODEL_CLS["SyntaxVAE2"](args=args, vocab=product_vocab(N_SRC, N_TGT))
    build_model_params_(pm, seed=2)
    pm.cuda().train()
    assert sum(p.numel() for p in pm.parameters()) > 350e6
    assert not ops._fits_persistent(BS, 2048) and not ops._fits_persistent(BS, 1024) and ops._fits_persistent(BS, 512)
    exs = make_examples(BS, N_SRC, N_TGT, src_len=30, tree_len=90, fixed=True, seed=1234)
    noise = _noise_z(BS, 256, 9)
    full, g_full = _loss_and_grads(pm, exs, noise)
    g_full = {k: v.cpu() for k, v in g_full.items()}  # 1.5 GB per gradient set: keep one on the device at a time
    acc = None
    part_terms = []
    for lo, hi in ((0, BS // 2), (BS // 2, BS)):
        sub = {k: v[lo:hi].contiguous() for k, v in noise.items()}
        r, g = _loss_and_grads(pm, exs[lo:hi], sub)
        part_terms.append(r)
        if acc is None:
            acc = {k: v.double().cpu() * 0.5 for k, v in g.items()}
        else:
            assert set(acc) == set(g)
            for k, v in g.items():
                acc[k] += v.double().cpu() * 0.5
        del g
    ops.check_persistent_errors()
    for k in SUM_TERMS:
        want = sum(float(torch.as_tensor(r[k]).double().mean()) * (BS // 2) for r in part_terms) / BS
        got = float(torch.as_tensor(full[k]).double().mean())
        assert abs(got - want) <= 1e-4 * abs(want), (k, got, want)
    assert set(g_full) == set(acc)
    sq_full = sq_sum = 0.0
    for n, g in g_full.items():
        assert rel_err(g, acc[n]) < 2e-3, (n, rel_err(g, acc[n]))
        sq_full += float((g.double() ** 2).sum())
        sq_sum += float((acc[n] ** 2).sum())
    assert abs(sq_full ** 0.5 - sq_sum ** 0.5) / sq_sum ** 0.5 < 1e-4


def test_stress_config_oracle_parity_on_a_slice():
    """the CPU oracle at the stress dims on 8 ragged sentences with the full vocabulary: losses 1e-4, gradients,
    grad-norm 1e-4, identical greedy tokens -- the stepwise GRU backward and the D = 2048 decode inside a model"""
    n = 8
    args = _stress_args()
    om, pm = build_pair("SyntaxVAE2", args, N_SRC, N_TGT, seed=2, train=True)
    exs = make_examples(64, N_SRC, N_TGT, src_len=30, tree_len=90, fixed=False, seed=77)[::8][:n]
    noise = _noise_z(n, 256, 10)
    om.noise, pm.noise = dict(noise), dict(noise)
    want = om.get_loss(exs, 6000)
    got = pm.get_loss(product_examples(exs), 6000)
    compare_loss_dicts(got, want, what="stress slice")
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
