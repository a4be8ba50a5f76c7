"""End-to-end parity of the HIP product path (through the C-ABI) on the MI355X:
  * against the golden vectors captured from the reference itself (tests/golden/*.pt);
  * against the CPU oracle on seeded synthetic batches at realistic dims.
Bar (north_star): per-term losses within 1e-4 relative in fp32; identical greedy decodes."""
import copy

import os

import pytest
import torch

import oracle
from conftest import CASES, golden_args, load_golden
from parity_util import (build_pair, compare_loss_dicts, grad_excess, make_args, make_examples, product_examples, rel_err,
                         words)

pytestmark = pytest.mark.gpu
# absolute part of the element-wise gradient bar, as a fraction of the tensor's largest element: fp32 sums of a few
# hundred to a few thousand terms taken in another order than the reference's differ by about 1e-7 x the sum of the
# terms' magnitudes, which an element that is itself a near-cancellation does not shrink with
GRAD_ATOL = float(os.environ.get("DSSVAE_TEST_GRAD_ATOL", "1e-6"))


def examples(g, key="examples"):
    return [oracle.OExample(s, t) for s, t in zip(g[key]["src"], g[key]["tgt"])]


def set_noise(m, g, randn, rand):
    m.noise = {}
    if g["model"] == "VanillaVAE":
        if randn:
            m.noise["eps"] = randn[0]
    elif randn:
        m.noise["eps_syn"], m.noise["eps_sem"] = randn[0], randn[1]
    if rand:
        m.noise["rand_tgt"] = rand[0]


def pair(g, train):
    return build_pair(g["model"], golden_args(g), g["n_src"], g["n_tgt"], state_dict=g["state_dict"], train=train)


def hip_tokens(pm, ret):
    return pm.decode_tokens(ret).cpu()


@pytest.mark.parametrize("case", CASES)
def test_golden_eval_losses_and_greedy_decodes(case):
    g = load_golden(case)
    _, pm = pair(g, train=False)
    exs = product_examples(examples(g))
    compare_loss_dicts(pm.get_loss(exs, g["train_iter"]), g["eval_loss"], what="eval")
    compare_loss_dicts(pm.get_loss(exs, 10), g["eval_loss_warm"], what="eval warm")
    with torch.no_grad():
        res = pm.predict(exs)
        assert [r[0][0] for r in res] == g["predict_words"]
        assert [r[1][0] for r in res] == [len(w) for w in g["predict_words"]]
        ret = pm.latent_for_init(pm.hidden_to_latent(pm.encode_to_hidden(exs), False))
        assert rel_err(ret["decode_init"], g["decode_init"]) < 1e-4
        assert torch.equal(hip_tokens(pm, ret), g["predict_tokens"])
        _, _, rd, _ = pm.decode(inputs=None, encoder_outputs=None, encoder_hidden=pm.bridger.forward(ret["decode_init"]))
        assert rd["length"] == g["predict_lengths"]
        if g["model"] == "VanillaVAE":
            pm.noise = {"eps": g["gen_eps"][0]}
        else:
            pm.noise = {"eps_syn": g["gen_eps"][0], "eps_sem": g["gen_eps"][1]}
        assert [r[0][0] for r in pm.generating(len(exs), batch_size=len(exs))] == g["gen_words"]
        if g["model"] != "VanillaVAE":
            e = g["cond_sem_eps"]
            pm.noise = {"eps_syn": e[0], "eps_sem": e[1]}
            assert [r[0][0] for r in pm.conditional_generating("sem", exs)["res"]] == g["cond_sem_words"]
            e = g["cond_semonly_eps"]
            pm.noise = {"eps_syn": e[0], "eps_sem": e[1], "eps_syn_prior": e[2]}
            assert [r[0][0] for r in pm.conditional_generating("sem-only", exs)["res"]] == g["cond_semonly_words"]
        pm.noise = {}
        ex2 = product_examples(examples(g, "examples2"))
        if g["model"] in ("SyntaxVAE2", "DisentangleVAE"):
            adv = pm.eval_adv(exs, ex2)
            assert [r[0][0] for r in adv["res"]] == g["transfer_words"]
            assert [r[0][0] for r in adv["ori syn"]] == g["transfer_ori_syn"]
            assert [r[0][0] for r in adv["ref syn"]] == g["transfer_ref_syn"]
            assert [r[0][0] for r in pm.eval_syntax(ex2)] == g["eval_syntax_words"]
        if g["model"] == "VanillaVAE":
            assert [r[0][0] for r in pm.eval_adv(exs, ex2)["res"]] == g["transfer_words"]


@pytest.mark.parametrize("case", [c for c in CASES if "Vanilla" not in c])
def test_reference_helper_surface_matches_loss_dict(case):
    """get_mul_loss / get_adv_loss / get_dis_loss / _dis_loss (syntax_vae.py:321-380, enhance_syntax_vae.py:338-403):
    same names and return forms as the reference; their values are the pieces of the golden-pinned loss dict"""
    g = load_golden(case)
    _, pm = pair(g, train=False)
    exs = product_examples(examples(g))
    step = 10
    with torch.no_grad():
        want = pm.get_loss(exs, step)
        dis = pm.get_loss(exs, step, is_dis=True)
        batch, ret = pm._front(exs)
        syn_h, sem_h, B = ret["syn_hidden"], ret["sem_hidden"], len(batch)
        mul_syn, mul_sem = pm.get_mul_loss(syn_h, sem_h, batch.syn_var, batch.tgt_var)
        adv_syn, adv_sem = pm.get_adv_loss(syn_h, sem_h, batch.syn_var, batch.tgt_var)
        dis_syn, dis_sem = pm.get_dis_loss(syn_h, sem_h, batch.syn_var, batch.tgt_var)
        d = pm._dis_loss(syn_h, sem_h, batch.syn_var, batch.tgt_var)
        fwd = pm.forward(exs)   # the reference's forward() entries (syntax_vae.py:270-276, enhance_syntax_vae.py:276-288)
    assert fwd["batch_size"] == B
    assert rel_err(fwd["mul"] / B, want["MUL Loss"]) < 1e-4 and rel_err(fwd["adv"] / B, want["ADV Loss"]) < 1e-4
    assert rel_err(fwd["nll_loss"] / B, want["NLL Loss"]) < 1e-4
    assert rel_err(fwd["sem_loss"] + fwd["syn_loss"], fwd["mul"]) < 1e-5
    assert rel_err((mul_syn + mul_sem) / B, want["MUL Loss"]) < 1e-4
    assert rel_err((adv_syn + adv_sem) / B, want["ADV Loss"]) < 1e-4
    assert rel_err(dis_syn, dis["dis syn"]) < 1e-4 and rel_err(dis_sem, dis["dis sem"]) < 1e-4
    assert len(d) in (2, 4) and all(isinstance(k, str) for k in d)
    assert pm.batch_beam_decode() is None and pm.extract_variable(exs) is None if hasattr(pm, "extract_variable") else True
    with pytest.raises(NotImplementedError):
        pm.score()


@pytest.mark.parametrize("case", CASES)
def test_golden_train_losses_grads_and_discriminator(case):
    g = load_golden(case)
    for key in ("train", "train_wd"):
        if key + "_loss" not in g:
            continue
        _, pm = pair(g, train=True)
        pm.args.tgt_wd = key == "train_wd"
        set_noise(pm, g, g[key + "_randn"], g[key + "_rand"])
        exs = product_examples(examples(g))
        ret = pm.get_loss(exs, g["train_iter"])
        compare_loss_dicts(ret, g[key + "_loss"], what=key)
        torch.mean(ret["Loss"]).backward()
        named = dict(pm.named_parameters())
        gn = 0.0
        for n, gg in g[key + "_grads"].items():
            assert named[n].grad is not None, n
            assert rel_err(named[n].grad, gg) < 2e-3, (key, n, rel_err(named[n].grad, gg))
            # ... and element by element: |d| <= 2e-3 |want| + GRAD_ATOL x max|want| (torch.allclose's form)
            assert grad_excess(named[n].grad, gg, 2e-3, GRAD_ATOL) <= 1.0, (key, n, grad_excess(named[n].grad, gg, 2e-3, GRAD_ATOL))
            gn += float((named[n].grad.double() ** 2).sum())
        for n in g[key + "_nograd"]:
            assert named[n].grad is None, n
        assert abs(gn ** 0.5 - g[key + "_gradnorm"]) / g[key + "_gradnorm"] < 1e-4
        if g["model"] != "VanillaVAE":
            for p in pm.parameters():
                p.requires_grad = True
            set_noise(pm, g, g[key + "_dis_randn"], g[key + "_dis_rand"])
            d = pm.get_loss(exs, g["train_iter"], is_dis=True)
            assert rel_err(d["dis syn"], g[key + "_dis_syn"]) < 1e-4
            assert rel_err(d["dis sem"], g[key + "_dis_sem"]) < 1e-4
    _, pm = pair(g, train=True)
    pm.args.tgt_wd = False
    set_noise(pm, g, g["train_early_randn"], [])
    compare_loss_dicts(pm.get_loss(product_examples(examples(g)), 10), g["train_early_loss"], what="train early")


@pytest.mark.parametrize("case", CASES)
def test_golden_three_optimisation_steps(case):
    """zero_grad -> get_loss -> mean(Loss).backward -> clip 5.0 -> Adam, flat-buffer HIP optimiser."""
    from dss_vae_amd.train import FlatAdam, train_step
    g = load_golden(case)
    _, pm = pair(g, train=True)
    pm.args.tgt_wd = False
    opt = FlatAdam(pm, lr=0.0005, betas=(0.9, 0.995), clip_grad=5.0)
    exs = product_examples(examples(g))
    per = 1 if g["model"] == "VanillaVAE" else 2
    for i, gold in enumerate(g["steps"]):
        set_noise(pm, g, g["steps_randn"][per * i: per * i + per], [])
        ret = train_step(pm, opt, exs, g["train_iter"] + 1 + i)
        compare_loss_dicts(ret, gold["loss"], rtol=2e-4, what="step%d" % i)
        assert abs(opt.grad_norm() - gold["gradnorm"]) / gold["gradnorm"] < 2e-4
    sd = pm.state_dict()
    for k, v in g["steps_final_state"].items():
        a, b = sd[k].cpu().double(), v.double()
        assert torch.allclose(a, b, rtol=1e-4, atol=3e-5), (k, (a - b).abs().max().item())
    for n, p in pm.named_parameters():
        assert bool(p.requires_grad) == g["steps_requires_grad"][n], n


@pytest.mark.parametrize("model,fixed", [("SyntaxVAE2", True), ("SyntaxVAE2", False), ("SyntaxVAE", False),
                                         ("VanillaVAE", False), ("DisentangleVAE", False)])
def test_oracle_parity_default_dims(model, fixed):
    """default dims of configs/model_configs/syntax-vae*.yaml (E=H=300 -> h=150, D=600, tau=300, Z=100:
    none a multiple of 128) at V=2000, B=24: losses, grad-norm, greedy decode vs the CPU oracle."""
    n_src, n_tgt, B = 2000, 104, 24
    args = make_args(dec_ed=0.0, dec_rd=0.0, tgt_wd=model not in ("VanillaVAE", "DisentangleVAE"))
    om, pm = build_pair(model, args, n_src, n_tgt, seed=1, train=True)
    exs = make_examples(B, n_src, n_tgt, src_len=30, tree_len=90, fixed=fixed, seed=11)
    gen = torch.Generator().manual_seed(5)
    noise = {"eps_syn": torch.randn(B, 100, generator=gen), "eps_sem": torch.randn(B, 100, generator=gen),
             "eps": torch.randn(B, 100, generator=gen), "rand_tgt": torch.rand(B, 32 if fixed else 64, generator=gen)}
    noise["rand_tgt"] = noise["rand_tgt"][:, :max(len(e.src) for e in exs) + 2].contiguous()
    om.noise, pm.noise = dict(noise), dict(noise)
    want = om.get_loss(exs, 6000)
    got = pm.get_loss(product_examples(exs), 6000)
    compare_loss_dicts(got, want, what=model)
    torch.mean(want["Loss"]).backward()
    torch.mean(got["Loss"]).backward()
    og = dict(om.named_parameters())
    sq_ref = sq_hip = 0.0
    for n, p in pm.named_parameters():
        if og[n].grad is None:
            assert p.grad is None, n
            continue
        assert rel_err(p.grad, og[n].grad) < 5e-3, (n, rel_err(p.grad, og[n].grad))
        assert grad_excess(p.grad, og[n].grad, 5e-3, GRAD_ATOL) <= 1.0, (n, grad_excess(p.grad, og[n].grad, 5e-3, GRAD_ATOL))
        sq_ref += float((og[n].grad.double() ** 2).sum())
        sq_hip += float((p.grad.double() ** 2).sum())
    assert abs(sq_hip ** 0.5 - sq_ref ** 0.5) / sq_ref ** 0.5 < 1e-4
    om.eval(), pm.eval()
    with torch.no_grad():
        t_ref = om.predict_tokens(exs)
        ret = pm.latent_for_init(pm.hidden_to_latent(pm.encode_to_hidden(product_examples(exs)), False))
        assert torch.equal(pm.decode_tokens(ret).cpu(), t_ref)


def test_smoke_entry():
    import __graft_entry__
    __graft_entry__.smoke()
