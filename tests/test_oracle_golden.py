"""Pin the CPU oracle against golden vectors captured from the imported reference
(tests/golden/make_golden.py).  fp32 CPU both sides -> tolerance 1e-6 relative."""
import copy

import pytest
import torch

import oracle
from conftest import CASES, golden_args, load_golden

RTOL = 2e-6


def build(g, train=False):
    args = golden_args(g)
    vocab = oracle.OVocab(oracle.OVocabEntry("w%d" % i for i in range(g["n_src"] - 4)),
                          oracle.OVocabEntry("t%d" % i for i in range(g["n_tgt"] - 4)))
    m = oracle.ORACLE_MODELS[g["model"]](args, vocab)
    m.load_state_dict(g["state_dict"], strict=True)
    m.train(train)
    return m, args


def examples(g, key="examples"):
    return [oracle.OExample(s, t) for s, t in zip(g[key]["src"], g[key]["tgt"])]


def close(a, b, rtol=RTOL, what="", atol=1e-7):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.allclose(a, b, rtol=rtol, atol=atol), (what, (a - b).abs().max().item())


def check_loss(ret, gold, what):
    assert set(ret) == set(gold), what
    for k in gold:
        close(ret[k], gold[k], what=what + ":" + k)


def set_noise(m, g, randn, rand):
    m.noise = {}
    if g["model"] == "VanillaVAE":
        if randn:
            m.noise["eps"] = randn[0]
    elif randn:
        m.noise["eps_syn"], m.noise["eps_sem"] = randn[0], randn[1]
    if rand:
        m.noise["rand_tgt"] = rand[0]


def words(tokens, entry):
    return [oracle.ids_to_words(tokens[:, i].tolist(), entry) for i in range(tokens.size(1))]


def test_misc_host_functions():
    g = load_golden("misc")
    for fn, step, val in g["anneal"]:
        assert oracle.anneal_weight(fn, step, 0.0025, 5000) == pytest.approx(val, rel=1e-12)
    v = oracle.OVocabEntry("w%d" % i for i in range(20))
    ids = oracle.tensorise([["w1", "w2", "zzz", "w3"], ["w4"], ["w5", "w6"]], v, boundary=True)
    assert torch.equal(ids, g["tensorise_ids"])
    assert torch.equal(oracle.word_dropout(ids, 0.5, v, g["unk_rand"]), g["unk_out"])


def test_gru_equations_match_reference_encoder():
    """explicit GRU equations + packing semantics vs the reference RNNEncoder's nn.GRU output."""
    g = load_golden("misc")
    sd, ids, lens = g["gru_state"], g["gru_ids"], g["gru_lens"]
    x = sd["embedding.embeddings.weight"][ids]
    B = ids.size(0)
    hn = []
    for layer in range(2):
        outs = []
        for rev, suf in ((False, ""), (True, "_reverse")):
            p = [sd["rnn.%s_l%d%s" % (n, layer, suf)] for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            o, h = oracle.gru_loop(x, torch.zeros(B, 10), *p, lengths=lens, reverse=rev)
            outs.append(o)
            hn.append(h)
        x = torch.cat(outs, -1)
    close(x, g["gru_out"], what="gru out")
    close(torch.stack(hn), g["gru_hn"], what="gru h_n")


@pytest.mark.parametrize("case", CASES)
def test_eval_losses_and_decodes(case):
    g = load_golden(case)
    m, args = build(g)
    exs = examples(g)
    check_loss(m.get_loss(exs, g["train_iter"]), g["eval_loss"], "eval")
    check_loss(m.get_loss(exs, 10), g["eval_loss_warm"], "eval warm")
    with torch.no_grad():
        toks = m.predict_tokens(exs)
        assert torch.equal(toks, g["predict_tokens"])
        assert words(toks, m.vocab.src) == g["predict_words"]
        ret = m.latent_for_init(m.hidden_to_latent(m.encode_to_hidden(exs), False))
        close(ret["decode_init"], g["decode_init"], what="decode_init")
        # prior generation with the recorded draws
        if g["model"] == "VanillaVAE":
            m.noise = {"eps": g["gen_eps"][0]}
        else:
            m.noise = {"eps_syn": g["gen_eps"][0], "eps_sem": g["gen_eps"][1]}
        B = len(exs)
        toks = m.decode_to_sentence(m.latent_for_init(m.sample_latent(B)))[1]
        assert words(toks, m.vocab.src) == g["gen_words"]
        if g["model"] != "VanillaVAE":
            m.noise = {"eps_syn": g["cond_sem_eps"][0], "eps_sem": g["cond_sem_eps"][1]}
            assert words(m.conditional_generating("sem", exs), m.vocab.src) == g["cond_sem_words"]
            e = g["cond_semonly_eps"]
            m.noise = {"eps_syn": e[0], "eps_sem": e[1], "eps_syn_prior": e[2]}
            assert words(m.conditional_generating("sem-only", exs), m.vocab.src) == g["cond_semonly_words"]
        m.noise = {}
        if g["model"] in ("SyntaxVAE2", "DisentangleVAE"):
            ex2 = examples(g, "examples2")
            toks, a, b = m.transfer_tokens(exs, ex2)
            head = m.mul_syn if g["model"] == "SyntaxVAE2" else m.sup_syn
            assert words(toks, m.vocab.src) == g["transfer_words"]
            assert words(head.predict(a["syn_hidden"]), m.vocab.tgt) == g["transfer_ori_syn"]
            assert words(head.predict(b["syn_hidden"]), m.vocab.tgt) == g["transfer_ref_syn"]
            assert words(m.syntax_tokens(ex2), m.vocab.tgt) == g["eval_syntax_words"]


@pytest.mark.parametrize("case", CASES)
def test_train_losses_and_grads(case):
    g = load_golden(case)
    for key in ("train", "train_wd"):
        if key + "_loss" not in g:
            continue
        m, args = build(g, train=True)
        args.tgt_wd = key == "train_wd"
        set_noise(m, g, g[key + "_randn"], g[key + "_rand"])
        ret = m.get_loss(examples(g), g["train_iter"])
        check_loss(ret, g[key + "_loss"], key)
        torch.mean(ret["Loss"]).backward()
        named = dict(m.named_parameters())
        for n, gg in g[key + "_grads"].items():
            assert named[n].grad is not None, n
            close(named[n].grad, gg, rtol=2e-5, what=key + " grad " + n)
        for n in g[key + "_nograd"]:
            assert named[n].grad is None, n
        if g["model"] != "VanillaVAE":
            for p in m.parameters():
                p.requires_grad = True
            set_noise(m, g, g[key + "_dis_randn"], g[key + "_dis_rand"])
            d = m.get_loss(examples(g), g["train_iter"], is_dis=True)
            close(d["dis syn"], g[key + "_dis_syn"], what="dis syn")
            close(d["dis sem"], g[key + "_dis_sem"], what="dis sem")
    m, args = build(g, train=True)
    args.tgt_wd = False
    set_noise(m, g, g["train_early_randn"], [])
    check_loss(m.get_loss(examples(g), 10), g["train_early_loss"], "train early")


@pytest.mark.parametrize("case", CASES)
def test_three_optimisation_steps(case):
    """bin/train.py step order incl. clip 5.0 and Adam(lr 5e-4, betas (0.9, 0.995))."""
    g = load_golden(case)
    m, args = build(g, train=True)
    args.tgt_wd = False
    opt = torch.optim.Adam(m.parameters(), lr=0.0005, betas=(0.9, 0.995))
    exs = examples(g)
    per = 1 if g["model"] == "VanillaVAE" else 2
    for i, gold in enumerate(g["steps"]):
        set_noise(m, g, g["steps_randn"][per * i: per * i + per], [])
        ret, gn = oracle.oracle_train_step(m, opt, exs, g["train_iter"] + 1 + i)
        check_loss(ret, gold["loss"], "step%d" % i)
        assert float(gn) == pytest.approx(gold["gradnorm"], rel=1e-5)
    sd = m.state_dict()
    for k, v in g["steps_final_state"].items():
        # Adam's first steps are ~lr*sign(g): where g ~ 0 rounding noise is amplified; lr = 5e-4
        close(sd[k], v, rtol=1e-5, atol=1e-5, what="final " + k)
    for n, p in m.named_parameters():
        assert bool(p.requires_grad) == g["steps_requires_grad"][n], n


def test_lstm_fixture_is_what_nn_lstm_computes():
    """rnn_cell='lstm' (SURVEY 8f N4): the reference's RNNEncoder is embedding -> pack -> torch.nn.LSTM -> pad; the
    CPU restatement of that (plain nn.LSTM here) reproduces the fixture the reference wrote, so the GPU test's
    expectations are pinned from both sides."""
    import torch.nn as nn
    from conftest import load_golden
    g = load_golden("lstm_encdec")
    sd = g["enc_state"]
    rnn = nn.LSTM(g["E"], g["H"], g["L"], batch_first=True, bidirectional=True)
    rnn.load_state_dict({k[len("rnn."):]: v for k, v in sd.items() if k.startswith("rnn.")})
    emb = sd["embedding.embeddings.weight"][g["enc_ids"]]
    packed = nn.utils.rnn.pack_padded_sequence(emb, g["enc_lengths"], batch_first=True)
    out, (hn, cn) = rnn(packed)
    out, _ = nn.utils.rnn.pad_packed_sequence(out, batch_first=True)
    assert torch.allclose(out, g["enc_out"], atol=1e-6) and torch.allclose(hn, g["enc_hn"], atol=1e-6)
    assert torch.allclose(cn, g["enc_cn"], atol=1e-6)
