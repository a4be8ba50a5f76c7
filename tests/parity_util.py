"""Shared parity helpers (test infrastructure): build the CPU oracle and the HIP product model with
identical parameters / inputs / injected noise and compare them.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import copy
import os
import sys
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import oracle  # noqa: E402

BASE_ARGS = dict(
    model_select="SyntaxVAE2", task_type="SyntaxVAE2", src_max_time_step=35, dec_ratio=3.2, tgt_max_time_step=100,
    decode_max_time_step=35, cuda=True, eval_mode="bleu", eval_bs=50, dev_item="ELBO", warm_up=1000, sort_key="src",
    embed_size=300, hidden_size=300, num_layers=2, rnn_drop=0.3, rnn_type="gru", enc_embed_dim=300,
    enc_hidden_dim=300, enc_num_layers=2, bidirectional=True, mapper_type="link", dec_embed_dim=300,
    dec_hidden_dim=300, dec_num_layers=2, latent_size=100, dis_train=True, sample_size=5, share_embed=True,
    use_attention=False, enc_ed=0.0, enc_rd=0.0, dec_ed=0.3, dec_rd=0.3, unk_rate=0.25, k=0.0025, x0=5000,
    epochs=10, anneal_function="logistic", unk_schedule="fixed", src_wd=False, tgt_wd=True, peak_anneal=False,
    init_step_kl_weight=None, stop_clip_kl=0.3, reload_model=False, kl_factor=0.5, mul_syn=0.3, mul_sem=0.3,
    syn_to_sem=0.3, sem_to_syn=0.3, infer_weight=1.0, syn_to_sent=0.1, sem_to_sent=0.1, syn_weight=1.0,
    sem_weight=1.0, eval_adv=False, eval_mul=False, report=None,
    # SyntaxVAE (configs/model_configs/syntax-vae.yaml) keys
    adv_syn=0.3, adv_sem=0.3, inf_syn=0.2, inf_sem=0.2)


def make_args(**over):
    """model_configs of configs/model_configs/syntax-vae2.yaml (+ syntax-vae.yaml keys), overridable."""
    d = dict(BASE_ARGS)
    d.update(over)
    return Namespace(**d)


def make_vocab(n_src, n_tgt, cls_entry=oracle.OVocabEntry, cls_vocab=oracle.OVocab):
    src, tgt = cls_entry(), cls_entry()
    for i in range(n_src - 4):
        src.add("w%d" % i)
    for i in range(n_tgt - 4):
        tgt.add("t%d" % i)
    return cls_vocab(src=src, tgt=tgt)


def make_examples(B, n_src, n_tgt, src_len=30, tree_len=90, fixed=True, seed=1234, cls=oracle.OExample):
    """SURVEY 8d synthetic batch: word ids Zipf(1.1) clipped to [4,V), tree ids uniform [4,V_t)."""
    rng = np.random.default_rng(seed)
    exs = []
    for _ in range(B):
        L = src_len if fixed else int(rng.integers(5, src_len + 1))
        TL = tree_len if fixed else 3 * L
        w = np.clip(rng.zipf(1.1, L) + 3, 4, n_src - 1)
        t = rng.integers(4, n_tgt, TL)
        exs.append(cls(["w%d" % (int(i) - 4) for i in w], ["t%d" % (int(i) - 4) for i in t]))
    exs.sort(key=lambda e: -len(e.src))
    return exs


def product_vocab(n_src, n_tgt):
    from dss_vae_amd.structs import Vocab, VocabEntry
    return make_vocab(n_src, n_tgt, VocabEntry, lambda src, tgt: Vocab(src=src, tgt=tgt))


def product_examples(exs):
    from dss_vae_amd.structs import PlainExample
    return [PlainExample(src=list(e.src), tgt=None if e.tgt is None else list(e.tgt)) for e in exs]


def build_pair(model_name, args, n_src, n_tgt, seed=0, state_dict=None, train=False):
    """(oracle model on CPU, product model on cuda:0) with identical parameters."""
    from dss_vae_amd.models import MODEL_CLS
    om = oracle.ORACLE_MODELS[model_name](copy.deepcopy(args), make_vocab(n_src, n_tgt))
    if state_dict is None:
        oracle.init_uniform_(om, seed=seed)
        state_dict = om.state_dict()
    else:
        om.load_state_dict(state_dict, strict=True)
    pm = MODEL_CLS[model_name](args=copy.deepcopy(args), vocab=product_vocab(n_src, n_tgt))
    pm.load_state_dict(state_dict, strict=True)
    pm.cuda()
    om.train(train)
    pm.train(train)
    return om, pm


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    denom = max(b.abs().max().item(), 1e-12) if b.numel() else 1.0
    return ((a - b).abs().max().item() / denom) if b.numel() else 0.0


def grad_excess(got, want, rtol=2e-3, atol_frac=1e-6):
    """element-wise bar for a gradient tensor: max over elements of |got - want| / (atol + rtol |want|) with atol =
    atol_frac x max|want| (<= 1 passes, as torch.allclose); a tensor-max-normalised error lets a small-magnitude row be
    wrong by more than its own size"""
    a = torch.as_tensor(got).detach().double().cpu()
    b = torch.as_tensor(want).detach().double().cpu()
    if not b.numel():
        return 0.0
    atol = atol_frac * max(b.abs().max().item(), 1e-30)
    return ((a - b).abs() / (atol + rtol * b.abs())).max().item()


def compare_loss_dicts(got, want, rtol=1e-4, what=""):
    """north_star bar: per-term losses within 1e-4 relative in fp32."""
    assert set(got) == set(want), (what, sorted(got), sorted(want))
    worst = 0.0
    for k in want:
        e = rel_err(got[k], want[k])
        worst = max(worst, e)
        assert e <= rtol, "%s: loss term %r differs: rel err %.3e (got %s, want %s)" % (what, k, e, got[k], want[k])
    return worst


def words(tokens, entry):
    return [oracle.ids_to_words(tokens[:, i].tolist(), entry) for i in range(tokens.size(1))]


def smoke_check(verbose=False):
    """tiny SyntaxVAE2: one train-mode loss + backward and one greedy decode, HIP vs oracle."""
    args = make_args(embed_size=32, hidden_size=32, enc_embed_dim=32, enc_hidden_dim=32, dec_embed_dim=32,
                     dec_hidden_dim=32, latent_size=16, src_max_time_step=12, tgt_max_time_step=30, dec_ed=0.0,
                     dec_rd=0.0, tgt_wd=False)
    n_src, n_tgt, B = 200, 40, 8
    om, pm = build_pair("SyntaxVAE2", args, n_src, n_tgt, seed=3, train=True)
    exs = make_examples(B, n_src, n_tgt, src_len=9, tree_len=27, fixed=False, seed=5)
    g = torch.Generator().manual_seed(7)
    eps = {"eps_syn": torch.randn(B, 16, generator=g), "eps_sem": torch.randn(B, 16, generator=g)}
    om.noise, pm.noise = dict(eps), dict(eps)
    want = om.get_loss(exs, 6000)
    got = pm.get_loss(product_examples(exs), 6000)
    worst = compare_loss_dicts(got, want, rtol=1e-4, what="smoke")
    torch.mean(want["Loss"]).backward()
    torch.mean(got["Loss"]).backward()
    og = dict(om.named_parameters())
    gerr = 0.0
    for n, p in pm.named_parameters():
        if og[n].grad is not None:
            assert p.grad is not None, n
            gerr = max(gerr, rel_err(p.grad, og[n].grad))
    assert gerr < 2e-3, gerr
    om.eval(), pm.eval()
    with torch.no_grad():
        t_ref = om.predict_tokens(exs)
        t_hip = pm.decode_tokens(pm.latent_for_init(pm.hidden_to_latent(pm.encode_to_hidden(product_examples(exs)),
                                                                        False))).cpu()
    assert torch.equal(t_ref, t_hip), "greedy decode differs"
    if verbose:
        print("smoke ok: worst loss rel err %.2e, worst grad rel err %.2e, greedy tokens identical %s"
              % (worst, gerr, tuple(t_hip.shape)))
    return worst
