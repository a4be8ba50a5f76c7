"""Synthetic workloads of BASELINE.json / SURVEY 8(d): model_configs of the shipped yaml files
(configs/model_configs/syntax-vae.yaml, syntax-vae2.yaml, read as text) and seeded sentence / tree
batches (word ids Zipf(1.1) clipped to [4,V), tree ids uniform in [4,V_t)).  Used by bench.py and
the training driver; the parity tests build the same batches for the oracle."""
from argparse import Namespace

import numpy as np

from .structs import PlainExample, Vocab, VocabEntry

MODEL_CONFIGS = dict(
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
    # keys only configs/model_configs/syntax-vae.yaml (SyntaxVAE) has
    adv_syn=0.3, adv_sem=0.3, inf_syn=0.2, inf_sem=0.2)

# BASELINE.json configs[3]: hidden 1024, latent 256 (D = 2048, tau = 1024 under mapper_type: link)
STRESS_OVERRIDES = dict(embed_size=300, hidden_size=1024, enc_hidden_dim=1024, dec_hidden_dim=1024, latent_size=256)


def model_args(**over):
    d = dict(MODEL_CONFIGS)
    d.update(over)
    return Namespace(**d)


def word_lists(B, n_src, n_tgt, src_len=30, tree_len=90, fixed=True, seed=1234):
    """-> list of (src words, tree symbols), sorted by -len(src) as Dataset.batch_iter does."""
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(B):
        L = src_len if fixed else int(rng.integers(5, src_len + 1))
        TL = tree_len if fixed else 3 * L
        w = np.clip(rng.zipf(1.1, L) + 3, 4, n_src - 1)
        t = rng.integers(4, n_tgt, TL)
        rows.append((["w%d" % (int(i) - 4) for i in w], ["t%d" % (int(i) - 4) for i in t]))
    rows.sort(key=lambda r: -len(r[0]))
    return rows


def vocab(n_src, n_tgt):
    src, tgt = VocabEntry(), VocabEntry()
    for i in range(n_src - 4):
        src.add("w%d" % i)
    for i in range(n_tgt - 4):
        tgt.add("t%d" % i)
    return Vocab(src=src, tgt=tgt)


def examples(B, n_src, n_tgt, **kw):
    return [PlainExample(src=s, tgt=t) for s, t in word_lists(B, n_src, n_tgt, **kw)]
