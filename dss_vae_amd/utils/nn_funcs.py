"""Host-side helpers of the hot path (reference: dss_vae/utils/nn_funcs.py:13-72,122-174,
utils/schedule_funs.py:43-57).  Tensors are created on the MI355X directly."""
import weakref

import numpy as np
import torch

from .. import ops


def device():
    if not torch.cuda.is_available():
        raise RuntimeError("dss_vae_amd runs on the MI355X only: no HIP device visible and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def kl_anneal_function(anneal_function, step, k, x0):
    if anneal_function == "fixed":
        return 1.0
    if anneal_function == "logistic":
        return float(1 / (1 + np.exp(-k * (step - x0))))
    if anneal_function == "sigmoid":
        return float(1 / (1 + np.exp(0.001 * (x0 - step))))
    if anneal_function == "negative-sigmoid":
        return float(1 / (1 + np.exp(-0.001 * (x0 - step))))
    if anneal_function == "linear":
        return min(1, step / x0)


def wd_anneal_function(unk_max, anneal_function, step, k, x0):
    return unk_max * kl_anneal_function(anneal_function, step, k, x0)


def word2id(sents, vocab):
    if type(sents[0]) == list:
        return [[vocab[w] for w in s] for s in sents]
    return [vocab[w] for w in sents]


def robust_id2word(sents, vocab):
    res = []
    for w in sents:
        if w == vocab.sos_id or w == vocab.pad_id:
            continue
        if w == vocab.eos_id:
            break
        res.append(vocab.id2word[w])
    return res


def id2word(sents, vocab):
    if type(sents[0]) == list:
        return [robust_id2word(s, vocab) for s in sents]
    return robust_id2word(sents, vocab)


_WORD_TABLES = weakref.WeakKeyDictionary()   # (beside the vocabulary, not on it: a VocabEntry is pickled into checkpoints)


def _word_table(vocab):
    """id -> word as a numpy object array (built once per vocabulary size: `add` grows the vocabulary)"""
    tab = _WORD_TABLES.get(vocab)
    if tab is None or len(tab) != len(vocab.id2word):
        tab = np.empty(len(vocab.id2word), dtype=object)
        for i, w in vocab.id2word.items():
            tab[i] = w
        _WORD_TABLES[vocab] = tab
    return tab


def rows_to_words(rows, vocab):
    """robust_id2word over every row of an id matrix [B, L] (numpy / list of lists): the same word lists -- cut at the
    first </s>, <s> and <pad> dropped -- with the per-token work in numpy (the greedy decodes hand back thousands of
    ids per batch; the per-token Python loop was a millisecond per 100 tree sequences)"""
    rows = np.asarray(rows)
    if rows.ndim != 2 or rows.shape[0] == 0:
        return [robust_id2word(list(r), vocab) for r in rows]
    tab = _word_table(vocab)
    if rows.size and (rows.min() < 0 or rows.max() >= len(tab)):
        return [robust_id2word(r, vocab) for r in rows.tolist()]   # (raises the KeyError the plain loop raises)
    is_eos = rows == vocab.eos_id
    cut = np.where(is_eos.any(axis=1), is_eos.argmax(axis=1), rows.shape[1])
    keep = (rows != vocab.sos_id) & (rows != vocab.pad_id)
    return [tab[r[:c][k[:c]]].tolist() for r, c, k in zip(rows, cut, keep)]


def rows_to_sentences(rows, vocab):
    """the evaluators' result format for a token matrix: [[[words], [len(words)]], ...] (vae/base_vae.py:176-190)"""
    return [[[w], [len(w)]] for w in rows_to_words(rows, vocab)]


def padding_input(sents, pad_token, max_len=-1):
    """right-pad token lists to a common length (utils/nn_funcs.py:122-129)"""
    if max_len == -1:
        max_len = max(len(s) for s in sents)
    return [list(s) + [pad_token] * (max_len - len(s)) for s in sents]


def to_input_variable(sequences, vocab, max_len=-1, cuda=True, training=True, append_boundary_sym=False,
                      batch_first=False, **_):
    """list of token lists -> padded int64 ids on the device ([B,L] if batch_first else [L,B])."""
    if not isinstance(sequences[0], list):
        sequences = [sequences]
    if append_boundary_sym:
        sequences = [["<s>"] + list(s) + ["</s>"] for s in sequences]
    width = max(len(s) for s in sequences) if max_len == -1 else max_len
    pad = vocab["<pad>"]
    rows = np.full((len(sequences), width), pad, dtype=np.int64)
    table = getattr(vocab, "word2id", None)
    if isinstance(table, dict):   # VocabEntry.__getitem__ without a Python frame per word
        get, unk = table.get, vocab.unk_id
        for i, s in enumerate(sequences):
            rows[i, :len(s)] = [get(w, unk) for w in s]
    else:
        for i, s in enumerate(sequences):
            rows[i, :len(s)] = [vocab[w] for w in s]
    ids = torch.from_numpy(rows).to(device(), non_blocking=True)
    return ids if batch_first else ids.t().contiguous()


def unk_replace(input_sequence, dropoutr, vocab, rand=None, seed=0):
    """word dropout (device kernel); ``rand`` injects the uniform draw for parity tests."""
    return ops.unk_replace(input_sequence, dropoutr, vocab, rand=rand, seed=seed)


def to_var(x, volatile=False):
    return x.to(device())
