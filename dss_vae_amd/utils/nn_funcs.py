"""Host-side helpers of the hot path (reference: dss_vae/utils/nn_funcs.py:13-72,122-174,
utils/schedule_funs.py:43-57).  Tensors are created on the MI355X directly."""
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
    for i, s in enumerate(sequences):
        rows[i, :len(s)] = [vocab[w] for w in s]
    ids = torch.from_numpy(rows).to(device(), non_blocking=True)
    return ids if batch_first else ids.t().contiguous()


def unk_replace(input_sequence, dropoutr, vocab, rand=None, seed=0):
    """word dropout (device kernel); ``rand`` injects the uniform draw for parity tests."""
    return ops.unk_replace(input_sequence, dropoutr, vocab, rand=rand, seed=seed)


def to_var(x, volatile=False):
    return x.to(device())
