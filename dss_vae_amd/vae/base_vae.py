"""BaseVAE (reference: dss_vae/vae/base_vae.py:13-220) on the HIP ops.  Same public methods and
``ret`` dict keys; ``self.noise`` is the RNG injection hook used by the parity tests (keys:
eps / eps_syn / eps_sem / eps_syn_prior / rand_src / rand_tgt), absent keys are drawn on device."""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..networks.rnn_base import GLOBAL_SEEDS
from ..utils.nn_funcs import device, kl_anneal_function, rows_to_sentences, to_input_variable, unk_replace, wd_anneal_function


class Lengths(list):
    """the sentence lengths of a minibatch as the host list the reference's interfaces pass around
    (RNNEncoder.forward(input_var, input_lengths)), carrying the same numbers as a device int32 tensor (`dev`), so that
    the packed recurrences never build one from the list (a host-to-device copy per step, and not capturable)"""
    dev = None


def _lengths(values, dev):
    out = Lengths(int(v) for v in values)
    out.dev = torch.from_numpy(np.asarray(out, dtype=np.int32)).pin_memory().to(dev, non_blocking=True)
    return out


def _n_keep(lengths, T):
    """number of (sentence, step) rows of a [B, T] teacher-forced layout whose target is not <pad> (targets are
    ids[:, 1:] of <s> w_1 .. w_L </s> <pad>..: L + 1 real targets per sentence), or None when no row is padding"""
    n = int(np.minimum(np.asarray(lengths, dtype=np.int64) + 1, T).sum())
    return None if n == len(lengths) * T else n


def _keep_rows(lengths, T, dev):
    """Row indices, in the flattened [B, T] layout of a teacher-forced decoder, whose target is not <pad>.  None when
    no row is padding.  Their NUMBER is known on the host from the lengths, so a ragged batch costs no device sync;
    the list itself is written on the device (ops.keep_rows); the scoring path projects only these rows onto the
    vocabulary (decoder/rnn_decoder.py: score_many)."""
    n = _n_keep(lengths, T)
    if n is None:
        return None
    lens = lengths if isinstance(lengths, Lengths) else _lengths(lengths, dev)
    return ops.keep_rows(lens.dev, T, n)


class DeviceBatch(object):
    """Pre-tensorised minibatch resident in HBM (kills the per-step Python tensorisation of
    utils/nn_funcs.py:143-174).  Accepted wherever a list of examples is.  Built from token lists (this
    constructor: evaluation, tests) or cut on the device from the dataset's id cache (structs/id_cache.py: training)."""

    def __init__(self, examples, vocab, with_tgt=True):
        self.examples = examples
        self.n = len(examples)
        words = [e.src for e in examples]
        self.src = to_input_variable(words, vocab.src, batch_first=True)
        self.length = _lengths([len(w) for w in words], self.src.device)
        self.tgt_var = to_input_variable(words, vocab.src, append_boundary_sym=True, batch_first=True)
        self.tgt_var._keep_rows = _keep_rows(self.length, self.tgt_var.size(1) - 1, self.tgt_var.device)
        self.syn_var, self.tree_length = None, None
        if with_tgt and examples[0].tgt is not None:
            trees = [e.tgt for e in examples]
            self.syn_var = to_input_variable(trees, vocab.tgt, append_boundary_sym=True, batch_first=True)
            self.tree_length = _lengths([len(t) for t in trees], self.src.device)
            self.syn_var._keep_rows = _keep_rows(self.tree_length, self.syn_var.size(1) - 1, self.syn_var.device)

    @classmethod
    def from_ids(cls, src, length, tgt_var, syn_var=None, tree_length=None):
        """from tensors already on the device (structs/id_cache.py); length / tree_length: Lengths"""
        self = cls.__new__(cls)
        self.examples, self.n = None, src.size(0)
        self.src, self.length, self.tgt_var, self.syn_var, self.tree_length = src, length, tgt_var, syn_var, tree_length
        tgt_var._keep_rows = _keep_rows(length, tgt_var.size(1) - 1, src.device)
        if syn_var is not None:
            syn_var._keep_rows = _keep_rows(tree_length, syn_var.size(1) - 1, src.device)
        return self

    def signature(self):
        """what a captured training step's launches depend on besides the values: the shapes"""
        def keep(t):
            k = None if t is None else getattr(t, "_keep_rows", None)
            return None if k is None else k.numel()
        return (self.n, tuple(self.src.shape), tuple(self.tgt_var.shape),
                None if self.syn_var is None else tuple(self.syn_var.shape), keep(self.tgt_var), keep(self.syn_var))

    def clone(self):
        """a copy with storage of its own (the static inputs of a captured step)"""
        def cp(t):
            if t is None:
                return None
            out = t.clone()
            k = getattr(t, "_keep_rows", None)
            out._keep_rows = None if k is None else k.clone()
            return out

        def cl(l):
            if l is None:
                return None
            out = Lengths(l)
            out.dev = l.dev.clone()
            return out
        other = DeviceBatch.__new__(DeviceBatch)
        other.examples, other.n = self.examples, self.n
        other.src, other.tgt_var, other.syn_var = cp(self.src), cp(self.tgt_var), cp(self.syn_var)
        other.length, other.tree_length = cl(self.length), cl(self.tree_length)
        return other

    def copy_from(self, other):
        """overwrite this batch's device tensors with another batch's of the same signature (stream-ordered)"""
        for a, b in ((self.src, other.src), (self.tgt_var, other.tgt_var), (self.syn_var, other.syn_var)):
            if a is None:
                continue
            a.copy_(b, non_blocking=True)
            ka = getattr(a, "_keep_rows", None)
            if ka is not None:
                ka.copy_(b._keep_rows, non_blocking=True)
        for a, b in ((self.length, other.length), (self.tree_length, other.tree_length)):
            if a is not None:
                a[:] = list(b)
                a.dev.copy_(b.dev, non_blocking=True)
        self.examples = other.examples

    def __len__(self):
        return self.n


class Linear(nn.Linear):
    """nn.Linear parameters (same state_dict keys), MFMA GEMM forward/backward."""

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class BaseVAE(nn.Module):
    def __init__(self, args, vocab, name="Sentence VAE"):
        super().__init__()
        self.args, self.vocab, self.name = args, vocab, name
        self.step_kl_weight = args.init_step_kl_weight
        self.noise = {}

    def score(self, **kwargs):  # base_vae.py:23-24
        raise NotImplementedError

    def batch_beam_decode(self, **kwargs):  # a no-op in the reference (base_vae.py:101-102): the VAEs decode greedily
        pass

    def base_information(self):
        a = self.args
        return "layers:{}\ntypes:{}\nembed:{}\nhidden:{}\nlatent:{}\nkl_factor:{}\nunk_rate:{}\nx0:{}\n".format(
            a.num_layers, a.rnn_type, a.embed_size, a.hidden_size, a.latent_size, a.kl_factor, a.unk_rate, a.x0)

    # ---- noise ---------------------------------------------------------------------------------
    def _randn(self, key, shape):
        v = self.noise.get(key)
        if v is not None:
            return v.to(device(), dtype=torch.float32)
        return torch.randn(shape, device=device())

    def _word_dropout(self, ids, key):
        return unk_replace(ids, self.step_unk_rate, self.vocab.src, rand=self.noise.get(key),
                           seed=GLOBAL_SEEDS.next())

    # ---- batches ---------------------------------------------------------------------------------
    def _batch(self, examples, with_tgt=True):
        if isinstance(examples, DeviceBatch):
            return examples
        if not isinstance(examples, list):
            examples = [examples]
        return DeviceBatch(examples, self.vocab, with_tgt=with_tgt)

    def predict(self, examples, to_word=True):
        ret = self.encode_to_hidden(examples)
        ret = self.hidden_to_latent(ret=ret, is_sampling=self.training)
        ret = self.latent_for_init(ret=ret)
        return self.decode_to_sentence(ret=ret)

    def predict_async(self, examples):
        """predict() in two halves for callers that decode batch after batch (metrics/evaluation.py): this one
        enqueues encode -> latent -> greedy decode and the copy of the token matrix to pinned host memory and
        returns a handle; `finish_predict(handle)` waits for that copy only and returns what predict() returns.
        Enqueue batch k+1 before finishing batch k and the host's id -> word work runs under the GPU's decode."""
        ret = self.latent_for_init(self.hidden_to_latent(self.encode_to_hidden(examples), is_sampling=self.training))
        toks = self.decode_tokens(ret).t()
        host = torch.empty(toks.shape, dtype=toks.dtype, pin_memory=True)
        host.copy_(toks, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return host, done

    def finish_predict(self, handle):
        return self._to_sentences(*handle)

    def generating(self, sample_num, batch_size=50):
        """base_vae.py:69-85.  Software-pipelined: the token matrix of batch k leaves for pinned host memory
        behind its decode, batch k+1 is enqueued, and only then does the host wait for batch k and turn ids into
        words -- the GPU decodes while the host converts.  Noise is drawn in the same order as without."""
        batch_size = batch_size if sample_num >= batch_size else sample_num
        res, pending = [], None
        for _ in range(int((sample_num - 1) / batch_size) + 1):
            toks = self.decode_tokens(self.latent_for_init(self.sample_latent(batch_size))).t()
            host = torch.empty(toks.shape, dtype=toks.dtype, pin_memory=True)
            host.copy_(toks, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            if pending is not None:
                res.extend(self._to_sentences(*pending))
            pending = (host, done)
        if pending is not None:
            res.extend(self._to_sentences(*pending))
        return res

    def _to_sentences(self, host_tokens, done):
        done.synchronize()
        return rows_to_sentences(host_tokens.numpy(), self.vocab.src)

    unsupervised_generating = generating

    def encode_to_hidden(self, examples, need_sort=False, **kwargs):
        batch = self._batch(examples, with_tgt=False)
        B = len(batch)
        if not need_sort:
            _, hidden = self.encode(input_var=batch.src, length=batch.length)
            return {"outputs": None, "hidden": hidden, "length": batch.length, "batch_size": B}
        lens = batch.length
        order = sorted(range(B), key=lambda i: -lens[i])
        back = [0] * B
        for new, old in enumerate(order):
            back[old] = new
        idx = torch.as_tensor(order, device=batch.src.device)
        ids = batch.src.index_select(0, idx)
        if self.training and self.args.src_wd:
            ids = self._word_dropout(ids, "rand_src")
        _, hidden = self.encode(ids, [lens[i] for i in order])
        bidx = torch.as_tensor(back, device=hidden.device)
        hidden = hidden.index_select(1, bidx) if hidden.dim() > 2 else hidden.index_select(0, bidx)
        return {"outputs": None, "hidden": hidden, "length": lens, "batch_size": B}

    def decode_tokens(self, ret):
        """greedy token matrix [max_len,B] (device) for the sentence decoder."""
        h0 = self.bridger.forward(input_tensor=ret["decode_init"])
        return self.decoder.greedy_tokens(h0)[0]

    def decode_to_sentence(self, ret):
        return rows_to_sentences(self.decode_tokens(ret).t().cpu().numpy(), self.vocab.src)

    GRAPH_SAFE = False  # True: one training step of the class can be captured into a HIP graph (train.GraphedTrainer)

    def step_scalars(self, train_iter):
        """the host scalars of a training step that change with train_iter: annealed word-dropout rate
        (syntax_vae.py:278) and KL weight x kl_factor (:286).  get_loss reads them from here, and so does the replay of
        a captured step, which writes them to the graph's per-step device slots (ops.StepScalars)."""
        a = self.args
        return {"unk_rate": wd_anneal_function(getattr(self, "unk_rate", 0.0), a.unk_schedule, train_iter, a.k, a.x0),
                "kl_weight": self.get_kl_weight(train_iter) * a.kl_factor}

    def get_kl_weight(self, step):
        if self.step_kl_weight is None:
            return kl_anneal_function(self.args.anneal_function, step, self.args.k, self.args.x0)
        return self.step_kl_weight

    def compute_kl_loss(self, mean, logv, step):
        """standalone KL (base_vae.py:198-201); the models fuse it with the reparameterisation."""
        _, kl = ops.reparam_kl(mean, logv, None)
        return kl, self.get_kl_weight(step)

    @classmethod
    def load(cls, path, map_location="cpu"):
        """model.bin ({'args','vocab','state_dict'}, written here or by the reference's save) -> model on the
        host; call .cuda() to run.  (models/__init__.py:36-43 calls MODEL_CLS[name].load(path); the
        reference's VAE classes never defined it.)  Nothing in the file is executed: structs/io.py."""
        from ..structs.io import load_model_bin
        ck = load_model_bin(path, map_location=map_location)
        model = cls(args=ck["args"], vocab=ck["vocab"])
        model.load_state_dict(ck["state_dict"], strict=True)
        return model

    def save(self, path):
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        if not os.path.exists(path + ".config"):
            with open(path + ".config", "w") as f:
                f.write("{}".format(self.base_information()))
        from ..structs.io import save_model_bin
        save_model_bin({"args": self.args, "vocab": self.vocab, "state_dict": self.state_dict()}, path)
