"""RNNBase (reference: dss_vae/networks/rnn_base.py:32-61) and the GRU parameter holder whose
state_dict keys equal nn.GRU's (weight_ih_l{k}[_reverse], weight_hh_*, bias_ih_*, bias_hh_*;
gate order r,z,n), so reference checkpoints load unchanged.  The recurrence itself runs in the
persistent HIP kernel (ops.gru_layer); only 'gru' is implemented in this round."""
import math

import torch
import torch.nn as nn

from .embeddings import Embeddings


class SeedStream:
    """per-model counter for the counter-based device RNG (dropout masks, word dropout)."""

    def __init__(self, seed=0):
        self.state = int(seed) * 1000003 + 12345

    def next(self):
        self.state = (self.state * 6364136223846793005 + 1442695040888963407) & (2 ** 64 - 1)
        return self.state


GLOBAL_SEEDS = SeedStream(0)


class GRUParams(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers=1, bidirectional=False, dropout=0.0, **_):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.bidirectional, self.dropout = bidirectional, float(dropout)
        self.dirs = 2 if bidirectional else 1
        bound = 1.0 / math.sqrt(hidden_size)
        for layer in range(num_layers):
            in_dim = input_size if layer == 0 else hidden_size * self.dirs
            for suffix in self.suffixes():
                for name, shape in (("weight_ih", (3 * hidden_size, in_dim)), ("weight_hh", (3 * hidden_size, hidden_size)),
                                    ("bias_ih", (3 * hidden_size,)), ("bias_hh", (3 * hidden_size,))):
                    p = nn.Parameter(torch.empty(*shape).uniform_(-bound, bound))
                    self.register_parameter("%s_l%d%s" % (name, layer, suffix), p)

    def suffixes(self):
        return ("", "_reverse") if self.bidirectional else ("",)

    def layer_params(self, layer, suffix=""):
        return tuple(getattr(self, "%s_l%d%s" % (n, layer, suffix)) for n in
                     ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))


class RNNBase(nn.Module):
    SYM_MASK = "MASK"
    SYM_EOS = "EOS"

    def __init__(self, vocab_size, max_len, input_size, hidden_size, embed_droprate, rnn_droprate, n_layers, rnn_cell,
                 **kwargs):
        super().__init__()
        self.vocab_size, self.max_len = vocab_size, max_len
        self.input_size, self.hidden_size, self.n_layers = input_size, hidden_size, n_layers
        self.rnn_droprate = rnn_droprate if n_layers > 1 else 0.0
        self.embedding = Embeddings(vocab_size, input_size, dropout=embed_droprate, add_position_embedding=False)
        self.rnn_cell_select = rnn_cell
        if rnn_cell.lower() in ("lstm", "rnn"):
            raise NotImplementedError("the MI355X hot path implements rnn_cell='gru' (the only cell the VAE "
                                      "wrappers work with, networks/bridger.py:32); lstm/rnn are SURVEY 8f N4")
        if rnn_cell.lower() != "gru":
            raise ValueError("Unsupported RNN Cell: {0}".format(rnn_cell))
