"""RNNBase (reference: dss_vae/networks/rnn_base.py:32-61) and the GRU parameter holder whose
state_dict keys equal nn.GRU's (weight_ih_l{k}[_reverse], weight_hh_*, bias_ih_*, bias_hh_*;
gate order r,z,n), so reference checkpoints load unchanged.  The recurrence itself runs in the
persistent HIP kernel (ops.gru_layer); 'lstm' (nn.LSTM's keys, gate order i,f,g,o) runs stepwise (ops.lstm_layer)."""
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
    GATES = 3  # r, z, n  (LSTMParams: i, f, g, o)

    def __init__(self, input_size, hidden_size, num_layers=1, bidirectional=False, dropout=0.0, **_):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.bidirectional, self.dropout = bidirectional, float(dropout)
        self.dirs = 2 if bidirectional else 1
        bound = 1.0 / math.sqrt(hidden_size)
        for layer in range(num_layers):
            in_dim = input_size if layer == 0 else hidden_size * self.dirs
            for suffix in self.suffixes():
                g = self.GATES
                for name, shape in (("weight_ih", (g * hidden_size, in_dim)), ("weight_hh", (g * hidden_size, hidden_size)),
                                    ("bias_ih", (g * hidden_size,)), ("bias_hh", (g * hidden_size,))):
                    p = nn.Parameter(torch.empty(*shape).uniform_(-bound, bound))
                    self.register_parameter("%s_l%d%s" % (name, layer, suffix), p)

    def suffixes(self):
        return ("", "_reverse") if self.bidirectional else ("",)

    def layer_params(self, layer, suffix=""):
        return tuple(getattr(self, "%s_l%d%s" % (n, layer, suffix)) for n in
                     ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))


class LSTMParams(GRUParams):
    """parameter holder with nn.LSTM's state_dict keys and gate order (i, f, g, o)"""
    GATES = 4


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
        if rnn_cell.lower() == "rnn":
            raise NotImplementedError("rnn_cell='rnn' (nn.RNN) is not implemented: 'gru' (the only cell the VAE wrappers "
                                      "work with, networks/bridger.py:32) and 'lstm' are")
        if rnn_cell.lower() not in ("gru", "lstm"):
            raise ValueError("Unsupported RNN Cell: {0}".format(rnn_cell))

    @property
    def is_lstm(self):
        return self.rnn_cell_select.lower() == "lstm"

    def make_rnn(self, input_size, hidden_size, n_layers, bidirectional=False, dropout=0.0):
        cls = LSTMParams if self.is_lstm else GRUParams
        return cls(input_size, hidden_size, n_layers, bidirectional=bidirectional, dropout=dropout)
