"""RNNEncoder (reference: dss_vae/encoder/rnn_encoder.py:7-55): embedding -> packed bi-GRU ->
(outputs, h_n).  Here: embedding rows gathered once, per-direction input projections as MFMA
GEMMs over all timesteps, and all directions of a layer (of all encoders passed to
``encode_many``) as instances of ONE persistent GRU launch with packed-sequence semantics."""
import torch

from .. import ops
from ..networks.embeddings import lookup
from ..networks.rnn_base import GLOBAL_SEEDS, RNNBase


class RNNEncoder(RNNBase):
    def __init__(self, vocab_size, max_len, input_size, hidden_size, embed_droprate=0, rnn_droprate=0, n_layers=1,
                 bidirectional=False, rnn_cell="gru", variable_lengths=False, embedding=None, update_embedding=True,
                 **kwargs):
        super().__init__(vocab_size, max_len, input_size, hidden_size, embed_droprate, rnn_droprate, n_layers,
                         rnn_cell)
        self.variable_lengths = variable_lengths
        if embedding is not None:
            self.embedding = embedding
        self.embedding.weight.requires_grad = update_embedding
        self.rnn = self.make_rnn(input_size, hidden_size, n_layers, bidirectional=bidirectional, dropout=rnn_droprate)
        self.bidirectional = bidirectional

    @property
    def out_dim(self):
        return self.hidden_size * 2 if self.bidirectional else self.hidden_size

    def forward(self, input_var, input_lengths=None):
        if self.is_lstm:
            return self._forward_lstm(input_var, input_lengths)
        return encode_many([self], input_var, input_lengths)[0]

    def _forward_lstm(self, input_var, input_lengths=None):
        """nn.LSTM(batch_first, bidirectional) over packed sequences (encoder/rnn_encoder.py:49-55 with rnn_cell='lstm')
        -> (output [B,L,dirs*H], (h_n, c_n) each [layers*dirs,B,H])"""
        dev = input_var.device
        lens = None
        if self.variable_lengths and input_lengths is not None:
            lens = getattr(input_lengths, "dev", None)
            if lens is None:
                lens = torch.as_tensor(list(input_lengths), dtype=torch.int32).to(dev, non_blocking=True)
            input_var = input_var[:, :int(max(input_lengths))].contiguous()
        x = lookup(self.embedding, input_var, GLOBAL_SEEDS)
        hn, cn = [], []
        for layer in range(self.n_layers):
            if layer > 0 and self.training and self.rnn_droprate > 0.0:
                x = ops.dropout(x, self.rnn_droprate, GLOBAL_SEEDS.next())
            params = [self.rnn.layer_params(layer, sfx) for sfx in self.rnn.suffixes()]
            # b_ih's gradient comes out of the recurrence's backward (as for the GRU)
            xps = ops.linear_many([(x, wih, bih.detach()) for wih, _, bih, _ in params])
            outs = []
            for di, ((wih, whh, bih, bhh), xp) in enumerate(zip(params, xps)):
                o, h, c = ops.lstm_layer(xp, whh, bhh, bih=bih, lengths=lens, reverse=di == 1)
                outs.append(o)
                hn.append(h)
                cn.append(c)
            x = outs[0] if len(outs) == 1 else torch.cat(outs, dim=-1)
        return x, (torch.stack(hn, 0), torch.stack(cn, 0))


def encode_many(encoders, input_var, input_lengths=None):
    """Run several encoders over the same ids; layer k of every encoder/direction shares a launch.

    -> list of (output [B,L,dirs*H], h_n [layers*dirs,B,H]) in nn.GRU's h_n order."""
    dev = input_var.device
    B, Lmax = input_var.shape
    lens = None
    if input_lengths is not None and all(e.variable_lengths for e in encoders):
        lens = getattr(input_lengths, "dev", None)  # vae/base_vae.py: Lengths (the batch's own device copy)
        if lens is None:
            lens = torch.as_tensor(list(input_lengths), dtype=torch.int32).to(dev, non_blocking=True)
        Lmax = int(max(input_lengths))
        input_var = input_var[:, :Lmax].contiguous()
    # one gather per distinct embedding table
    emb_cache, xs = {}, []
    for e in encoders:
        key = id(e.embedding)
        if key not in emb_cache:
            emb_cache[key] = lookup(e.embedding, input_var, GLOBAL_SEEDS)
        xs.append(emb_cache[key])
    n_layers = encoders[0].n_layers
    assert all(e.n_layers == n_layers for e in encoders)
    hns = [[] for _ in encoders]
    for layer in range(n_layers):
        insts, proj = [], []
        for ei, e in enumerate(encoders):
            x = xs[ei]
            if layer > 0 and e.training and e.rnn_droprate > 0.0:
                x = ops.dropout(x, e.rnn_droprate, GLOBAL_SEEDS.next())
            for di, suffix in enumerate(e.rnn.suffixes()):
                wih, whh, bih, bhh = e.rnn.layer_params(layer, suffix)
                proj.append((x, wih, bih.detach()))  # b_ih's gradient comes out of the recurrence's backward
                insts.append(dict(whh=whh, bhh=bhh, bih=bih, h0=None, reverse=di == 1, group=ei,
                                  col=di * e.hidden_size))
        # the input projections of every direction / encoder of the layer share grouped MFMA launches
        for inst, xp in zip(insts, ops.linear_many(proj)):
            inst["xproj"] = xp
        outs, hl = ops.gru_layer(insts, lengths=lens)
        k = 0
        for ei, e in enumerate(encoders):
            xs[ei] = outs[ei]
            for _ in e.rnn.suffixes():
                hns[ei].append(hl[k])
                k += 1
    return [(xs[ei], torch.stack(hns[ei], 0)) for ei in range(len(encoders))]
