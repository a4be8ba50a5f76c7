"""Embeddings block (reference: dss_vae/networks/embeddings.py:13-87, used with
add_position_embedding=False by RNNBase).  state_dict key: ``embeddings.weight``."""
import torch.nn as nn

from .. import ops


class Embeddings(nn.Module):
    def __init__(self, num_embeddings, embedding_dim, dropout=0.0, add_position_embedding=False, padding_idx=None):
        super().__init__()
        if add_position_embedding:
            raise NotImplementedError("position embeddings belong to the transformer models (out of scope)")
        self.p = float(dropout)
        self.embeddings = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
        nn.init.uniform_(self.embeddings.weight, -0.1, 0.1)

    @property
    def weight(self):
        return self.embeddings.weight


def lookup(module, ids, seeds=None):
    """rows of a shared nn.Embedding (no dropout) or of an owned Embeddings (dropout in train mode)."""
    if isinstance(module, Embeddings):
        mask = None
        if module.training and module.p > 0.0:
            mask = ops.dropout_mask((ids.numel(), module.weight.size(1)), module.p, seeds.next(), ids.device)
        return ops.embed(module.weight, ids, mask)
    return ops.embed(module.weight, ids, None)
