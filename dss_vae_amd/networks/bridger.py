"""MLPBridger (reference: dss_vae/networks/bridger.py:12-37): encoder-side state [L,B,D] ->
decoder initial state [dec_layers,B,dec_dim]; 'link' is a pure re-indexing, 'mapping' a Linear."""
import torch.nn as nn

from .. import ops


class _Mapper(nn.Linear):
    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class MLPBridger(nn.Module):
    def __init__(self, rnn_type, mapper_type, encoder_dim, encoder_layer, decoder_dim, decoder_layer):
        super().__init__()
        self.rnn_type = rnn_type
        self.parameter_type = mapper_type
        self.input_dim = encoder_dim * encoder_layer
        self.output_dim = decoder_dim * decoder_layer
        self.decoder_dim, self.decoder_layer = decoder_dim, decoder_layer
        if mapper_type == "mapping":
            self.mapper = _Mapper(self.input_dim, self.output_dim)
        elif self.input_dim == self.output_dim:
            self.mapper = None
        else:
            raise ValueError("bridge dim is not right")

    def forward(self, input_tensor):
        B = input_tensor.size(1)
        flat = input_tensor.permute(1, 0, 2).contiguous().view(B, -1)
        assert flat.size(1) == self.input_dim, "bridge dim is not right"
        if self.mapper is not None:
            flat = self.mapper(flat)
        return flat.view(-1, self.decoder_layer, self.decoder_dim).permute(1, 0, 2).contiguous()
