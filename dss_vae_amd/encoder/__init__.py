from .rnn_encoder import RNNEncoder
