from .rnn_decoder import RNNDecoder, score_many
