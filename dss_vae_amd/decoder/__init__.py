from .beam_decoder import TopKDecoder
from .rnn_decoder import RNNDecoder, score_many
