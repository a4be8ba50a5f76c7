"""VanillaVAE (reference: dss_vae/vae/vanilla_vae.py:15-224): one bi-GRU encoder, one latent, one
decoder.  The train-mode tgt_wd branch implements the evident intent (teacher-forced decode on
word-dropped inputs, scored against clean targets): the reference calls a method that does not
exist there (vanilla_vae.py:112), so that branch has no oracle and is 'parity unpinned'."""
import torch
import torch.nn as nn

from .. import ops
from ..decoder.rnn_decoder import RNNDecoder
from ..encoder.rnn_encoder import RNNEncoder
from ..networks.bridger import MLPBridger
from ..utils.nn_funcs import wd_anneal_function
from .base_vae import BaseVAE, Linear


class VanillaVAE(BaseVAE):
    GRAPH_SAFE = True
    def __init__(self, args, vocab, word_embed=None):
        super().__init__(args, vocab, name="MySentVAE")
        src_embed = nn.Embedding(len(vocab.src), args.embed_size) if word_embed is None else word_embed
        if args.share_embed:
            tgt_embed = src_embed
            args.dec_embed_dim = args.enc_embed_dim
        else:
            tgt_embed = None
        self.latent_size = args.latent_size
        self.unk_rate = args.unk_rate
        self.step_unk_rate = 0.0
        self.hidden_size = args.enc_hidden_dim
        self.hidden_factor = (2 if args.bidirectional else 1) * args.enc_num_layers
        args.use_attention = False
        self.enc_dim = args.enc_hidden_dim * (2 if args.bidirectional else 1)
        self.dec_hidden = self.enc_dim if args.mapper_type == "link" else args.dec_hidden_dim
        self.encoder = RNNEncoder(
            vocab_size=len(vocab.src), max_len=args.src_max_time_step, input_size=args.enc_embed_dim,
            hidden_size=args.enc_hidden_dim, embed_droprate=args.enc_ed, rnn_droprate=args.enc_rd,
            n_layers=args.enc_num_layers, bidirectional=args.bidirectional, rnn_cell=args.rnn_type,
            variable_lengths=True, embedding=src_embed)
        self.bridger = MLPBridger(args.rnn_type, args.mapper_type, self.enc_dim, args.enc_num_layers, self.dec_hidden,
                                  args.dec_num_layers)
        self.decoder = RNNDecoder(
            vocab=len(vocab.src), max_len=args.src_max_time_step, input_size=args.dec_embed_dim,
            hidden_size=self.dec_hidden, embed_droprate=args.dec_ed, rnn_droprate=args.dec_rd,
            n_layers=args.dec_num_layers, rnn_cell=args.rnn_type, use_attention=False, embedding=tgt_embed,
            eos_id=vocab.src.eos_id, sos_id=vocab.src.sos_id)
        self.hidden2mean = Linear(args.hidden_size * self.hidden_factor, args.latent_size)
        self.hidden2logv = Linear(args.hidden_size * self.hidden_factor, args.latent_size)
        self.latent2hidden = Linear(args.latent_size, args.hidden_size * self.hidden_factor)

    def encode(self, input_var, length):
        if self.training and self.args.src_wd:
            input_var = self._word_dropout(input_var, "rand_src")
        return self.encoder.forward(input_var, length)

    def decode(self, inputs, encoder_outputs, encoder_hidden):
        return self.decoder.forward(inputs=inputs, encoder_outputs=encoder_outputs, encoder_hidden=encoder_hidden)

    def forward(self, examples):
        batch = self._batch(examples, with_tgt=False)
        ret = self.latent_for_init(self.hidden_to_latent(self.encode_to_hidden(batch), is_sampling=self.training))
        h0 = self.bridger.forward(ret["decode_init"])
        tgt_var = batch.tgt_var
        inputs = tgt_var
        if self.training and self.args.tgt_wd > 0.0:
            inputs = self._word_dropout(tgt_var, "rand_tgt")
        nll = -torch.sum(self.decoder.score_inputs(inputs, tgt_var, h0, uniform=True))
        return {"mean": ret["mean"], "logv": ret["logv"], "z": ret["latent"], "kl": ret["kl"], "nll_loss": nll,
                "batch_size": len(batch)}

    def get_loss(self, examples, step, **kwargs):
        a = self.args
        sc = self.step_scalars(step)
        self.step_unk_rate = ops.dyn_value("unk_rate", sc["unk_rate"])
        explore = self.forward(examples)
        B = explore["batch_size"]
        kl_weight = ops.dyn_value("kl_weight", sc["kl_weight"])
        nll_loss, kl_loss = explore["nll_loss"] / B, explore["kl"] / B
        kl_item = kl_loss * ops.dyn_tensor(kl_weight)
        return {"KL Loss": kl_loss, "NLL Loss": nll_loss, "KL Weight": float(kl_weight), "Model Score": kl_loss + nll_loss,
                "ELBO": kl_item + nll_loss, "Loss": kl_item + nll_loss, "KL Item": kl_item}

    def sample_latent(self, batch_size):
        return {"latent": self._randn("eps", [batch_size, self.latent_size])}

    def latent_for_init(self, ret):
        z = ret["latent"]
        hidden = self.latent2hidden(z)
        if self.hidden_factor > 1:
            hidden = hidden.view(z.size(0), self.hidden_factor, self.hidden_size).permute(1, 0, 2)
        else:
            hidden = hidden.unsqueeze(0)
        ret["decode_init"] = hidden
        return ret

    def hidden_to_latent(self, ret, is_sampling=True):
        hidden = ret["hidden"]
        B = hidden.size(1)
        hidden = hidden.permute(1, 0, 2).contiguous().view(B, self.hidden_size * self.hidden_factor)
        mean, logv = self.hidden2mean(hidden), self.hidden2logv(hidden)
        eps = self._randn("eps", [B, self.latent_size]) if is_sampling else None
        ret["latent"], ret["kl"] = ops.reparam_kl(mean, logv, eps)
        ret["mean"], ret["logv"] = mean, logv
        return ret

    def conditional_generating(self, condition="sem", examples=None):
        if not isinstance(examples, list):
            examples = [examples]
        if condition is None:
            return {"res": self.unsupervised_generating(sample_num=len(examples))}
        if condition.startswith("sem"):
            ret = self.latent_for_init(self.hidden_to_latent(self.encode_to_hidden(examples), is_sampling=True))
            return {"res": self.decode_to_sentence(ret=ret)}

    def eval_adv(self, sem_in, syn_ref):
        a = self.hidden_to_latent(self.encode_to_hidden(sem_in), is_sampling=self.training)
        b = self.hidden_to_latent(self.encode_to_hidden(syn_ref, need_sort=True), is_sampling=self.training)
        ret = self.latent_for_init({"latent": (a["latent"] + b["latent"]) * 0.5})
        ret["res"] = self.decode_to_sentence(ret=ret)
        return ret
