"""SyntaxVAE and its task heads (reference: dss_vae/vae/syntax_vae.py:50-602) on the HIP ops.

Reference quirks that are part of the contract and kept: the tree-decoder head returns a
per-sentence [B] vector (syntax_vae.py:547-556) so 'MUL Loss' / 'ADV Loss' / 'Loss' are [B]
vectors; the adversaries run under no_grad in train mode (:322-324)."""
import os

import torch
import torch.nn as nn

from .. import ops
from ..decoder.rnn_decoder import RNNDecoder, score_many
from ..encoder.rnn_encoder import RNNEncoder, encode_many
from ..networks.bridger import MLPBridger
from ..networks.rnn_base import GLOBAL_SEEDS
from ..utils.nn_funcs import rows_to_sentences, wd_anneal_function
from .base_vae import BaseVAE, Linear


class BridgeRNN(nn.Module):
    """bridger + sequence decoder head; forward -> per-sentence NLL [B]."""

    def __init__(self, args, vocab, enc_hidden_dim, dec_hidden_dim, embed, mode="src"):
        super().__init__()
        self.bridger = MLPBridger(args.rnn_type, args.mapper_type, enc_hidden_dim, args.enc_num_layers,
                                  dec_hidden_dim, args.dec_num_layers)
        ent = vocab.src if mode == "src" else vocab.tgt
        self.decoder = RNNDecoder(
            vocab=len(ent), max_len=args.src_max_time_step if mode == "src" else args.tgt_max_time_step,
            input_size=args.dec_embed_dim, hidden_size=dec_hidden_dim, embed_droprate=args.dec_ed,
            rnn_droprate=args.dec_rd, n_layers=args.dec_num_layers, rnn_cell=args.rnn_type,
            use_attention=args.use_attention, embedding=embed, eos_id=ent.eos_id, sos_id=ent.sos_id)
        self.pad_id = ent.pad_id

    def job(self, hidden, tgt_var):
        """scoring job; `uniform`: the models only ever sum per-sentence NLLs with one common weight"""
        return dict(dec=self.decoder, inputs=tgt_var, targets=tgt_var, h0=self.bridger.forward(input_tensor=hidden),
                    uniform=True)

    def forward(self, hidden, tgt_var):
        return -score_many([self.job(hidden, tgt_var)])[0]

    def predict(self, hidden):
        return self.decoder.greedy_tokens(self.bridger.forward(input_tensor=hidden))[0]


class BridgeMLP(nn.Module):
    """bridger + bag-of-words scorer (Dropout -> Linear -> LogSoftmax -> NLL over all non-pad
    tokens incl. <s>, </s>), fused: one LSE per sentence, no [B*T,V] expansion."""

    def __init__(self, args, vocab, enc_dim, dec_hidden):
        super().__init__()
        self.bridger = MLPBridger(args.rnn_type, args.mapper_type, enc_dim, args.enc_num_layers, dec_hidden,
                                  args.dec_num_layers)
        self.scorer = nn.Sequential(nn.Dropout(args.dec_rd),
                                    nn.Linear(dec_hidden * args.dec_num_layers, len(vocab.src), bias=True),
                                    nn.LogSoftmax(dim=-1))
        self.pad_id = vocab.src.pad_id
        self.p = float(args.dec_rd)

    def per_sentence(self, hidden, tgt_var, uniform=True):
        B = tgt_var.size(0)
        feat = self.bridger.forward(input_tensor=hidden).permute(1, 0, 2).contiguous().view(B, -1)
        if self.training and self.p > 0.0:
            feat = ops.dropout(feat, self.p, GLOBAL_SEEDS.next())
        lin = self.scorer[1]
        return ops.linear_ce(feat, None, lin.weight, lin.bias, tgt_var, self.pad_id, uniform_upstream=uniform)

    def forward(self, hidden, tgt_var):
        return self.per_sentence(hidden, tgt_var).sum()


class TwoEncoderVAE(BaseVAE):
    """shared body of SyntaxVAE / EnhanceSyntaxVAE: two half-width bi-GRU encoders, semantic +
    syntactic latents, main sentence decoder."""

    ENC_NAMES = ("word_encoder", "syntax_encoder")

    def _build_common(self, args, vocab, src_embed=None, tgt_embed=None):
        self.src_embed = nn.Embedding(len(vocab.src), args.embed_size) if src_embed is None else src_embed
        self.tgt_embed = nn.Embedding(len(vocab.tgt), args.embed_size) if tgt_embed is None else tgt_embed
        self.latent_size = int(args.latent_size)
        self.rnn_type = args.rnn_type
        self.unk_rate = args.unk_rate
        self.step_unk_rate = 0.0
        self.direction_num = 2 if args.bidirectional else 1
        self.enc_hidden_dim = args.enc_hidden_dim
        self.enc_layer_dim = args.enc_hidden_dim * self.direction_num
        self.enc_hidden_factor = self.direction_num * args.enc_num_layers
        self.dec_hidden_factor = args.dec_num_layers
        args.use_attention = False
        self.dec_layer_dim = self.enc_layer_dim if args.mapper_type == "link" else args.dec_hidden_dim
        var_dim = int(self.enc_hidden_dim * self.enc_hidden_factor / 2)
        self.task_enc_dim, self.task_dec_dim = int(self.enc_layer_dim / 2), int(self.dec_layer_dim / 2)
        for name in self.ENC_NAMES:
            setattr(self, name, RNNEncoder(
                vocab_size=len(vocab.src), max_len=args.src_max_time_step, input_size=args.enc_embed_dim,
                hidden_size=int(self.enc_hidden_dim / 2), embed_droprate=args.enc_ed, rnn_droprate=args.enc_rd,
                n_layers=args.enc_num_layers, bidirectional=args.bidirectional, rnn_cell=args.rnn_type,
                variable_lengths=True, embedding=self.src_embed))
        pack = BridgeRNN(args, vocab, self.enc_layer_dim, self.dec_layer_dim,
                         self.src_embed if args.share_embed else None, "src")
        self.bridger, self.decoder = pack.bridger, pack.decoder
        self.syn_mean, self.syn_logv = Linear(var_dim, self.latent_size), Linear(var_dim, self.latent_size)
        self.syn_to_h = Linear(self.latent_size, var_dim)
        self.sem_mean, self.sem_logv = Linear(var_dim, self.latent_size), Linear(var_dim, self.latent_size)
        self.sem_to_h = Linear(self.latent_size, var_dim)

    def decode(self, inputs, encoder_outputs, encoder_hidden):
        return self.decoder.forward(inputs=inputs, encoder_outputs=encoder_outputs, encoder_hidden=encoder_hidden)

    def encode(self, input_var, length):
        if self.training and self.args.src_wd > 0.0:
            input_var = self._word_dropout(input_var, "rand_src")
        (_, ha), (_, hb) = encode_many([getattr(self, n) for n in self.ENC_NAMES], input_var, length)
        B = ha.size(1)
        return None, torch.cat([ha.permute(1, 0, 2).contiguous().view(B, -1),
                                hb.permute(1, 0, 2).contiguous().view(B, -1)], dim=-1)

    def hidden_to_latent(self, ret, is_sampling=True):
        hid = ret["hidden"]
        B = hid.size(0)
        halves = hid.contiguous().view(B, 2, -1)
        sem_h, syn_h = halves[:, 0, :], halves[:, 1, :]
        # four [B, var_dim] x [var_dim, Z] products: one grouped launch (alone each is a chain of k-tiles on 4 CUs)
        ret["syn_mean"], ret["syn_logv"], ret["sem_mean"], ret["sem_logv"] = ops.linear_many(
            [(x, m.weight, m.bias) for x, m in ((syn_h, self.syn_mean), (syn_h, self.syn_logv),
                                                (sem_h, self.sem_mean), (sem_h, self.sem_logv))])
        e_syn = self._randn("eps_syn", [B, self.latent_size]) if is_sampling else None  # syntax eps first
        e_sem = self._randn("eps_sem", [B, self.latent_size]) if is_sampling else None
        ret["syn_z"], ret["syn_kl"] = ops.reparam_kl(ret["syn_mean"], ret["syn_logv"], e_syn)
        ret["sem_z"], ret["sem_kl"] = ops.reparam_kl(ret["sem_mean"], ret["sem_logv"], e_sem)
        return ret

    def sample_latent(self, batch_size):
        return {"syn_z": self._randn("eps_syn", [batch_size, self.latent_size]),
                "sem_z": self._randn("eps_sem", [batch_size, self.latent_size])}

    def latent_for_init(self, ret):
        B = ret["sem_z"].size(0)

        def lay(x):
            return x.view(B, self.enc_hidden_factor, int(self.enc_hidden_dim / 2)).permute(1, 0, 2)

        syn_h, sem_h = ops.linear_many([(ret["syn_z"], self.syn_to_h.weight, self.syn_to_h.bias),
                                        (ret["sem_z"], self.sem_to_h.weight, self.sem_to_h.bias)])
        ret["syn_hidden"], ret["sem_hidden"] = lay(syn_h), lay(sem_h)
        ret["decode_init"] = torch.cat([ret["syn_hidden"], ret["sem_hidden"]], dim=-1)
        return ret

    def _front(self, examples):
        batch = self._batch(examples)
        ret = self.encode_to_hidden(batch)
        ret = self.hidden_to_latent(ret=ret, is_sampling=self.training)
        ret = self.latent_for_init(ret=ret)
        return batch, ret

    def _main_job(self, ret, tgt_var):
        inputs = tgt_var
        if self.training and self.args.tgt_wd:
            inputs = self._word_dropout(tgt_var, "rand_tgt")
        return dict(dec=self.decoder, inputs=inputs, targets=tgt_var,
                    h0=self.bridger.forward(ret["decode_init"]), uniform=True)

    def _kl_terms(self, ret, step):
        a = self.args
        w = self.get_kl_weight(step) * a.kl_factor
        kl = (a.sem_weight * ret["sem_kl"] + a.syn_weight * ret["syn_kl"]) / (a.sem_weight + a.syn_weight)
        return ret["sem_kl"], ret["syn_kl"], kl, w

    def evaluate_(self, examples, beam_size=5):
        batch, ret = self._front(examples)
        ret["res"] = self.decode_to_sentence(ret=ret)
        return ret

    def predict_syntax(self, hidden, predictor):
        return rows_to_sentences(predictor.predict(hidden).t().cpu().numpy(), self.vocab.tgt)

    def conditional_generating(self, condition="sem", examples=None):
        ret = self.hidden_to_latent(self.encode_to_hidden(examples), is_sampling=True)
        if condition.startswith("sem"):
            ret["sem_z"] = ret["sem_mean"]
        else:
            ret["syn_z"] = ret["syn_mean"]
        if condition == "sem-only":
            ret["syn_z"] = self._randn("eps_syn_prior", [ret["batch_size"], self.latent_size])
        ret = self.latent_for_init(ret=ret)
        ret["res"] = self.decode_to_sentence(ret=ret)
        return ret

    def _transfer(self, sem_in, syn_ref, predictor):
        if not self.training and isinstance(sem_in, list) and isinstance(syn_ref, list) and \
                os.environ.get("DSSVAE_TRANSFER_MERGE", "1") != "0":
            # evaluation: nothing is sampled and every row is encoded on its own (packed lengths), so the two inputs
            # go through the encoders, the latent heads and the init layers as ONE batch -- the front of a pair batch
            # is ~40 small launches per input that the host, not the GPU, takes its time over
            n1 = len(sem_in)
            both = self.latent_for_init(self.hidden_to_latent(self.encode_to_hidden(sem_in + syn_ref), is_sampling=False))
            ret = self.latent_for_init(dict(sem_z=both["sem_z"][:n1], syn_z=both["syn_z"][n1:]))
            h_ori, h_ref = both["syn_hidden"][:, :n1].contiguous(), both["syn_hidden"][:, n1:].contiguous()
        else:
            sem_ret = self.hidden_to_latent(self.encode_to_hidden(sem_in), is_sampling=self.training)
            syn_ret = self.hidden_to_latent(self.encode_to_hidden(syn_ref, need_sort=True), is_sampling=self.training)
            sem_ret, syn_ret = self.latent_for_init(ret=sem_ret), self.latent_for_init(ret=syn_ret)
            ret = self.latent_for_init(dict(sem_z=sem_ret["sem_z"], syn_z=syn_ret["syn_z"]))
            h_ori, h_ref = sem_ret["syn_hidden"], syn_ret["syn_hidden"]
        # the sentence tokens leave for pinned host memory behind their decode; they are turned into words
        # below while the GPU runs the tree decodes
        n = h_ori.size(1)
        # The tree decodes (100 steps of three launches that fill a fraction of the chip each) run on a side stream
        # BESIDE the sentence decode instead of behind it.
        cur = torch.cuda.current_stream()
        side = _side_stream(h_ori.device) if os.environ.get("DSSVAE_TRANSFER_STREAMS", "1") != "0" else cur
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if h_ref.size(1) == n and 2 * n <= 128:
                # the two tree decodes are launch-bound and independent row by row: one decode of 2B rows
                # (still within the decode-batch kernels' 128 rows, so every row sees the same arithmetic)
                tree_tokens = [predictor.predict(torch.cat((h_ori, h_ref), dim=1))]
            else:
                tree_tokens = [predictor.predict(h_ori), predictor.predict(h_ref)]
        sent_dev = self.decode_tokens(ret).t()
        sent_tokens = torch.empty(sent_dev.shape, dtype=sent_dev.dtype, pin_memory=True)
        sent_tokens.copy_(sent_dev, non_blocking=True)
        sent_done = torch.cuda.Event()
        sent_done.record()
        cur.wait_stream(side)
        for t in tree_tokens:
            t.record_stream(cur)
        sent_done.synchronize()
        ret["res"] = self._words(sent_tokens, self.vocab.src)
        trees = [t for tok in tree_tokens for t in self._words(tok.t(), self.vocab.tgt)]
        ret["ori syn"], ret["ref syn"] = trees[:n], trees[n:]
        return ret

    @staticmethod
    def _words(tokens, vocab):
        """token matrix [B, max_len] -> [[[words], [len]], ...] cut at the first </s>"""
        return rows_to_sentences(tokens.cpu().numpy(), vocab)


_SIDE = {}


def _side_stream(dev):
    key = torch.device(dev).index
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


class SyntaxVAE(TwoEncoderVAE):
    GRAPH_SAFE = True
    ENC_NAMES = ("word_encoder", "syntax_encoder")

    def __init__(self, args, vocab, src_embed=None, tgt_embed=None):
        super().__init__(args, vocab, name="Syntax VAE")
        self._build_common(args, vocab, src_embed, tgt_embed)
        self.pad_idx = vocab.src.sos_id
        te, td = self.task_enc_dim, self.task_dec_dim
        share = args.share_embed
        self.sup_syn = BridgeRNN(args, vocab, te, td, tgt_embed, "tgt")
        self.sup_sem = BridgeMLP(args, vocab, te, td)
        self.syn_adv = BridgeRNN(args, vocab, te, td, self.tgt_embed if share else None, "tgt")
        self.syn_infer = BridgeRNN(args, vocab, te, td, self.src_embed if share else None, "src")
        self.sem_adv = BridgeMLP(args, vocab, te, td)
        self.sem_infer = BridgeRNN(args, vocab, te, td, self.src_embed if share else None, "src")

    def _dis_terms(self, syn_h, sem_h, syn_tgt, sem_tgt, extra_jobs=(), no_grad=False):
        """the four adversary heads; sequence heads share persistent launches (plus extra_jobs,
        whose log-probs are returned last).  no_grad: the adversaries are only evaluated (the
        reference runs them under torch.no_grad() in train mode, syntax_vae.py:322-324)."""
        a = self.args
        jobs, names = [], []
        if a.adv_syn > 0.0:
            jobs.append(self.syn_adv.job(sem_h, syn_tgt)); names.append("adv_syn_sup")
        if a.infer_weight > 0.0:
            if a.inf_syn > 0.0:
                jobs.append(self.syn_infer.job(syn_h, sem_tgt)); names.append("adv_syn_inf")
            if a.inf_sem > 0.0:
                jobs.append(self.sem_infer.job(sem_h, sem_tgt)); names.append("adv_sem_inf")
        for j in jobs:
            j["no_grad"] = no_grad
        res = score_many(jobs + list(extra_jobs)) if jobs or extra_jobs else []
        with torch.set_grad_enabled(torch.is_grad_enabled() and not no_grad):
            d = {"adv_syn_sup": 0.0, "adv_sem_sup": self.sem_adv(syn_h, sem_tgt) if a.adv_sem > 0.0 else 0.0}
        if a.infer_weight > 0.0:
            d.update(adv_syn_inf=0.0, adv_sem_inf=0.0)
        for n, r in zip(names, res):
            d[n] = self._head_loss(-r)
        return d, res[len(names):]

    @staticmethod
    def _head_loss(per_sentence):
        """SyntaxVAE keeps the tree / sentence heads' per-sentence [B] vector (syntax_vae.py:547-556)"""
        return per_sentence

    def _adv_weights(self, d):
        a = self.args
        if a.infer_weight > 0.0:
            return (a.adv_syn * d["adv_syn_sup"] + a.infer_weight * a.inf_sem * d["adv_sem_inf"],
                    a.adv_sem * d["adv_sem_sup"] + a.infer_weight * a.inf_syn * d["adv_syn_inf"])
        return a.adv_syn * d["adv_syn_sup"], a.adv_sem * d["adv_sem_sup"]

    # ---- the reference's helper surface (syntax_vae.py:321-380): same names, arguments and return forms; forward()
    # below does not go through them (it batches the heads of a pass into shared launches), they are for callers
    # that want one group of terms
    def _dis_loss(self, syntax_hidden, semantic_hidden, syn_tgt, sem_tgt):
        return self._dis_terms(syntax_hidden, semantic_hidden, syn_tgt, sem_tgt)[0]

    def get_dis_loss(self, syntax_hidden, semantic_hidden, syn_tgt, sem_tgt):
        d = self._dis_loss(syntax_hidden.detach(), semantic_hidden.detach(), syn_tgt, sem_tgt)
        if self.args.infer_weight > 0.0:
            return d["adv_syn_sup"] + d["adv_sem_inf"], d["adv_sem_sup"] + d["adv_syn_inf"]
        return d["adv_syn_sup"], d["adv_sem_sup"]

    def get_adv_loss(self, syntax_hidden, semantic_hidden, syn_tgt, sem_tgt):
        d, _ = self._dis_terms(syntax_hidden, semantic_hidden, syn_tgt, sem_tgt, no_grad=self.training)
        return self._adv_weights(d)

    def get_mul_loss(self, syntax_hidden, semantic_hidden, syn_tgt, sem_tgt):
        syn_loss = self._head_loss(self.sup_syn.forward(hidden=syntax_hidden, tgt_var=syn_tgt))
        sem_loss = self.sup_sem.forward(hidden=semantic_hidden, tgt_var=sem_tgt)
        return self.args.mul_syn * syn_loss, self.args.mul_sem * sem_loss

    def get_gpu(self):  # syntax_vae.py:187-194 moves every sub-module to cuda:0
        return self.cuda()

    def extract_variable(self, examples):  # a stub in the reference too (syntax_vae.py:463-464)
        pass

    def forward(self, examples, is_dis=False):
        batch, ret = self._front(examples)
        tgt_var, syn_var = batch.tgt_var, batch.syn_var
        syn_h, sem_h = ret["syn_hidden"], ret["sem_hidden"]
        if is_dis:
            d, _ = self._dis_terms(syn_h.detach(), sem_h.detach(), syn_var, tgt_var)
            if self.args.infer_weight > 0.0:
                ret["dis syn"] = d["adv_syn_sup"] + d["adv_sem_inf"]
                ret["dis sem"] = d["adv_sem_sup"] + d["adv_syn_inf"]
            else:
                ret["dis syn"], ret["dis sem"] = d["adv_syn_sup"], d["adv_sem_sup"]
            return ret
        main_jobs = [self._main_job(ret, tgt_var), self.sup_syn.job(syn_h, syn_var)]
        d, (rec, tree) = self._dis_terms(syn_h, sem_h, syn_var, tgt_var, extra_jobs=main_jobs,
                                         no_grad=self.training)
        adv_syn, adv_sem = self._adv_weights(d)
        mul_syn = self.args.mul_syn * self._head_loss(-tree)
        mul_sem = self.args.mul_sem * self.sup_sem(sem_h, tgt_var)
        ret["adv"], ret["mul"] = adv_syn + adv_sem, mul_syn + mul_sem
        ret["nll_loss"] = -torch.sum(rec)
        ret["sem_loss"], ret["syn_loss"], ret["batch_size"] = mul_sem, mul_syn, len(batch)
        return ret

    def get_loss(self, examples, train_iter, is_dis=False, **kwargs):
        a = self.args
        sc = self.step_scalars(train_iter)
        self.step_unk_rate = ops.dyn_value("unk_rate", sc["unk_rate"])
        explore = self.forward(examples, is_dis)
        if is_dis:
            return explore
        sem_kl, syn_kl, kl_loss, _ = self._kl_terms(explore, train_iter)
        kl_weight = ops.dyn_value("kl_weight", sc["kl_weight"])
        B = explore["batch_size"]
        kl_loss = kl_loss / B
        mul_loss, adv_loss, nll_loss = explore["mul"] / B, explore["adv"] / B, explore["nll_loss"] / B
        kl_item = kl_loss * ops.dyn_tensor(kl_weight)
        return {"KL Loss": kl_loss, "NLL Loss": nll_loss, "MUL Loss": mul_loss, "ADV Loss": adv_loss,
                "KL Weight": float(kl_weight), "KL Item": kl_item, "Model Score": kl_loss + nll_loss,
                "ELBO": kl_item + nll_loss, "Loss": kl_item + nll_loss + mul_loss - adv_loss,
                "SYN KL Loss": syn_kl / B, "SEM KL Loss": sem_kl / B}

    def eval_syntax(self, examples):
        ret = self.latent_for_init(self.hidden_to_latent(self.encode_to_hidden(examples, need_sort=True), False))
        return self.predict_syntax(hidden=ret["syn_hidden"], predictor=self.sup_syn)

    def eval_adv(self, sem_in, syn_ref):
        return self._transfer(sem_in, syn_ref, self.sup_syn)
