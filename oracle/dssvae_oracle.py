"""Plain-PyTorch CPU restatement of the DSS-VAE hot path (oracle; test infrastructure only).

Every function cites the reference file:line (paths under /root/reference) whose
behaviour it restates.  Module attribute names are kept identical to the
reference so that a reference ``state_dict`` loads with ``strict=True``.

Third-party arithmetic: the reference calls ``torch.nn.GRU`` (unpinned torch,
"PyTorch 0.4+", README.md:4).  ``gru_loop`` below restates its published cell
equations explicitly; ``tests/test_oracle_golden.py`` checks it against the
golden vectors that were produced by the reference's ``nn.GRU`` under torch
2.10.0 CPU.  The model classes use ``nn.GRU`` themselves (exactly the calls the
reference makes) so that the CPU baseline times the same work.
"""
from __future__ import annotations

import math
from argparse import Namespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = [
    "OVocabEntry", "OVocab", "OExample", "tensorise", "anneal_weight", "word_dropout",
    "ids_to_words", "gru_loop", "bridge_state", "bow_nll", "OracleEncoder", "OracleDecoder",
    "OracleVanillaVAE", "OracleSyntaxVAE", "OracleEnhanceSyntaxVAE", "OracleDisentangleVAE", "ORACLE_MODELS",
    "oracle_train_step", "init_uniform_",
]


# --------------------------------------------------------------------------------------
# host-side structures (dss_vae/structs/vocab.py:10-93, dss_vae/structs/dataset.py:87-122)
# --------------------------------------------------------------------------------------
class OVocabEntry:
    """ids: <pad>=0 <s>=1 </s>=2 <unk>=3 (structs/vocab.py:14-22); OOV -> unk (vocab.py:26-27)."""

    def __init__(self, words=()):
        self.word2id = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3}
        self.pad_id, self.sos_id, self.eos_id, self.unk_id = 0, 1, 2, 3
        self.id2word = {i: w for w, i in self.word2id.items()}
        for w in words:
            self.add(w)

    def add(self, w):
        if w not in self.word2id:
            i = len(self.word2id)
            self.word2id[w] = i
            self.id2word[i] = w
        return self.word2id[w]

    def __getitem__(self, w):
        return self.word2id.get(w, self.unk_id)

    def __len__(self):
        return len(self.word2id)


class OVocab:
    def __init__(self, src, tgt):
        self.src, self.tgt = src, tgt
        self.entries = ["src", "tgt"]


class OExample:
    def __init__(self, src, tgt=None):
        self.src, self.tgt = src, tgt

    def __len__(self):
        return len(self.src)


def tensorise(seqs, entry, boundary=False):
    """list[list[str]] -> LongTensor [B, Lmax], right-padded with <pad>.

    Restates to_input_variable(batch_first=True) / padding_input / word2id
    (utils/nn_funcs.py:143-174, :122-129, :43-47)."""
    if not isinstance(seqs[0], list):
        seqs = [seqs]
    if boundary:
        seqs = [["<s>"] + list(s) + ["</s>"] for s in seqs]
    width = max(len(s) for s in seqs)
    rows = [[entry[w] for w in s] + [entry["<pad>"]] * (width - len(s)) for s in seqs]
    return torch.tensor(rows, dtype=torch.long)


def anneal_weight(kind, step, k, x0):
    """KL / word-dropout schedule (utils/nn_funcs.py:26-36)."""
    if kind == "fixed":
        return 1.0
    if kind == "logistic":
        return float(1.0 / (1.0 + np.exp(-k * (step - x0))))
    if kind == "sigmoid":
        return float(1.0 / (1.0 + np.exp(0.001 * (x0 - step))))
    if kind == "negative-sigmoid":
        return float(1.0 / (1.0 + np.exp(-0.001 * (x0 - step))))
    if kind == "linear":
        return min(1, step / x0)
    return None


def word_dropout(ids, rate, entry, rand=None):
    """<unk> replacement, never on <s>/<pad>/</s> (utils/nn_funcs.py:13-23).

    ``rand`` injects the uniform draw the reference takes from torch.rand."""
    if not rate > 0.0:
        return ids
    prob = torch.rand(ids.shape) if rand is None else rand.clone()
    keep = (ids == entry.sos_id) | (ids == entry.pad_id) | (ids == entry.eos_id)
    prob = torch.where(keep, torch.ones_like(prob), prob)
    out = ids.clone()
    out[prob < rate] = entry.unk_id
    return out


def ids_to_words(ids, entry):
    """cut at first </s>, drop <s>/<pad> (utils/nn_funcs.py:57-66)."""
    words = []
    for i in ids:
        if i == entry.sos_id or i == entry.pad_id:
            continue
        if i == entry.eos_id:
            break
        words.append(entry.id2word[i])
    return words


# --------------------------------------------------------------------------------------
# GRU cell equations (torch.nn.GRU restated; SURVEY 8c "GRU definition to restate")
# --------------------------------------------------------------------------------------
def gru_loop(x, h0, w_ih, w_hh, b_ih, b_hh, lengths=None, reverse=False):
    """One GRU layer, one direction, batch-first x[B,T,I]; gate order r,z,n.

    r = s(W_ir x + b_ir + W_hr h + b_hr); z likewise; n = tanh(W_in x + b_in + r*(W_hn h + b_hn));
    h' = (1-z)*n + z*h.  With ``lengths`` the packed-sequence semantics of
    rnn_encoder.py:50-54 apply: row b only advances for t < lengths[b]; the reverse
    direction starts at each row's own last token; padded outputs are zero.
    Returns (out[B,T,H], h_last[B,H])."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = h0
    out = x.new_zeros(B, T, H)
    lens = None if lengths is None else torch.as_tensor(lengths)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gi = x[:, t] @ w_ih.t() + b_ih
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        hn = (1.0 - z) * n + z * h
        if lens is not None:
            live = (lens > t).to(x.dtype).unsqueeze(1)
            h = live * hn + (1.0 - live) * h
            out[:, t] = live * hn
        else:
            h = hn
            out[:, t] = hn
    return out, h


def bridge_state(state, dec_layers, dec_dim, mapper=None):
    """MLPBridger.forward (networks/bridger.py:27-37): [L,B,D] -> [dec_layers,B,dec_dim]."""
    B = state.size(1)
    flat = state.permute(1, 0, 2).contiguous().view(B, -1)
    if mapper is not None:
        flat = mapper(flat)
    return flat.view(-1, dec_layers, dec_dim).permute(1, 0, 2).contiguous()


def bow_nll(logp_row, targets, pad_id):
    """bag_of_word_loss + NLLLoss(ignore_index=pad, reduction='none') (utils/loss_funcs.py:117-129).

    logp_row [B,V] (one log-softmax row per sentence), targets [B,T] -> [B*T] per-token NLL
    (zero at pad).  <s> and </s> ARE counted."""
    B, T = targets.shape
    picked = -logp_row.gather(1, targets)
    picked = picked * (targets != pad_id).to(picked.dtype)
    return picked.reshape(-1)


# --------------------------------------------------------------------------------------
# network blocks
# --------------------------------------------------------------------------------------
class _OwnEmbedding(nn.Module):
    """networks/embeddings.py:13-87 with add_position_embedding=False (rnn_base.py:47-53)."""

    def __init__(self, n, dim, dropout):
        super().__init__()
        self.embeddings = nn.Embedding(n, dim)
        nn.init.uniform_(self.embeddings.weight, -0.1, 0.1)
        self.dropout = nn.Dropout(dropout) if dropout > 0.0 else None

    @property
    def weight(self):
        return self.embeddings.weight

    def forward(self, ids):
        e = self.embeddings(ids)
        return e if self.dropout is None else self.dropout(e)


class _Bridger(nn.Module):
    """networks/bridger.py:12-37.  'link' = identity (dims must agree), 'mapping' = Linear."""

    def __init__(self, mapper_type, enc_dim, enc_layers, dec_dim, dec_layers):
        super().__init__()
        self.in_dim, self.out_dim = enc_dim * enc_layers, dec_dim * dec_layers
        self.dec_dim, self.dec_layers = dec_dim, dec_layers
        if mapper_type == "mapping":
            self.mapper = nn.Linear(self.in_dim, self.out_dim)
        else:
            assert self.in_dim == self.out_dim, "bridge dim is not right"
            self.mapper = None

    def forward(self, state):
        return bridge_state(state, self.dec_layers, self.dec_dim, self.mapper)


class OracleEncoder(nn.Module):
    """RNNEncoder (encoder/rnn_encoder.py:7-55): embedding -> pack -> bi-GRU -> (out, h_n)."""

    def __init__(self, vocab_size, input_size, hidden_size, n_layers, bidirectional, embed_droprate=0.0,
                 rnn_droprate=0.0, embedding=None):
        super().__init__()
        self.embedding = embedding if embedding is not None else _OwnEmbedding(vocab_size, input_size,
                                                                               embed_droprate)
        self.rnn = nn.GRU(input_size, hidden_size, n_layers, batch_first=True, bidirectional=bidirectional,
                          dropout=rnn_droprate if n_layers > 1 else 0.0)

    def forward(self, ids, lengths):
        emb = self.embedding(ids)
        packed = nn.utils.rnn.pack_padded_sequence(emb, lengths, batch_first=True)
        out, h_n = self.rnn(packed)
        out, _ = nn.utils.rnn.pad_packed_sequence(out, batch_first=True)
        return out, h_n


class OracleDecoder(nn.Module):
    """RNNDecoder (decoder/rnn_decoder.py:22-203), use_attention=False, use_last_output=True."""

    def __init__(self, vocab_size, max_len, input_size, hidden_size, n_layers, sos_id, eos_id, pad_id=0,
                 embed_droprate=0.0, rnn_droprate=0.0, embedding=None):
        super().__init__()
        self.embedding = embedding if embedding is not None else _OwnEmbedding(vocab_size, input_size,
                                                                               embed_droprate)
        self.rnn = nn.GRU(input_size, hidden_size, n_layers, batch_first=True,
                          dropout=rnn_droprate if n_layers > 1 else 0.0)
        self.out = nn.Linear(hidden_size + input_size, vocab_size)
        self.max_length, self.sos_id, self.eos_id, self.pad_id = max_len, sos_id, eos_id, pad_id
        self.faithful_side_work = False  # the per-step topk + .cpu() of rnn_decoder.py:132-144,153-159

    def step_logits(self, ids, h):
        """forward_step with function=identity (rnn_decoder.py:61-74): -> ([B,T,V], h)."""
        emb = self.embedding(ids)
        o, h = self.rnn(emb, h)
        feats = torch.cat((o, emb), dim=-1)
        return self.out(feats.contiguous().view(-1, feats.size(-1))).view(ids.size(0), ids.size(1), -1), h

    def decode(self, inputs, h0):
        """teacher-forced raw logits [T,B,V] (rnn_decoder.py:76-86,149-159)."""
        logits, _ = self.step_logits(inputs[:, :-1], h0)
        if self.faithful_side_work:
            for t in range(logits.size(1)):
                sym = logits[:, t, :].topk(1)[1]
                sym.eq(self.eos_id).cpu().view(-1).numpy()
        return logits.transpose(0, 1)

    def score_decoding_results(self, scores, tgt):
        """log_softmax -> gather tgt[1:] -> pad mask -> sum over time -> [B] (rnn_decoder.py:88-108)."""
        T, B, V = scores.shape
        logp = F.log_softmax(scores.reshape(-1, V), dim=-1)
        gold = tgt.t()[1:].contiguous().view(-1)
        tok = logp.gather(1, gold.unsqueeze(1)).squeeze(1)
        tok = tok * (1.0 - gold.eq(self.pad_id).float())
        return tok.view(-1, B).sum(dim=0)

    def score(self, inputs, h0):
        return self.score_decoding_results(self.decode(inputs, h0), inputs)

    @torch.no_grad()
    def greedy(self, h0):
        """inputs=None branch (rnn_decoder.py:160-168,193-199): always max_length steps.

        Returns tokens [max_len,B] int64 and the host-side lengths bookkeeping list."""
        B = h0.size(1)
        tok = torch.full((B, 1), self.sos_id, dtype=torch.long)
        h = h0
        lengths = np.array([self.max_length] * B)
        seq = []
        for t in range(self.max_length):
            logits, h = self.step_logits(tok, h)
            tok = F.log_softmax(logits.squeeze(1), dim=-1).topk(1)[1]
            seq.append(tok)
            hit = tok.eq(self.eos_id).view(-1).numpy()
            upd = ((lengths > t) & hit) != 0
            lengths[upd] = len(seq)
        return torch.stack(seq).view(self.max_length, B), lengths.tolist()


class _SeqHead(nn.Module):
    """BridgeRNN (vae/syntax_vae.py:504-569) / RNNPredictor (vae/enhance_syntax_vae.py:517-592)."""

    def __init__(self, args, vocab, enc_dim, dec_dim, embed, mode):
        super().__init__()
        self.bridger = _Bridger(args.mapper_type, enc_dim, args.enc_num_layers, dec_dim, args.dec_num_layers)
        ent = vocab.src if mode == "src" else vocab.tgt
        self.decoder = OracleDecoder(
            len(ent), args.src_max_time_step if mode == "src" else args.tgt_max_time_step, args.dec_embed_dim,
            dec_dim, args.dec_num_layers, ent.sos_id, ent.eos_id, embed_droprate=args.dec_ed,
            rnn_droprate=args.dec_rd, embedding=embed)
        self.pad_id = ent.pad_id

    def neg_logp(self, hidden, tgt):
        return -self.decoder.score(tgt, self.bridger(hidden))

    def predict(self, hidden):
        return self.decoder.greedy(self.bridger(hidden))[0]


class _BowHead(nn.Module):
    """BridgeMLP (syntax_vae.py:572-602) / MLPPredictor (enhance_syntax_vae.py:595-637)."""

    def __init__(self, args, vocab, enc_dim, dec_dim):
        super().__init__()
        self.bridger = _Bridger(args.mapper_type, enc_dim, args.enc_num_layers, dec_dim, args.dec_num_layers)
        self.scorer = nn.Sequential(nn.Dropout(args.dec_rd),
                                    nn.Linear(dec_dim * args.dec_num_layers, len(vocab.src)),
                                    nn.LogSoftmax(dim=-1))
        self.pad_id = vocab.src.pad_id

    def forward(self, hidden, tgt):
        B = tgt.size(0)
        feat = self.bridger(hidden).permute(1, 0, 2).contiguous().view(B, -1)
        return bow_nll(self.scorer(feat), tgt, self.pad_id).view(B, -1).sum(-1).sum()


# --------------------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------------------
class _OracleBase(nn.Module):
    """BaseVAE (vae/base_vae.py:13-220): encode_to_hidden / predict / generating / KL."""

    def __init__(self, args, vocab):
        super().__init__()
        self.args, self.vocab = args, vocab
        self.step_kl_weight = args.init_step_kl_weight
        self.step_unk_rate = 0.0
        self.unk_rate = args.unk_rate  # captured at construction, as the reference does
        self.noise = {}  # optional injected draws: eps_syn, eps_sem, eps, rand_src, rand_tgt

    # base_vae.py:192-201
    def get_kl_weight(self, step):
        if self.step_kl_weight is None:
            return anneal_weight(self.args.anneal_function, step, self.args.k, self.args.x0)
        return self.step_kl_weight

    def compute_kl_loss(self, mean, logv, step):
        return -0.5 * torch.sum(1 + logv - mean.pow(2) - logv.exp()), self.get_kl_weight(step)

    def _draw(self, key, shape):
        v = self.noise.get(key)
        return torch.randn(shape) if v is None else v

    # base_vae.py:110-156
    def encode_to_hidden(self, examples, need_sort=False):
        if not isinstance(examples, list):
            examples = [examples]
        words = [e.src for e in examples]
        B = len(words)
        if not need_sort:
            lens = [len(w) for w in words]
            _, hid = self.encode(tensorise(words, self.vocab.src), lens)
            return {"hidden": hid, "length": lens, "batch_size": B}
        lens = [len(w) for w in words]
        order = sorted(range(B), key=lambda i: -lens[i])
        back = [0] * B
        for new, old in enumerate(order):
            back[old] = new
        sw = [words[i] for i in order]
        ids = tensorise(sw, self.vocab.src)
        if self.training and self.args.src_wd:
            ids = word_dropout(ids, self.step_unk_rate, self.vocab.src, self.noise.get("rand_src"))
        _, hid = self.encode(ids, [len(w) for w in sw])
        hid = hid[:, back, :] if hid.dim() > 2 else hid[back, :]
        return {"hidden": hid, "length": lens, "batch_size": B}

    # base_vae.py:169-190
    def decode_to_sentence(self, ret):
        toks, _ = self.decoder.greedy(self.bridger(ret["decode_init"]))
        res = [ids_to_words(toks[:, i].tolist(), self.vocab.src) for i in range(toks.size(1))]
        return [[[r], [len(r)]] for r in res], toks

    def predict_tokens(self, examples):
        """predict() (base_vae.py:56-67) but returning the raw [max_len,B] token matrix."""
        ret = self.encode_to_hidden(examples)
        ret = self.hidden_to_latent(ret, is_sampling=self.training)
        ret = self.latent_for_init(ret)
        return self.decode_to_sentence(ret)[1]


class OracleVanillaVAE(_OracleBase):
    """VanillaVAE (vae/vanilla_vae.py:15-224); the tgt_wd train branch implements the evident
    intent (decode on word-dropped inputs) since vanilla_vae.py:112 calls a missing method."""

    def __init__(self, args, vocab):
        super().__init__(args, vocab)
        embed = nn.Embedding(len(vocab.src), args.embed_size)
        if args.share_embed:
            dec_embed = embed
            args.dec_embed_dim = args.enc_embed_dim
        else:
            dec_embed = None
        self.hidden_size = args.enc_hidden_dim
        dirs = 2 if args.bidirectional else 1
        self.hidden_factor = dirs * args.enc_num_layers
        enc_dim = args.enc_hidden_dim * dirs
        dec_hidden = enc_dim if args.mapper_type == "link" else args.dec_hidden_dim
        self.encoder = OracleEncoder(len(vocab.src), args.enc_embed_dim, args.enc_hidden_dim, args.enc_num_layers,
                                     args.bidirectional, args.enc_ed, args.enc_rd, embedding=embed)
        self.bridger = _Bridger(args.mapper_type, enc_dim, args.enc_num_layers, dec_hidden, args.dec_num_layers)
        self.decoder = OracleDecoder(len(vocab.src), args.src_max_time_step, args.dec_embed_dim, dec_hidden,
                                     args.dec_num_layers, vocab.src.sos_id, vocab.src.eos_id,
                                     embed_droprate=args.dec_ed, rnn_droprate=args.dec_rd, embedding=dec_embed)
        self.hidden2mean = nn.Linear(args.hidden_size * self.hidden_factor, args.latent_size)
        self.hidden2logv = nn.Linear(args.hidden_size * self.hidden_factor, args.latent_size)
        self.latent2hidden = nn.Linear(args.latent_size, args.hidden_size * self.hidden_factor)

    def encode(self, ids, lens):
        if self.training and self.args.src_wd:
            ids = word_dropout(ids, self.step_unk_rate, self.vocab.src, self.noise.get("rand_src"))
        return self.encoder(ids, lens)

    def hidden_to_latent(self, ret, is_sampling=True):
        h = ret["hidden"]
        B = h.size(1)
        h = h.permute(1, 0, 2).contiguous().view(B, -1)
        mean, logv = self.hidden2mean(h), self.hidden2logv(h)
        z = self._draw("eps", [B, self.args.latent_size]) * torch.exp(0.5 * logv) + mean if is_sampling else mean
        ret.update(latent=z, mean=mean, logv=logv)
        return ret

    def sample_latent(self, B):
        return {"latent": self._draw("eps", [B, self.args.latent_size])}

    def latent_for_init(self, ret):
        z = ret["latent"]
        h = self.latent2hidden(z)
        ret["decode_init"] = h.view(z.size(0), self.hidden_factor, self.hidden_size).permute(1, 0, 2)
        return ret

    def get_loss(self, examples, step, **_):
        self.step_unk_rate = self.unk_rate * anneal_weight(self.args.unk_schedule, step, self.args.k, self.args.x0)
        if not isinstance(examples, list):
            examples = [examples]
        B = len(examples)
        ret = self.latent_for_init(self.hidden_to_latent(self.encode_to_hidden(examples), self.training))
        tgt = tensorise([e.src for e in examples], self.vocab.src, boundary=True)
        h0 = self.bridger(ret["decode_init"])
        if self.training and self.args.tgt_wd > 0.0:
            inp = word_dropout(tgt, self.step_unk_rate, self.vocab.src, self.noise.get("rand_tgt"))
            nll = -torch.sum(self.decoder.score_decoding_results(self.decoder.decode(inp, h0), tgt))
        else:
            nll = -torch.sum(self.decoder.score(tgt, h0))
        kl, w = self.compute_kl_loss(ret["mean"], ret["logv"], step)
        w = w * self.args.kl_factor
        nll, kl = nll / B, kl / B
        return {"KL Loss": kl, "NLL Loss": nll, "KL Weight": w, "Model Score": kl + nll, "ELBO": kl * w + nll,
                "Loss": kl * w + nll, "KL Item": kl * w}


class _TwoEncoderBase(_OracleBase):
    """Shared body of SyntaxVAE (vae/syntax_vae.py) and EnhanceSyntaxVAE (vae/enhance_syntax_vae.py)."""

    ENC_NAMES = ("word_encoder", "syntax_encoder")

    def _build_common(self, args, vocab):
        self.src_embed = nn.Embedding(len(vocab.src), args.embed_size)
        self.tgt_embed = nn.Embedding(len(vocab.tgt), args.embed_size)
        self.latent_size = int(args.latent_size)
        dirs = 2 if args.bidirectional else 1
        self.enc_hidden_dim = args.enc_hidden_dim
        self.enc_layer_dim = args.enc_hidden_dim * dirs
        self.enc_hidden_factor = dirs * args.enc_num_layers
        args.use_attention = False
        self.dec_layer_dim = self.enc_layer_dim if args.mapper_type == "link" else args.dec_hidden_dim
        var_dim = int(self.enc_hidden_dim * self.enc_hidden_factor / 2)
        self.task_enc_dim, self.task_dec_dim = int(self.enc_layer_dim / 2), int(self.dec_layer_dim / 2)
        for name in self.ENC_NAMES:
            setattr(self, name, OracleEncoder(len(vocab.src), args.enc_embed_dim, int(self.enc_hidden_dim / 2),
                                              args.enc_num_layers, args.bidirectional, args.enc_ed, args.enc_rd,
                                              embedding=self.src_embed))
        main = _SeqHead(args, vocab, self.enc_layer_dim, self.dec_layer_dim,
                        self.src_embed if args.share_embed else None, "src")
        self.bridger, self.decoder = main.bridger, main.decoder
        self.syn_mean = nn.Linear(var_dim, self.latent_size)
        self.syn_logv = nn.Linear(var_dim, self.latent_size)
        self.syn_to_h = nn.Linear(self.latent_size, var_dim)
        self.sem_mean = nn.Linear(var_dim, self.latent_size)
        self.sem_logv = nn.Linear(var_dim, self.latent_size)
        self.sem_to_h = nn.Linear(self.latent_size, var_dim)

    # syntax_vae.py:197-207 / enhance_syntax_vae.py:198-210: first half semantic, second syntactic
    def encode(self, ids, lens):
        if self.training and self.args.src_wd > 0.0:
            ids = word_dropout(ids, self.step_unk_rate, self.vocab.src, self.noise.get("rand_src"))
        a, b = (getattr(self, n) for n in self.ENC_NAMES)
        _, ha = a(ids, lens)
        _, hb = b(ids, lens)
        B = ha.size(1)
        return None, torch.cat([ha.permute(1, 0, 2).contiguous().view(B, -1),
                                hb.permute(1, 0, 2).contiguous().view(B, -1)], dim=-1)

    # syntax_vae.py:390-424 / enhance_syntax_vae.py:413-440: syntax eps is drawn FIRST
    def hidden_to_latent(self, ret, is_sampling=True):
        hid = ret["hidden"]
        B = hid.size(0)
        halves = hid.contiguous().view(B, 2, -1)
        sem_h, syn_h = halves[:, 0, :].contiguous(), halves[:, 1, :].contiguous()
        ret["syn_mean"], ret["syn_logv"] = self.syn_mean(syn_h), self.syn_logv(syn_h)
        ret["sem_mean"], ret["sem_logv"] = self.sem_mean(sem_h), self.sem_logv(sem_h)
        if is_sampling:
            e_syn = self._draw("eps_syn", [B, self.latent_size])
            e_sem = self._draw("eps_sem", [B, self.latent_size])
            ret["syn_z"] = e_syn * torch.exp(0.5 * ret["syn_logv"]) + ret["syn_mean"]
            ret["sem_z"] = e_sem * torch.exp(0.5 * ret["sem_logv"]) + ret["sem_mean"]
        else:
            ret["syn_z"], ret["sem_z"] = ret["syn_mean"], ret["sem_mean"]
        return ret

    def sample_latent(self, B):
        return {"syn_z": self._draw("eps_syn", [B, self.latent_size]),
                "sem_z": self._draw("eps_sem", [B, self.latent_size])}

    # syntax_vae.py:426-441 / enhance_syntax_vae.py:442-457
    def latent_for_init(self, ret):
        B = ret["sem_z"].size(0)

        def lay(x):
            return x.view(B, self.enc_hidden_factor, int(self.enc_hidden_dim / 2)).permute(1, 0, 2)

        ret["syn_hidden"] = lay(self.syn_to_h(ret["syn_z"]))
        ret["sem_hidden"] = lay(self.sem_to_h(ret["sem_z"]))
        ret["decode_init"] = torch.cat([ret["syn_hidden"], ret["sem_hidden"]], dim=-1)
        return ret

    def _front(self, examples):
        if not isinstance(examples, list):
            examples = [examples]
        tgt = tensorise([e.src for e in examples], self.vocab.src, boundary=True)
        syn = tensorise([e.tgt for e in examples], self.vocab.tgt, boundary=True)
        ret = self.latent_for_init(self.hidden_to_latent(self.encode_to_hidden(examples), self.training))
        return examples, tgt, syn, ret

    def _reconstruct(self, ret, tgt):
        h0 = self.bridger(ret["decode_init"])
        if self.training and self.args.tgt_wd:
            inp = word_dropout(tgt, self.step_unk_rate, self.vocab.src, self.noise.get("rand_tgt"))
            return -self.decoder.score_decoding_results(self.decoder.decode(inp, h0), tgt)
        return -self.decoder.score(tgt, h0)

    def _kl_terms(self, ret, step):
        sem_kl, w = self.compute_kl_loss(ret["sem_mean"], ret["sem_logv"], step)
        syn_kl, _ = self.compute_kl_loss(ret["syn_mean"], ret["syn_logv"], step)
        a = self.args
        kl = (a.sem_weight * sem_kl + a.syn_weight * syn_kl) / (a.sem_weight + a.syn_weight)
        return sem_kl, syn_kl, kl, w * a.kl_factor

    def conditional_generating(self, condition="sem", examples=None):
        ret = self.hidden_to_latent(self.encode_to_hidden(examples), is_sampling=True)
        if condition.startswith("sem"):
            ret["sem_z"] = ret["sem_mean"]
        else:
            ret["syn_z"] = ret["syn_mean"]
        if condition == "sem-only":
            ret["syn_z"] = self._draw("eps_syn_prior", [ret["batch_size"], self.latent_size])
        ret = self.latent_for_init(ret)
        return self.decode_to_sentence(ret)[1]

    def transfer_tokens(self, sem_in, syn_ref):
        """eval_adv (enhance_syntax_vae.py:485-497): sem_z of sem_in, syn_z of syn_ref."""
        a = self.hidden_to_latent(self.encode_to_hidden(sem_in), is_sampling=self.training)
        b = self.hidden_to_latent(self.encode_to_hidden(syn_ref, need_sort=True), is_sampling=self.training)
        a, b = self.latent_for_init(a), self.latent_for_init(b)
        mix = self.latent_for_init(dict(sem_z=a["sem_z"], syn_z=b["syn_z"]))
        return self.decode_to_sentence(mix)[1], a, b


class OracleSyntaxVAE(_TwoEncoderBase):
    """SyntaxVAE (vae/syntax_vae.py:50-602).  Quirks kept: tree-decoder loss is a per-sentence
    [B] vector (syntax_vae.py:547-556) so MUL/ADV/Loss are vectors; adversaries run under
    no_grad in train mode (:322-324)."""

    ENC_NAMES = ("word_encoder", "syntax_encoder")

    def __init__(self, args, vocab):
        super().__init__(args, vocab)
        self._build_common(args, vocab)
        te, td = self.task_enc_dim, self.task_dec_dim
        self.sup_syn = _SeqHead(args, vocab, te, td, None, "tgt")
        self.sup_sem = _BowHead(args, vocab, te, td)
        self.syn_adv = _SeqHead(args, vocab, te, td, self.tgt_embed if args.share_embed else None, "tgt")
        self.syn_infer = _SeqHead(args, vocab, te, td, self.src_embed if args.share_embed else None, "src")
        self.sem_adv = _BowHead(args, vocab, te, td)
        self.sem_infer = _SeqHead(args, vocab, te, td, self.src_embed if args.share_embed else None, "src")

    def _dis_terms(self, syn_h, sem_h, syn_tgt, sem_tgt):
        a = self.args
        d = {"adv_syn_sup": self.syn_adv.neg_logp(sem_h, syn_tgt) if a.adv_syn > 0.0 else 0.0,
             "adv_sem_sup": self.sem_adv(syn_h, sem_tgt) if a.adv_sem > 0.0 else 0.0}
        if a.infer_weight > 0.0:
            d["adv_syn_inf"] = self.syn_infer.neg_logp(syn_h, sem_tgt) if a.inf_syn > 0.0 else 0.0
            d["adv_sem_inf"] = self.sem_infer.neg_logp(sem_h, sem_tgt) if a.inf_sem > 0.0 else 0.0
        return d

    def _adv(self, syn_h, sem_h, syn_tgt, sem_tgt):
        a = self.args
        if self.training:
            with torch.no_grad():
                d = self._dis_terms(syn_h, sem_h, syn_tgt, sem_tgt)
        else:
            d = self._dis_terms(syn_h, sem_h, syn_tgt, sem_tgt)
        if a.infer_weight > 0.0:
            return (a.adv_syn * d["adv_syn_sup"] + a.infer_weight * a.inf_sem * d["adv_sem_inf"],
                    a.adv_sem * d["adv_sem_sup"] + a.infer_weight * a.inf_syn * d["adv_syn_inf"])
        return a.adv_syn * d["adv_syn_sup"], a.adv_sem * d["adv_sem_sup"]

    def get_loss(self, examples, train_iter, is_dis=False, **_):
        a = self.args
        self.step_unk_rate = self.unk_rate * anneal_weight(a.unk_schedule, train_iter, a.k, a.x0)
        examples, tgt, syn, ret = self._front(examples)
        B = len(examples)
        syn_h, sem_h = ret["syn_hidden"], ret["sem_hidden"]
        if is_dis:
            d = self._dis_terms(syn_h.detach(), sem_h.detach(), syn, tgt)
            if a.infer_weight > 0.0:
                ret["dis syn"] = d["adv_syn_sup"] + d["adv_sem_inf"]
                ret["dis sem"] = d["adv_sem_sup"] + d["adv_syn_inf"]
            else:
                ret["dis syn"], ret["dis sem"] = d["adv_syn_sup"], d["adv_sem_sup"]
            return ret
        nll = torch.sum(self._reconstruct(ret, tgt))
        mul = a.mul_syn * self.sup_syn.neg_logp(syn_h, syn) + a.mul_sem * self.sup_sem(sem_h, tgt)
        adv_syn, adv_sem = self._adv(syn_h, sem_h, syn, tgt)
        adv = adv_syn + adv_sem
        sem_kl, syn_kl, kl, w = self._kl_terms(ret, train_iter)
        kl, mul, adv, nll = kl / B, mul / B, adv / B, nll / B
        item = kl * w
        return {"KL Loss": kl, "NLL Loss": nll, "MUL Loss": mul, "ADV Loss": adv, "KL Weight": w,
                "KL Item": item, "Model Score": kl + nll, "ELBO": item + nll,
                "Loss": item + nll + mul - adv, "SYN KL Loss": syn_kl / B, "SEM KL Loss": sem_kl / B}


class OracleEnhanceSyntaxVAE(_TwoEncoderBase):
    """EnhanceSyntaxVAE, registry key SyntaxVAE2 (vae/enhance_syntax_vae.py:47-637).  Scalar heads;
    adversary parameters frozen (requires_grad toggled) so gradient reaches the latents
    (:338-343,:573-579); the -adv term only enters Loss when train_iter > warm_up (:333)."""

    ENC_NAMES = ("sem_encoder", "syn_encoder")
    ADVERSARIES = ("sem_to_sent", "syn_to_sent", "syn_to_sem", "sem_to_syn")

    def __init__(self, args, vocab):
        super().__init__(args, vocab)
        self._build_common(args, vocab)
        te, td = self.task_enc_dim, self.task_dec_dim
        self.mul_syn = _SeqHead(args, vocab, te, td, None, "tgt")
        self.mul_sem = _BowHead(args, vocab, te, td)
        self.sem_to_syn = _SeqHead(args, vocab, te, td, self.tgt_embed if args.share_embed else None, "tgt")
        self.syn_to_sem = _BowHead(args, vocab, te, td)
        self.sem_to_sent = _SeqHead(args, vocab, te, td, self.src_embed if args.share_embed else None, "src")
        self.syn_to_sent = _SeqHead(args, vocab, te, td, self.src_embed if args.share_embed else None, "src")

    def _freeze(self, flag):
        for n in self.ADVERSARIES:
            for p in getattr(self, n).parameters():
                p.requires_grad = not flag

    def _dis_terms(self, syn_h, sem_h, syn_tgt, sem_tgt):
        a = self.args
        return {"sem_to_syn": self.sem_to_syn.neg_logp(sem_h, syn_tgt).sum() if a.sem_to_syn > 0.0 else 0.0,
                "syn_to_sem": self.syn_to_sem(syn_h, sem_tgt) if a.syn_to_sem > 0.0 else 0.0,
                "sem_to_sent": self.sem_to_sent.neg_logp(sem_h, sem_tgt).sum() if a.sem_to_sent > 0.0 else 0.0,
                "syn_to_sent": self.syn_to_sent.neg_logp(syn_h, sem_tgt).sum() if a.syn_to_sent > 0.0 else 0.0}

    def get_loss(self, examples, train_iter, is_dis=False, **_):
        a = self.args
        self.step_unk_rate = self.unk_rate * anneal_weight(a.unk_schedule, train_iter, a.k, a.x0)
        examples, tgt, syn, ret = self._front(examples)
        B = len(examples)
        syn_h, sem_h = ret["syn_hidden"], ret["sem_hidden"]
        if is_dis:
            self._freeze(False)
            d = self._dis_terms(syn_h.detach(), sem_h.detach(), syn, tgt)
            if a.infer_weight > 0.0:
                ret["dis syn"] = d["syn_to_sem"] + d["syn_to_sent"]
                ret["dis sem"] = d["sem_to_syn"] + d["sem_to_sent"]
            else:
                ret["dis syn"], ret["dis sem"] = d["syn_to_sem"], d["sem_to_syn"]
            return ret
        nll = self._reconstruct(ret, tgt).sum()
        mul = a.mul_syn * self.mul_syn.neg_logp(syn_h, syn).sum() + a.mul_sem * self.mul_sem(sem_h, tgt)
        if self.training:
            self._freeze(True)
        d = self._dis_terms(syn_h, sem_h, syn, tgt)
        if a.infer_weight > 0.0:
            adv = (a.syn_to_sem * d["syn_to_sem"] + a.infer_weight * a.syn_to_sent * d["syn_to_sent"]
                   + a.sem_to_syn * d["sem_to_syn"] + a.infer_weight * a.sem_to_sent * d["sem_to_sent"])
        else:
            adv = a.syn_to_sem * d["syn_to_sem"] + a.sem_to_syn * d["sem_to_syn"]
        sem_kl, syn_kl, kl, w = self._kl_terms(ret, train_iter)
        kl, mul, adv, nll = kl / B, mul / B, adv / B, nll / B
        item = kl * w
        loss = item + nll + mul - adv if train_iter > a.warm_up else item + nll + mul
        return {"KL Loss": kl, "NLL Loss": nll, "MUL Loss": mul, "ADV Loss": adv, "KL Weight": w,
                "KL Item": item, "Model Score": kl + nll, "ELBO": item + nll, "Loss": loss,
                "SYN KL Loss": syn_kl / B, "SEM KL Loss": sem_kl / B}

    def syntax_tokens(self, examples):
        """eval_syntax (enhance_syntax_vae.py:479-483): greedy tree decode through mul_syn."""
        ret = self.latent_for_init(self.hidden_to_latent(self.encode_to_hidden(examples, need_sort=True), False))
        return self.mul_syn.predict(ret["syn_hidden"])


class _BowHeadMean(nn.Module):
    """DisentangleVAE's BridgeMLP (vae/disentangle_vae.py:570-615): same scorer, but the criterion is
    nn.NLLLoss(ignore_index=pad) with its default MEAN reduction over all non-pad targets of the batch
    (:607); with a `stack_mlp` key in args the scorer is Dropout-Linear-ReLU-Linear-LogSoftmax (:591-606)."""

    def __init__(self, args, vocab, enc_dim, dec_dim):
        super().__init__()
        self.bridger = _Bridger(args.mapper_type, enc_dim, args.enc_num_layers, dec_dim, args.dec_num_layers)
        if "stack_mlp" not in args:
            self.scorer = nn.Sequential(nn.Dropout(args.dec_rd),
                                        nn.Linear(dec_dim * args.dec_num_layers, len(vocab.src)),
                                        nn.LogSoftmax(dim=-1))
        else:
            self.scorer = nn.Sequential(nn.Dropout(args.dec_rd), nn.Linear(dec_dim * args.dec_num_layers, dec_dim),
                                        nn.ReLU(), nn.Linear(dec_dim, len(vocab.src)), nn.LogSoftmax(dim=-1))
        self.pad_id = vocab.src.pad_id

    def forward(self, hidden, tgt):
        B = tgt.size(0)
        feat = self.bridger(hidden).permute(1, 0, 2).contiguous().view(B, -1)
        return bow_nll(self.scorer(feat), tgt, self.pad_id).sum() / tgt.ne(self.pad_id).sum().to(torch.float32)


class OracleDisentangleVAE(_TwoEncoderBase):
    """DisentangleVAE (vae/disentangle_vae.py:33-500) with the int() fix of its latent_for_init (:427; the
    fixed twin is vae/syntax_vae.py:428): ONE bi-GRU encoder of full width whose per-(layer, direction)
    final state is cut in halves, first half semantic, second syntactic (:400-404); latent heads with a
    shared Linear+ReLU trunk when args has a `report` key (:97-121); scalar head losses (:546-554); BoW
    heads with mean reduction; adversaries under no_grad in train mode (:320-322)."""

    def __init__(self, args, vocab):
        _OracleBase.__init__(self, args, vocab)
        self.src_embed = nn.Embedding(len(vocab.src), args.embed_size)
        self.tgt_embed = nn.Embedding(len(vocab.tgt), args.embed_size)
        self.latent_size = int(args.latent_size)
        dirs = 2 if args.bidirectional else 1
        self.enc_hidden_dim = args.enc_hidden_dim
        self.enc_layer_dim = args.enc_hidden_dim * dirs
        self.enc_hidden_factor = dirs * args.enc_num_layers
        args.use_attention = False
        self.dec_layer_dim = self.enc_layer_dim if args.mapper_type == "link" else args.dec_hidden_dim
        var_dim = int(self.enc_hidden_dim * self.enc_hidden_factor / 2)
        te, td = int(self.enc_layer_dim / 2), int(self.dec_layer_dim / 2)
        self.encoder = OracleEncoder(len(vocab.src), args.enc_embed_dim, self.enc_hidden_dim, args.enc_num_layers,
                                     args.bidirectional, args.enc_ed, args.enc_rd, embedding=self.src_embed)
        main = _SeqHead(args, vocab, self.enc_layer_dim, self.dec_layer_dim,
                        self.src_embed if args.share_embed else None, "src")
        self.bridger, self.decoder = main.bridger, main.decoder
        Z = self.latent_size
        if "report" in args:
            syn_common = nn.Sequential(nn.Linear(var_dim, 2 * Z), nn.ReLU())
            self.syn_mean = nn.Sequential(syn_common, nn.Linear(2 * Z, Z))
            self.syn_logv = nn.Sequential(syn_common, nn.Linear(2 * Z, Z))
            sem_common = nn.Sequential(nn.Linear(var_dim, 2 * Z), nn.ReLU())
            self.sem_mean = nn.Sequential(sem_common, nn.Linear(2 * Z, Z))
            self.sem_logv = nn.Sequential(sem_common, nn.Linear(2 * Z, Z))
        else:
            self.syn_mean, self.syn_logv = nn.Linear(var_dim, Z), nn.Linear(var_dim, Z)
            self.sem_mean, self.sem_logv = nn.Linear(var_dim, Z), nn.Linear(var_dim, Z)
        self.syn_to_h = nn.Linear(Z, var_dim)
        self.sem_to_h = nn.Linear(Z, var_dim)
        self.sup_syn = _SeqHead(args, vocab, te, td, None, "tgt")
        self.sup_sem = _BowHeadMean(args, vocab, te, td)
        self.syn_adv = _SeqHead(args, vocab, te, td, self.tgt_embed if args.share_embed else None, "tgt")
        self.syn_infer = _SeqHead(args, vocab, te, td, self.src_embed if args.share_embed else None, "src")
        self.sem_adv = _BowHeadMean(args, vocab, te, td)
        self.sem_infer = _SeqHead(args, vocab, te, td, self.src_embed if args.share_embed else None, "src")

    def encode(self, ids, lens):
        if self.training and self.args.src_wd > 0.0:
            ids = word_dropout(ids, self.step_unk_rate, self.vocab.src, self.noise.get("rand_src"))
        return self.encoder(ids, lens)

    def hidden_to_latent(self, ret, is_sampling=True):
        hid = ret["hidden"]  # [layers*dirs, B, H]
        B, F_ = hid.size(1), hid.size(0)
        cut = hid.permute(1, 0, 2).contiguous().view(B, F_, 2, -1)
        sem_h, syn_h = cut[:, :, 0, :].contiguous().view(B, -1), cut[:, :, 1, :].contiguous().view(B, -1)
        ret["sem_mean"], ret["sem_logv"] = self.sem_mean(sem_h), self.sem_logv(sem_h)
        ret["syn_mean"], ret["syn_logv"] = self.syn_mean(syn_h), self.syn_logv(syn_h)
        if is_sampling:  # syntax eps first (:413-414)
            e_syn = self._draw("eps_syn", [B, self.latent_size])
            e_sem = self._draw("eps_sem", [B, self.latent_size])
            ret["syn_z"] = e_syn * torch.exp(0.5 * ret["syn_logv"]) + ret["syn_mean"]
            ret["sem_z"] = e_sem * torch.exp(0.5 * ret["sem_logv"]) + ret["sem_mean"]
        else:
            ret["syn_z"], ret["sem_z"] = ret["syn_mean"], ret["sem_mean"]
        return ret

    def _dis_terms(self, syn_h, sem_h, syn_tgt, sem_tgt):
        a = self.args
        d = {"adv_syn_sup": self.syn_adv.neg_logp(sem_h, syn_tgt).sum(), "adv_sem_sup": self.sem_adv(syn_h, sem_tgt)}
        if a.infer_weight > 0.0:
            d = {"adv_syn_sup": d["adv_syn_sup"] if a.adv_syn > 0.0 else 0.0,
                 "adv_sem_sup": d["adv_sem_sup"] if a.adv_sem > 0.0 else 0.0,
                 "adv_syn_inf": self.syn_infer.neg_logp(syn_h, sem_tgt).sum() if a.inf_syn > 0.0 else 0.0,
                 "adv_sem_inf": self.sem_infer.neg_logp(sem_h, sem_tgt).sum() if a.inf_sem > 0.0 else 0.0}
        return d

    def get_loss(self, examples, step, is_dis=False, **_):
        a = self.args
        self.step_unk_rate = self.unk_rate * anneal_weight(a.unk_schedule, step, a.k, a.x0)
        examples, tgt, syn, ret = self._front(examples)
        B = len(examples)
        syn_h, sem_h = ret["syn_hidden"], ret["sem_hidden"]
        if is_dis:
            d = self._dis_terms(syn_h.detach(), sem_h.detach(), syn, tgt)
            if a.infer_weight > 0.0:
                ret["dis syn"] = d["adv_syn_sup"] + d["adv_sem_inf"]
                ret["dis sem"] = d["adv_sem_sup"] + d["adv_syn_inf"]
            else:
                ret["dis syn"], ret["dis sem"] = d["adv_syn_sup"], d["adv_sem_sup"]
            return ret
        nll = torch.sum(self._reconstruct(ret, tgt))
        mul = a.mul_syn * self.sup_syn.neg_logp(syn_h, syn).sum() + a.mul_sem * self.sup_sem(sem_h, tgt)
        with torch.set_grad_enabled(torch.is_grad_enabled() and not self.training):
            d = self._dis_terms(syn_h, sem_h, syn, tgt)
        if a.infer_weight > 0.0:
            adv = (a.adv_syn * d["adv_syn_sup"] + a.infer_weight * a.inf_sem * d["adv_sem_inf"]
                   + a.adv_sem * d["adv_sem_sup"] + a.infer_weight * a.inf_syn * d["adv_syn_inf"])
        else:
            adv = a.adv_syn * d["adv_syn_sup"] + a.adv_sem * d["adv_sem_sup"]
        sem_kl, syn_kl, kl, w = self._kl_terms(ret, step)
        kl, mul, adv, nll = kl / B, mul / B, adv / B, nll / B
        item = kl * w
        return {"KL Loss": kl, "NLL Loss": nll, "MUL Loss": mul, "ADV Loss": adv, "KL Weight": w,
                "KL Item": item, "Model Score": kl + nll, "ELBO": item + nll,
                "Loss": item + nll + mul - adv, "SYN KL Loss": syn_kl / B, "SEM KL Loss": sem_kl / B}

    def syntax_tokens(self, examples):
        """eval_syntax (vae/disentangle_vae.py:465-469): greedy tree decode through sup_syn."""
        ret = self.latent_for_init(self.hidden_to_latent(self.encode_to_hidden(examples, need_sort=True), False))
        return self.sup_syn.predict(ret["syn_hidden"])


ORACLE_MODELS = {"VanillaVAE": OracleVanillaVAE, "SyntaxVAE": OracleSyntaxVAE,
                 "SyntaxVAE2": OracleEnhanceSyntaxVAE, "DisentangleVAE": OracleDisentangleVAE}


def init_uniform_(model, seed=0, bound=0.08):
    """build_model's re-init of every parameter (utils/model_utils.py:189-190), seeded."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.empty(p.shape).uniform_(-bound, bound, generator=g))
    return model


def oracle_train_step(model, optimizer, examples, train_iter, clip=5.0):
    """bin/train.py:84-88,119-126: zero_grad -> get_loss -> mean(Loss).backward -> clip -> Adam."""
    optimizer.zero_grad()
    ret = model.get_loss(examples, train_iter)
    torch.mean(ret["Loss"]).backward()
    gn = None
    if clip > 0.0:
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
    optimizer.step()
    return ret, gn
