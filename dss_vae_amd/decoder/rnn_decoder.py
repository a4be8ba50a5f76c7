"""RNNDecoder (reference: dss_vae/decoder/rnn_decoder.py:22-203; use_attention=False,
use_last_output=True).  Teacher-forced scoring never materialises [B,T,V] logits: the
``[h_t ; e_t] W_out^T`` projection is fused with an online log-sum-exp cross-entropy
(ops.linear_ce); the per-step topk/.cpu() side work of rnn_decoder.py:132-144 is gone.
``score_many`` runs several decoders (main decoder + task heads) layer by layer so that their
recurrences share persistent GRU launches."""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..networks.embeddings import lookup
from ..networks.rnn_base import GLOBAL_SEEDS, RNNBase


class RNNDecoder(RNNBase):
    KEY_ATTN_SCORE = "attention_score"
    KEY_LENGTH = "length"
    KEY_SEQUENCE = "sequence"

    def __init__(self, vocab, max_len, input_size, hidden_size, eos_id, sos_id, pad_id=0, n_layers=1, rnn_cell="gru",
                 embed_droprate=0, rnn_droprate=0, use_attention=False, embedding=None, update_embedding=True,
                 use_last_output=True):
        super().__init__(vocab, max_len, input_size, hidden_size, embed_droprate, rnn_droprate, n_layers, rnn_cell)
        if use_attention or not use_last_output:
            raise NotImplementedError("the VAE hot path forces use_attention=False (vae/enhance_syntax_vae.py:71) "
                                      "and use_last_output=True")
        self.rnn = self.make_rnn(input_size, hidden_size, n_layers, dropout=rnn_droprate)
        self.max_length = max_len
        self.use_attention, self.use_last_output = False, True
        self.eos_id, self.sos_id, self.pad_id = eos_id, sos_id, pad_id
        self.soft_input_size = hidden_size + input_size
        self.out = nn.Linear(self.soft_input_size, self.vocab_size)
        if embedding is not None:
            self.embedding = embedding
        self.embedding.weight.requires_grad = update_embedding

    # ---- teacher-forced scoring (fused) ------------------------------------------------------
    def score_inputs(self, inputs, targets, encoder_hidden, norm_by_word=False, uniform=False):
        """log p(targets[:,1:] | inputs[:,:-1], h0) per sentence, [B].  uniform=True: the caller only
        ever sums the result with one common weight (see ops._LinearCEFn)."""
        if self.is_lstm:
            out, emb, _ = self._features(inputs[:, :-1].contiguous(), encoder_hidden)
            B, T, H = out.shape
            logp = -ops.linear_ce(out.reshape(B * T, H), emb.reshape(B * T, -1), self.out.weight, self.out.bias,
                                  targets[:, 1:].reshape(-1), self.pad_id, uniform_upstream=uniform, group=T)
            if norm_by_word:
                logp = logp.div(targets.ne(self.pad_id).sum(dim=-1).float())
            return logp
        return score_many([dict(dec=self, inputs=inputs, targets=targets, h0=encoder_hidden,
                                norm_by_word=norm_by_word, uniform=uniform)])[0]

    def score(self, inputs, encoder_hidden, encoder_outputs=None, norm_by_word=False):
        return self.score_inputs(inputs, inputs, encoder_hidden, norm_by_word)

    # ---- compatibility: materialised logits (not used by the model hot path) -------------------
    def _features(self, ids, h0):
        emb = lookup(self.embedding, ids, GLOBAL_SEEDS)
        if self.is_lstm:
            # hidden is nn.LSTM's (h_0, c_0), each [layers,B,H] (decoder/rnn_decoder.py:65 with rnn_cell='lstm')
            h0s, c0s = h0
            x, hl, cl = emb, [], []
            for layer in range(self.n_layers):
                if layer > 0 and self.training and self.rnn_droprate > 0.0:
                    x = ops.dropout(x, self.rnn_droprate, GLOBAL_SEEDS.next())
                wih, whh, bih, bhh = self.rnn.layer_params(layer)
                x, h, c = ops.lstm_layer(ops.linear(x, wih, bih.detach()), whh, bhh, h0=h0s[layer], c0=c0s[layer],
                                         bih=bih)
                hl.append(h)
                cl.append(c)
            return x, emb, (torch.stack(hl, 0), torch.stack(cl, 0))
        x, hl = emb, []
        for layer in range(self.n_layers):
            if layer > 0 and self.training and self.rnn_droprate > 0.0:
                x = ops.dropout(x, self.rnn_droprate, GLOBAL_SEEDS.next())
            wih, whh, bih, bhh = self.rnn.layer_params(layer)
            (x,), (h,) = ops.gru_layer([dict(xproj=ops.linear(x, wih, bih), whh=whh, bhh=bhh, h0=h0[layer],
                                             group=0, col=0)])
            hl.append(h)
        return x, emb, torch.stack(hl, 0)

    def forward_step(self, input_var, hidden, encoder_outputs=None, function=F.log_softmax):
        out, emb, hidden = self._features(input_var, hidden)
        logits = ops.linear(torch.cat((out, emb), dim=-1), self.out.weight, self.out.bias)
        return function(logits, dim=-1), hidden, None

    def decode(self, inputs, encoder_hidden, encoder_outputs=None, teacher_forcing_ratio=1.0):
        """raw logits [T,B,V] (rnn_decoder.py:76-86).  Materialises; kept for interface parity only."""
        out, emb, _ = self._features(inputs[:, :-1], encoder_hidden)
        return ops.linear(torch.cat((out, emb), dim=-1), self.out.weight, self.out.bias).transpose(0, 1)

    def score_decoding_results(self, scores, tgt_sents_var, norm_by_word=False):
        B = scores.size(1)
        if tgt_sents_var.size(0) == B:
            tgt_sents_var = tgt_sents_var.contiguous().transpose(1, 0)
        logp = F.log_softmax(scores.reshape(-1, scores.size(2)), dim=-1)
        gold = tgt_sents_var[1:].contiguous().view(-1)
        tok = torch.gather(logp, 1, gold.unsqueeze(1)).squeeze(1) * (1.0 - torch.eq(gold, self.pad_id).float())
        tok = tok.view(-1, B).sum(dim=0)
        if norm_by_word:
            tok = tok.div(tgt_sents_var.transpose(1, 0).ne(self.pad_id).sum(dim=-1).float())
        return tok

    # ---- greedy generation ---------------------------------------------------------------------
    @torch.no_grad()
    def greedy_tokens(self, encoder_hidden, max_length=None):
        """always max_length steps from <s> (rnn_decoder.py:160-168,193-199) -> (int64 [max_len,B], h_T).
        The whole loop is one native call: per step embedding rows, 2 small GEMMs + fused cell per
        layer, projection fused with the arg-max; no host round trip per step."""
        max_length = self.max_length if max_length is None else max_length
        params = [self.rnn.layer_params(l) for l in range(self.n_layers)]
        if self.is_lstm:   # hidden is nn.LSTM's (h_0, c_0): the same native loop on the LSTM cell
            return ops.greedy_decode(self.embedding.weight, params, self.out.weight, self.out.bias, encoder_hidden[0],
                                     self.sos_id, max_length, c0=encoder_hidden[1])
        return ops.greedy_decode(self.embedding.weight, params, self.out.weight, self.out.bias, encoder_hidden,
                                 self.sos_id, max_length)

    def _greedy_lstm_stepwise(self, hidden, max_length):
        """the LSTM greedy loop enqueued step by step from the host (per step and layer two MFMA products + the fused
        cell kernel, then the projection fused with the arg-max): what the native loop replaced; kept as the
        second opinion of tests/test_generate_gpu.py"""
        h, c = [t.contiguous() for t in hidden[0].unbind(0)], [t.contiguous() for t in hidden[1].unbind(0)]
        B = h[0].size(0)
        tok = torch.full((B,), self.sos_id, device=h[0].device, dtype=torch.int64)
        params = [self.rnn.layer_params(l) for l in range(self.n_layers)]
        seq = []
        for _ in range(max_length):
            emb = ops.embed(self.embedding.weight, tok, None)
            x = emb
            for l, (wih, whh, bih, bhh) in enumerate(params):
                h[l], c[l] = ops.lstm_cell(ops.mm_nt(x, wih, bih), ops.mm_nt(h[l], whh, bhh), c[l])
                x = h[l]
            tok = ops.linear_argmax(x, emb, self.out.weight, self.out.bias)
            seq.append(tok)
        return torch.stack(seq, 0), (torch.stack(h, 0), torch.stack(c, 0))

    def forward(self, inputs=None, encoder_hidden=None, encoder_outputs=None, function=F.log_softmax,
                teacher_forcing_ratio=0.0):
        """Reference signature.  inputs=None -> greedy decode; the per-step [B,V] log-prob rows are NOT
        materialised (decoder_outputs is empty), callers use ret_dict['sequence'] / ['length']."""
        if inputs is not None and teacher_forcing_ratio > 0:
            logits = self.decode(inputs, encoder_hidden)
            outs = [function(logits[t], dim=-1) for t in range(logits.size(0))]
            seq = [o.topk(1)[1] for o in outs]
            return outs, None, {self.KEY_SEQUENCE: seq, self.KEY_LENGTH: self._lengths(torch.cat(seq, 1).t())}, \
                encoder_hidden
        if inputs is not None:
            raise NotImplementedError("free-running decode from a given first column: pass inputs=None")
        if encoder_hidden is None:
            raise ValueError("encoder_hidden is required")
        seq, hidden = self.greedy_tokens(encoder_hidden)
        ret = {self.KEY_SEQUENCE: [seq[t].unsqueeze(1) for t in range(seq.size(0))], self.KEY_LENGTH: self._lengths(seq)}
        return [], hidden, ret, encoder_hidden

    def _lengths(self, seq):
        """host bookkeeping of rnn_decoder.py:139-143: first </s> position + 1, else max length."""
        T = seq.size(0)
        hit = (seq == self.eos_id).cpu().numpy()
        first = np.where(hit.any(0), hit.argmax(0) + 1, T)
        return first.tolist()

    def init_state(self, encoder_hidden):
        return encoder_hidden

    def step(self, input_var, hidden, encoder_outputs=None, function=F.log_softmax):
        """the single-step method TopKDecoder.forward calls on its decoder (beam_decoder.py:148): forward_step"""
        return self.forward_step(input_var, hidden, encoder_outputs, function)

    def valid_args(self, inputs, encoder_hidden, encoder_outputs, function, teacher_forcing_ratio):
        """-> (inputs, batch_size, max_length), rnn_decoder.py:175-203"""
        if inputs is None and encoder_hidden is None:
            batch_size = 1
        elif inputs is not None:
            batch_size = inputs.size(0)
        else:
            batch_size = (encoder_hidden[0] if isinstance(encoder_hidden, tuple) else encoder_hidden).size(1)
        if inputs is None:
            if teacher_forcing_ratio > 0:
                raise ValueError("Teacher forcing has to be disabled (set 0) when no inputs is provided.")
            dev = ((encoder_hidden[0] if isinstance(encoder_hidden, tuple) else encoder_hidden).device
                   if encoder_hidden is not None else self.out.weight.device)
            inputs = torch.full((batch_size, 1), self.sos_id, dtype=torch.int64, device=dev)
            max_length = self.max_length
        else:
            max_length = inputs.size(1) - 1
        return inputs, batch_size, max_length

    @torch.no_grad()
    def step_rows(self, tok, h, c=None, out=None, function=F.log_softmax):
        """One generation step on the cell kernels, for searches that keep many rows alive (beam search):
        tok [N] int64; h (and c for the LSTM): per-layer lists of [N,H] -> (function(logits) [N,V], h', c').
        F.log_softmax rows are written into `out` [N,V] when given (projection + log-sum-exp in one kernel)."""
        emb = lookup(self.embedding, tok, GLOBAL_SEEDS)
        x, hn, cn = emb, [], []
        for l in range(self.n_layers):
            if l > 0 and self.training and self.rnn_droprate > 0.0:
                x = ops.dropout(x, self.rnn_droprate, GLOBAL_SEEDS.next())
            wih, whh, bih, bhh = self.rnn.layer_params(l)
            if self.is_lstm:
                x, cl = ops.lstm_cell(ops.mm_nt(x, wih, bih), ops.mm_nt(h[l], whh, bhh), c[l])
                cn.append(cl)
            else:
                x = ops.gru_cell(ops.mm_nt(x, wih, bih), ops.mm_nt(h[l], whh, bhh), h[l])
            hn.append(x)
        if function is F.log_softmax:
            logp = ops.linear_log_softmax(x, emb, self.out.weight, self.out.bias, out=out)
        else:
            logp = function(ops.mm_nt(torch.cat((x, emb), dim=-1), self.out.weight, self.out.bias), dim=-1)
            if out is not None:
                logp = out.copy_(logp)
        return logp, hn, (cn if self.is_lstm else None)


def prepare_job(job):
    """embedding rows of a scoring job, looked up NOW (records autograd under the current
    requires_grad flags: EnhanceSyntaxVAE freezes its adversaries, and with them the shared
    embeddings, between the main heads and the adversary heads)."""
    if "_emb" not in job:
        if job.get("no_grad"):
            with torch.no_grad():
                job["_emb"] = lookup(job["dec"].embedding, job["inputs"][:, :-1].contiguous(), GLOBAL_SEEDS)
        else:
            job["_emb"] = lookup(job["dec"].embedding, job["inputs"][:, :-1].contiguous(), GLOBAL_SEEDS)
    return job


# Measured and rejected as a default (round 2, same-box A/B at BASELINE configs[1]): 34.4 ms/step with the overlap
# against 33.4 without.  The tails do run beside the projections (their launches take 2x as long there), but a
# resident recurrence workgroup takes the place of one of the two GEMM workgroups of its CU, and the projections of
# the main decoder lose more (LSE 110 -> 85 TFLOP/s, dgrad / wgrad -10 %) than the hidden recurrence steps were worth.
OVERLAP_LONG_HEADS = os.environ.get("DSSVAE_OVERLAP", "0") == "1"


def score_many(jobs):
    """jobs: dicts {dec, inputs [B,T+1], targets [B,T+1], h0 [layers,B,H], norm_by_word, no_grad, uniform}.
    -> list of per-sentence log-probabilities [B], one per job.

    Heads of very different lengths (the linearised-tree decoders run 3x the steps of the sentence heads) are cut
    in time: steps [0, Ts) of EVERY head advance together in shared persistent launches on the current stream; the
    remaining steps of the long heads -- a recurrence over a third of the chip, latency-bound -- continue from the
    handed-over states on a side stream, BESIDE the vocabulary projections of the sentence heads (MFMA-bound).
    Autograd replays every op on the stream of its forward, so backward overlaps the same way.  Persistent
    launches themselves never overlap each other (ops._PersistentTurn)."""
    jobs = list(jobs)
    for j in jobs:
        prepare_job(j)
    steps = [j["inputs"].size(1) - 1 for j in jobs]
    ts = min(steps)
    long_ = [k for k, n in enumerate(steps) if n >= 2 * ts and n - ts >= 16]
    if (not OVERLAP_LONG_HEADS or not long_ or len(long_) == len(jobs) or not jobs[0]["inputs"].is_cuda or
            any(jobs[k].get("norm_by_word") for k in long_)):
        return _score_group(jobs)[0]
    # first part: every head, the long ones cut at ts (their embedding rows are split, not looked up again: the
    # lookup is already on the autograd tape under the requires_grad flags of its moment)
    first, tails = [], {}
    for k, j in enumerate(jobs):
        if k not in long_:
            first.append(j)
            continue
        head = dict(j, inputs=j["inputs"][:, :ts + 1], targets=j["targets"][:, :ts + 1],
                    _emb=j["_emb"][:, :ts].contiguous())
        tails[k] = dict(j, inputs=j["inputs"][:, ts:], targets=j["targets"][:, ts:], _emb=j["_emb"][:, ts:].contiguous())
        first.append(head)
    xs, embs, states = _recur(first)
    main = torch.cuda.current_stream()
    side = ops._side_stream(jobs[0]["inputs"].device, 7)
    side.wait_stream(main)  # i.e. for the recurrences above: the projections of this stream are enqueued below
    tail_jobs = []
    for k in long_:
        t = tails[k]
        t.pop("h0", None)
        t["h0_layers"] = states[k]
        for x in [t["_emb"], t["inputs"], t["targets"]] + list(states[k]):
            x.record_stream(side)  # produced on this stream, read on the side stream
        tail_jobs.append(t)
    with torch.cuda.stream(side):
        rest, _ = _score_group(tail_jobs)
    res = _project(first, xs, embs)
    main.wait_stream(side)
    for k, r in zip(long_, rest):
        r.record_stream(main)
        res[k] = res[k] + r
    return res


def _score_group(jobs, want_states=False):
    """recurrences, then projections, of one group of jobs on the current stream
    -> (per-sentence log-probabilities, per job the list of final states per layer if want_states)"""
    xs, embs, finals = _recur(jobs)
    return _project(jobs, xs, embs), (finals if want_states else None)


def _recur(jobs):
    """All jobs advance layer by layer together, so their recurrences share persistent GRU launches; a `no_grad`
    job (an adversary that is only evaluated) is forward-only inside the shared launches.  A job carries its initial
    states as `h0` [layers,B,H] or as `h0_layers` (a list: the final states another group handed over).
    -> (top-layer outputs per job, embedding rows per job, per job the list of final states per layer)"""
    n_layers = jobs[0]["dec"].n_layers
    assert all(j["dec"].n_layers == n_layers for j in jobs)
    grad_on = torch.is_grad_enabled()
    embs, xs = [], []
    for j in jobs:
        e = prepare_job(j)["_emb"]
        embs.append(e)
        xs.append(e)
    # per-layer initial states: ONE unbind per head (its backward is one stack) instead of a select per layer
    # (whose backward is a zero-fill + copy per layer and an add to join them)
    h0_layers = [j["h0_layers"] if "h0_layers" in j else j["h0"].unbind(0) for j in jobs]
    finals = [[] for _ in jobs]
    for layer in range(n_layers):
        insts, proj = [], {True: [], False: []}
        for k, j in enumerate(jobs):
            dec, x = j["dec"], xs[k]
            ng = bool(j.get("no_grad")) or not grad_on
            with torch.set_grad_enabled(not ng):
                if layer > 0 and dec.training and dec.rnn_droprate > 0.0:
                    x = ops.dropout(x, dec.rnn_droprate, GLOBAL_SEEDS.next())
            wih, whh, bih, bhh = dec.rnn.layer_params(layer)
            h0 = h0_layers[k][layer]
            if ng:
                whh, bhh, h0 = whh.detach(), bhh.detach(), h0.detach()
            proj[ng].append((k, (x, wih, bih.detach())))  # b_ih's gradient comes out of the recurrence's backward
            insts.append(dict(whh=whh, bhh=bhh, bih=None if ng else bih, h0=h0, group=k, col=0, save=not ng))
        # the input projections of all heads of the layer share grouped MFMA launches (evaluation-only
        # heads in a group of their own, outside autograd)
        for ng, items in proj.items():
            if items:
                with torch.set_grad_enabled(not ng):
                    for (k, _), xp in zip(items, ops.linear_many([t for _, t in items])):
                        insts[k]["xproj"] = xp
        xs, hl = ops.gru_layer(insts)
        for k in range(len(jobs)):
            finals[k].append(hl[k])
    return xs, embs, finals


def _project(jobs, xs, embs):
    """fused vocabulary projection + cross-entropy of every job's top-layer outputs -> per-sentence log-probs"""
    grad_on = torch.is_grad_enabled()
    res = []
    for k, j in enumerate(jobs):
        dec = j["dec"]
        B, T, H = xs[k].shape
        gold = j["targets"][:, 1:].reshape(-1)
        keep = getattr(j["targets"], "_keep_rows", None)  # ragged batch: rows whose target is not <pad>
        with torch.set_grad_enabled(grad_on and not j.get("no_grad")):
            xh, xe = xs[k].reshape(B * T, H), embs[k].reshape(B * T, -1)
            uni = bool(j.get("uniform")) and not j.get("norm_by_word")
            if keep is None:
                # the T rows of a sentence are summed by the loss kernel itself
                logp = -ops.linear_ce(xh, xe, dec.out.weight, dec.out.bias, gold, dec.pad_id, uniform_upstream=uni,
                                      group=T)
            else:
                # pad rows add nothing to the loss or to any gradient: leave them out of the projection
                part = ops.linear_ce(xh.index_select(0, keep), xe.index_select(0, keep), dec.out.weight,
                                     dec.out.bias, gold.index_select(0, keep), dec.pad_id, uniform_upstream=uni)
                nll = torch.zeros(B * T, device=part.device, dtype=part.dtype).index_copy(0, keep, part)
                logp = -nll.view(B, T).sum(1)
            if j.get("norm_by_word"):
                logp = logp.div(j["targets"].ne(dec.pad_id).sum(dim=-1).float())
        res.append(logp)
    return res
