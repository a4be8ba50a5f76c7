"""Beam search over an RNNDecoder (reference: dss_vae/decoder/beam_decoder.py:47-350, TopKDecoder).

Same constructor, `beam_search` / `forward` signatures, return triple and metadata keys as the reference.  The
schedule is not the reference's:

  * the `dmts` steps are enqueued back to back with no device->host traffic: per step the embedding rows, per layer
    two MFMA products + the fused cell kernel, the vocabulary projection with its log-sum-exp written straight into
    that step's rows of ONE [T, B*k, V] buffer (ops.linear_log_softmax), torch.topk over each sentence's k*V
    candidates, and the reordered states written straight into one [layers, T, B*k, H] buffer;
  * the back-pointers, symbols and step scores ([T, B*k] each, a few KB) then go to the host ONCE; the reference's
    backtracking (beam_decoder.py:236-341: one .nonzero() sync and a Python loop with tensor indexing per step) is
    run there on numpy as an index plan (`backtrack_plan`), and the big tensors are gathered on the device with one
    index_select each.

Faithful to the reference, deliberately, in what its results depend on:
  * `_inflate` tiles ([s0 s1 .. s0 s1 ..], tensor.repeat) while the scores / back-pointers are sentence-major
    (row = sentence * k + beam, beam_decoder.py:113,131-134): for B > 1 beam rows start from OTHER sentences' states.
    The reference's one call site decodes one sentence (models/seq2seq.py:203-216), where the two agree.  Set
    `sentence_major = True` on the instance to inflate as the indexing assumes instead;
  * stored outputs of step t are in the beam order of step t-1 but backtracked with step t's pointers
    (beam_decoder.py:153 / 266); `h_n` stays zero for beams that never emitted </s> (beam_decoder.py:252-256);
    metadata['sequence'][t] is sentence 0's k beams (beam_decoder.py:217).
Not mirrored: the graph for "local backprop" (the search runs under no_grad; nothing in the reference differentiates
through it) and attention (`use_attention` is forced off on the VAE path)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def _inflate(tensor, times, dim):
    """replicate the tensor `times` times along `dim`, whole-tensor tiling (beam_decoder.py:10-44)"""
    repeat_dims = [1] * tensor.dim()
    repeat_dims[dim] = times
    return tensor.repeat(*repeat_dims)


def _topk_desc(x):
    """indices of each row's entries in descending order (ties: lower index first)"""
    return np.argsort(-x, axis=1, kind="stable")


def backtrack_plan(predecessors, symbols, scores, b, k, eos):
    """Host restatement of TopKDecoder._backtrack (beam_decoder.py:236-341) as an INDEX PLAN over what the search
    stored: predecessors / symbols int [T, b*k], scores float32 [T, b*k] (numpy).  Returns a dict:
      src     int64 [T, b*k]  row of step t's stored tensors that ends up at (sentence, rank) = divmod(column, k)
      hn_t, hn_row  int64 [b*k]  step and row of the state `h_n` takes for that place; -1: none (stays zero)
      score   float32 [b, k]; length  list[b][k]
    An ended hypothesis (its row emitted </s> at step t) replaces, from the last place of its sentence upwards, a
    hypothesis that is still running at the horizon; places are sorted by score again at the end."""
    T, bk = symbols.shape
    assert bk == b * k and predecessors.shape == (T, bk) and scores.shape == (T, bk)
    pos = (np.arange(b, dtype=np.int64) * k)[:, None]
    last = scores[T - 1].reshape(b, k)
    order = _topk_desc(last)
    ret_scores = np.take_along_axis(last, order, axis=1).copy()
    lengths = [[T] * k for _ in range(b)]
    found = [0] * b
    tp = (order + pos).reshape(bk)
    src = np.empty((T, bk), dtype=np.int64)
    hn_t = np.full(bk, -1, dtype=np.int64)
    hn_row = np.full(bk, -1, dtype=np.int64)
    for t in range(T - 1, -1, -1):
        cur = tp.copy()
        tp = predecessors[t][cur]
        for idx in np.nonzero(symbols[t] == eos)[0][::-1]:
            bi = int(idx) // k
            rk = k - (found[bi] % k) - 1
            found[bi] += 1
            r = bi * k + rk
            tp[r] = predecessors[t][idx]
            cur[r] = idx
            hn_t[r], hn_row[r] = t, idx
            ret_scores[bi, rk] = scores[t][idx]
            lengths[bi][rk] = t + 1
        src[t] = cur
    order = _topk_desc(ret_scores)
    ret_scores = np.take_along_axis(ret_scores, order, axis=1)
    lengths = [[lengths[bi][int(j)] for j in order[bi]] for bi in range(b)]
    flat = (order + pos).reshape(bk)
    return {"src": np.ascontiguousarray(src[:, flat]), "hn_t": hn_t[flat], "hn_row": hn_row[flat],
            "score": ret_scores, "length": lengths}


class TopKDecoder(nn.Module):
    """Top-k decoding with beam search.  decoder_rnn: an RNNDecoder (gru or lstm); k: beam size."""

    def __init__(self, decoder_rnn, k):
        super().__init__()
        self.rnn = decoder_rnn
        self.k = k
        self.hidden_size = self.rnn.hidden_size
        self.V = self.rnn.vocab_size
        self.SOS = self.rnn.sos_id
        self.EOS = self.rnn.eos_id
        self.sentence_major = False

    def beam_search(self, encoder_hidden, encoder_outputs, decode_max_time_step, beam_size=5):
        self.k = beam_size
        _, _, metadata = self.forward(inputs=None, encoder_hidden=encoder_hidden, encoder_outputs=encoder_outputs,
                                      teacher_forcing_ratio=0.0, dmts=decode_max_time_step)
        return metadata

    @torch.no_grad()
    def forward(self, inputs=None, encoder_hidden=None, encoder_outputs=None, function=F.log_softmax,
                teacher_forcing_ratio=0.0, retain_output_probs=True, dmts=None):
        inputs, B, T = self.rnn.valid_args(inputs, encoder_hidden, encoder_outputs, function, teacher_forcing_ratio)
        if teacher_forcing_ratio <= 0.0 and dmts is not None:
            T = dmts
        if encoder_hidden is None:
            raise ValueError("encoder_hidden is required")
        k, V, rnn = self.k, self.V, self.rnn
        lstm = isinstance(encoder_hidden, tuple)
        Bk = B * k
        state = rnn.init_state(encoder_hidden)
        dev = (state[0] if lstm else state).device

        def inflate(x):  # [layers, B, H] -> per-layer [B*k, H]
            x = x.float()
            x = x.repeat_interleave(k, dim=1) if self.sentence_major else _inflate(x, k, 1)
            return [t.contiguous() for t in x.unbind(0)]

        h = inflate(state[0] if lstm else state)
        c = inflate(state[1]) if lstm else None
        n_layers, H = len(h), h[0].size(1)
        pos = (torch.arange(B, device=dev, dtype=torch.int64) * k).view(B, 1)
        # the inflated copies start at -inf so that the first step's top k are k different words
        seq_scores = torch.full((B, k), -float("inf"), device=dev, dtype=torch.float32)
        seq_scores[:, 0] = 0.0
        seq_scores = seq_scores.view(Bk)
        tok = torch.full((Bk,), self.SOS, device=dev, dtype=torch.int64)
        outputs = torch.empty(T, Bk, V, device=dev, dtype=torch.float32)
        hid = torch.empty(n_layers, T, Bk, H, device=dev, dtype=torch.float32)
        cid = torch.empty_like(hid) if lstm else None
        book = torch.empty(2, T, Bk, device=dev, dtype=torch.int64)  # back-pointers, symbols
        step_scores = torch.empty(T, Bk, device=dev, dtype=torch.float32)
        for t in range(T):
            logp, h, c = rnn.step_rows(tok, h, c, out=outputs[t], function=function)
            cand = (seq_scores.unsqueeze(1) + logp).view(B, k * V)
            scores, candidates = cand.topk(k, dim=1)
            tok = torch.remainder(candidates, V).view(Bk)
            pred = (torch.div(candidates, V, rounding_mode="floor") + pos).view(Bk)
            for l in range(n_layers):
                torch.index_select(h[l], 0, pred, out=hid[l, t])
                h[l] = hid[l, t]
                if lstm:
                    torch.index_select(c[l], 0, pred, out=cid[l, t])
                    c[l] = cid[l, t]
            step_scores[t] = scores.view(Bk)
            book[0, t], book[1, t] = pred, tok
            seq_scores = scores.view(Bk).masked_fill(tok == self.EOS, -float("inf"))

        # the only device->host traffic of the search: [T, B*k] back-pointers, symbols, scores
        book_h = book.cpu().numpy()
        plan = backtrack_plan(book_h[0], book_h[1], step_scores.cpu().numpy(), B, k, self.EOS)
        steps = torch.arange(T, dtype=torch.int64).view(T, 1) * Bk
        flat = (torch.from_numpy(plan["src"]) + steps).reshape(-1).to(dev, non_blocking=True)
        has = torch.from_numpy(plan["hn_t"] >= 0)
        hn_flat = torch.from_numpy(np.where(plan["hn_t"] >= 0, plan["hn_t"] * Bk + plan["hn_row"], 0)).to(dev)
        has = has.to(dev).view(1, Bk, 1)

        output = list(outputs.view(T * Bk, V).index_select(0, flat).view(T, B, k, V).unbind(0))
        ret_p = list(book[1].reshape(T * Bk).index_select(0, flat).view(T, B, k, 1).unbind(0))

        def gather_states(buf):  # [layers, T, B*k, H] -> ([T] x [layers, B, k, H], h_n [layers, B, k, H])
            rows = buf.view(n_layers, T * Bk, H)
            h_t = rows.index_select(1, flat).view(n_layers, T, B, k, H).transpose(0, 1).contiguous()
            h_n = torch.where(has, rows.index_select(1, hn_flat), torch.zeros((), device=dev)).view(n_layers, B, k, H)
            return list(h_t.unbind(0)), h_n

        h_t, h_n = gather_states(hid)
        if lstm:
            c_t, c_n = gather_states(cid)
            h_t = [(a, b_) for a, b_ in zip(h_t, c_t)]
            h_n = (h_n, c_n)
        lengths = plan["length"]
        decoder_outputs = [step[:, 0, :] for step in output]
        decoder_hidden = tuple(x[:, :, 0, :] for x in h_n) if lstm else h_n[:, :, 0, :]
        metadata = {"inputs": inputs, "output": output if retain_output_probs else [], "h_t": h_t,
                    "score": torch.from_numpy(plan["score"]).to(dev), "topk_length": lengths, "topk_sequence": ret_p,
                    "length": [seq_len[0] for seq_len in lengths], "sequence": [seq[0] for seq in ret_p]}
        return decoder_outputs, decoder_hidden, metadata
