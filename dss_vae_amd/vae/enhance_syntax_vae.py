"""EnhanceSyntaxVAE, registry key 'SyntaxVAE2' (reference: dss_vae/vae/enhance_syntax_vae.py:47-637).
Scalar head losses; adversary parameters are frozen through requires_grad (the reference's
detach()/reload(), :573-579) so gradient reaches the latents but not the adversaries; the -adv term
enters 'Loss' only when train_iter > warm_up (:333)."""
import torch

from .. import ops
from ..decoder.rnn_decoder import prepare_job, score_many
from .syntax_vae import BridgeMLP, BridgeRNN, TwoEncoderVAE


class RNNPredictor(BridgeRNN):
    def forward(self, hidden, tgt_var, div_by_word=False):
        job = self.job(hidden, tgt_var)
        job["norm_by_word_len"] = div_by_word
        job["uniform"] = not div_by_word
        return -_finish(job, score_many([job])[0]).sum()

    def detach(self):
        for p in self.parameters():
            p.requires_grad = False

    def reload(self):
        for p in self.parameters():
            p.requires_grad = True


def _finish(job, logp):
    if job.get("norm_by_word_len"):
        logp = logp.div(job["targets"].ne(job["dec"].pad_id).sum(-1).float() + 1e-9)
    return logp


class _Row(dict):
    """a linear combination {component index: coefficient} of the assembly's component sums.  A key ("d", k) is the
    part of component k's coefficient that is multiplied by the step's KL weight ON THE DEVICE (`dyn`, an ops.Dyn:
    only while a step is being captured into a HIP graph; otherwise the weight is an ordinary factor)."""
    dyn = None

    def __add__(self, other):
        out = _Row(self)
        out.dyn = self.dyn if self.dyn is not None else other.dyn
        for k, c in other.items():
            out[k] = out.get(k, 0.0) + c
        return out

    def __mul__(self, c):
        if isinstance(c, ops.Dyn) and ops.DYN is not None:
            assert not any(isinstance(k, tuple) for k in self), "one per-step factor per term"
            out = _Row({("d", k): v for k, v in self.items()})
            out.dyn = c
            return out
        out = _Row({k: v * c for k, v in self.items()})
        out.dyn = self.dyn
        return out


class _Assembly:
    """The loss dictionary as rows of coefficients over the sums of a few tensors (per-sentence log-likelihoods,
    bag-of-words losses, KL terms); `run` evaluates every row in one launch (ops.weighted_sums)."""

    def __init__(self):
        self.parts, self.rows = [], {}

    def term(self, signed, weight):
        """weight * sign * sum(tensor) as a row; `signed` = (sign, tensor), or None for a head that is switched off"""
        if signed is None:
            return _Row()
        self.parts.append(signed[1])
        return _Row({len(self.parts) - 1: signed[0] * weight})

    def run(self, rows=None):
        rows = self.rows if rows is None else rows
        names = list(rows)
        ks = range(len(self.parts))
        dyn = next((rows[n].dyn for n in names if rows[n].dyn is not None), None)
        coef_dyn = None if dyn is None else [[rows[n].get(("d", k), 0.0) for k in ks] for n in names]
        vals = ops.weighted_sums([[rows[n].get(k, 0.0) for k in ks] for n in names], self.parts, coef_dyn, dyn)
        return dict(zip(names, vals))


class MLPPredictor(BridgeMLP):
    def forward(self, hidden, tgt_var, div_by_word=False):
        loss = self.per_sentence(hidden, tgt_var, uniform=not div_by_word)
        if div_by_word:
            loss = loss.div(1e-9 + tgt_var.ne(self.pad_id).sum(-1).float())
        return loss.sum()

    def detach(self):
        for p in self.parameters():
            p.requires_grad = False

    def reload(self):
        for p in self.parameters():
            p.requires_grad = True


class EnhanceSyntaxVAE(TwoEncoderVAE):
    GRAPH_SAFE = True
    ENC_NAMES = ("sem_encoder", "syn_encoder")
    ADVERSARIES = ("sem_to_sent", "syn_to_sent", "syn_to_sem", "sem_to_syn")

    def __init__(self, args, vocab, src_embed=None, tgt_embed=None):
        super().__init__(args, vocab, name="Syntax VAE with 2 seperate encoder")
        self._build_common(args, vocab, src_embed, tgt_embed)
        self.pad_idx = vocab.src.pad_id
        te, td = self.task_enc_dim, self.task_dec_dim
        share = args.share_embed
        self.mul_syn = RNNPredictor(args, vocab, te, td, tgt_embed, "tgt")
        self.mul_sem = MLPPredictor(args, vocab, te, td)
        self.sem_to_syn = RNNPredictor(args, vocab, te, td, self.tgt_embed if share else None, "tgt")
        self.syn_to_sem = MLPPredictor(args, vocab, te, td)
        self.sem_to_sent = RNNPredictor(args, vocab, te, td, self.src_embed if share else None, "src")
        self.syn_to_sent = RNNPredictor(args, vocab, te, td, self.src_embed if share else None, "src")

    def _adv_jobs(self, syn_h, sem_h, syn_tgt, sem_tgt):
        a = self.args
        jobs, names = [], []
        if a.sem_to_syn > 0.0:
            jobs.append(self.sem_to_syn.job(sem_h, syn_tgt)); names.append("sem_to_syn")
        if a.sem_to_sent > 0.0:
            jobs.append(self.sem_to_sent.job(sem_h, sem_tgt)); names.append("sem_to_sent")
        if a.syn_to_sent > 0.0:
            jobs.append(self.syn_to_sent.job(syn_h, sem_tgt)); names.append("syn_to_sent")
        return jobs, names

    def _dis_terms(self, syn_h, sem_h, syn_tgt, sem_tgt, extra_jobs=(), raw=False):
        """adversary terms; `extra_jobs` (already prepared main-head jobs) ride in the same persistent
        GRU launches, their log-probs are returned second.  raw: the terms stay (sign, per-sentence vector) pairs
        -- None for a head that is switched off -- for the fused loss assembly (`_Assembly`)."""
        a = self.args
        jobs, names = self._adv_jobs(syn_h, sem_h, syn_tgt, sem_tgt)
        for j in jobs:
            prepare_job(j)
        res = score_many(list(extra_jobs) + jobs) if (jobs or extra_jobs) else []
        ne = len(extra_jobs)
        if raw:
            d = {"sem_to_syn": None, "sem_to_sent": None, "syn_to_sent": None,
                 "syn_to_sem": (1.0, self.syn_to_sem.per_sentence(syn_h, sem_tgt)) if a.syn_to_sem > 0.0 else None}
            for n, r in zip(names, res[ne:]):
                d[n] = (-1.0, r)
            return d, res[:ne]
        d = {"sem_to_syn": 0.0, "sem_to_sent": 0.0, "syn_to_sent": 0.0,
             "syn_to_sem": self.syn_to_sem.forward(hidden=syn_h, tgt_var=sem_tgt) if a.syn_to_sem > 0.0 else 0.0}
        for n, r in zip(names, res[ne:]):
            d[n] = -r.sum()
        return d, res[:ne]

    # ---- the reference's helper surface (enhance_syntax_vae.py:338-403); see SyntaxVAE for the why
    def _dis_loss(self, syn_hid, sem_hid, syn_tgt, sem_tgt, norm_by_word=False):
        if norm_by_word:
            a = self.args
            return {"sem_to_syn": self.sem_to_syn.forward(sem_hid, syn_tgt, div_by_word=True) if a.sem_to_syn > 0.0 else 0.0,
                    "syn_to_sem": self.syn_to_sem.forward(hidden=syn_hid, tgt_var=sem_tgt) if a.syn_to_sem > 0.0 else 0.0,
                    "sem_to_sent": self.sem_to_sent.forward(sem_hid, sem_tgt, div_by_word=True) if a.sem_to_sent > 0.0 else 0.0,
                    "syn_to_sent": self.syn_to_sent.forward(syn_hid, sem_tgt, div_by_word=True) if a.syn_to_sent > 0.0 else 0.0}
        return self._dis_terms(syn_hid, sem_hid, syn_tgt, sem_tgt)[0]

    def get_dis_loss(self, syn_hid, sem_hid, syn_tgt, sem_tgt, norm_by_word=False):
        for n in self.ADVERSARIES:
            getattr(self, n).reload()
        d = self._dis_loss(syn_hid.detach(), sem_hid.detach(), syn_tgt, sem_tgt, norm_by_word)
        if self.args.infer_weight > 0.0:
            return d["syn_to_sem"] + d["syn_to_sent"], d["sem_to_syn"] + d["sem_to_sent"]
        return d["syn_to_sem"], d["sem_to_syn"]

    def get_adv_loss(self, syn_hid, sem_hid, syn_tgt, sem_tgt, norm_by_word=False):
        a = self.args
        if self.training:
            for n in self.ADVERSARIES:
                getattr(self, n).detach()
        d = self._dis_loss(syn_hid, sem_hid, syn_tgt, sem_tgt, norm_by_word)
        if a.infer_weight > 0.0:
            return (a.syn_to_sem * d["syn_to_sem"] + a.infer_weight * a.syn_to_sent * d["syn_to_sent"],
                    a.sem_to_syn * d["sem_to_syn"] + a.infer_weight * a.sem_to_sent * d["sem_to_sent"])
        return a.syn_to_sem * d["syn_to_sem"], a.sem_to_syn * d["sem_to_syn"]

    def get_mul_loss(self, syntax_hidden, semantic_hidden, syn_tgt, sem_tgt, norm_by_word=False):
        syn_loss = self.mul_syn.forward(hidden=syntax_hidden, tgt_var=syn_tgt, div_by_word=norm_by_word)
        sem_loss = self.mul_sem.forward(hidden=semantic_hidden, tgt_var=sem_tgt)
        return self.args.mul_syn * syn_loss, self.args.mul_sem * sem_loss

    def forward(self, examples, is_dis=False, norm_by_word=False, defer_assembly=False):
        batch, ret = self._front(examples)
        tgt_var, syn_var = batch.tgt_var, batch.syn_var
        syn_h, sem_h = ret["syn_hidden"], ret["sem_hidden"]
        a = self.args
        if is_dis:
            for n in self.ADVERSARIES:
                getattr(self, n).reload()
            d, _ = self._dis_terms(syn_h.detach(), sem_h.detach(), syn_var, tgt_var)
            if a.infer_weight > 0.0:
                ret["dis syn"] = d["syn_to_sem"] + d["syn_to_sent"]
                ret["dis sem"] = d["sem_to_syn"] + d["sem_to_sent"]
            else:
                ret["dis syn"], ret["dis sem"] = d["syn_to_sem"], d["sem_to_syn"]
            return ret
        # same order as the reference: the main decoder + multitask heads read their (shared)
        # embeddings first, THEN the adversaries are frozen (this also flips requires_grad of the
        # shared src/tgt embeddings, a reference quirk pinned by the golden 'nograd' lists), then the
        # adversaries read theirs.  The recurrences of all five sequence heads then share launches.
        main_jobs = [prepare_job(self._main_job(ret, tgt_var)), prepare_job(self.mul_syn.job(syn_h, syn_var))]
        bow = self.mul_sem.per_sentence(sem_h, tgt_var)
        if self.training:
            for n in self.ADVERSARIES:
                getattr(self, n).detach()
        d, (rec, tree) = self._dis_terms(syn_h, sem_h, syn_var, tgt_var, extra_jobs=main_jobs, raw=True)
        # Everything from here to the loss dictionary is linear in the sums of these vectors: ONE launch forms all of
        # it (ops.weighted_sums) instead of ~40 one-element torch ops forward and ~25 backward.
        asm = _Assembly()
        iw = a.infer_weight
        adv = asm.term(d["syn_to_sem"], a.syn_to_sem) + asm.term(d["sem_to_syn"], a.sem_to_syn)
        if iw > 0.0:
            adv = adv + asm.term(d["syn_to_sent"], iw * a.syn_to_sent) + asm.term(d["sem_to_sent"], iw * a.sem_to_sent)
        syn_loss = asm.term((-1.0, tree), a.mul_syn)
        sem_loss = asm.term((1.0, bow), a.mul_sem)
        asm.rows.update({"adv": adv, "mul": syn_loss + sem_loss, "nll_loss": asm.term((-1.0, rec), 1.0),
                         "sem_loss": sem_loss, "syn_loss": syn_loss})
        ret["batch_size"] = len(batch)
        if defer_assembly:
            ret["assembly"] = asm
        else:
            ret.update(asm.run())
        return ret

    def get_loss(self, examples, train_iter, is_dis=False, **kwargs):
        a = self.args
        sc = self.step_scalars(train_iter)
        self.step_unk_rate = ops.dyn_value("unk_rate", sc["unk_rate"])
        explore = self.forward(examples, is_dis, norm_by_word=False, defer_assembly=True)
        if is_dis:
            return explore
        asm, B = explore["assembly"], float(explore["batch_size"])
        kl_weight = ops.dyn_value("kl_weight", sc["kl_weight"])
        sem_kl, syn_kl = asm.term((1.0, explore["sem_kl"]), 1.0), asm.term((1.0, explore["syn_kl"]), 1.0)
        r = asm.rows
        kl_loss = (sem_kl * a.sem_weight + syn_kl * a.syn_weight) * (1.0 / (a.sem_weight + a.syn_weight)) * (1.0 / B)
        mul_loss, adv_loss, nll_loss = r["mul"] * (1.0 / B), r["adv"] * (1.0 / B), r["nll_loss"] * (1.0 / B)
        kl_item = kl_loss * kl_weight
        loss = kl_item + nll_loss + mul_loss
        if train_iter > a.warm_up:
            loss = loss + adv_loss * -1.0
        out = asm.run({"KL Loss": kl_loss, "NLL Loss": nll_loss, "MUL Loss": mul_loss, "ADV Loss": adv_loss,
                       "KL Item": kl_item, "Model Score": kl_loss + nll_loss, "ELBO": kl_item + nll_loss, "Loss": loss,
                       "SYN KL Loss": syn_kl * (1.0 / B), "SEM KL Loss": sem_kl * (1.0 / B)})
        out["KL Weight"] = float(kl_weight)
        return out

    def eval_syntax(self, examples):
        ret = self.latent_for_init(self.hidden_to_latent(self.encode_to_hidden(examples, need_sort=True), False))
        return self.predict_syntax(hidden=ret["syn_hidden"], predictor=self.mul_syn)

    def eval_adv(self, sem_in, syn_ref):
        return self._transfer(sem_in, syn_ref, self.mul_syn)
