"""Test-time evaluators around the GPU decode path (reference: dss_vae/metrics/vae_metrics.py:22-268):
reconstruction (+ loss terms), prior generation statistics, paraphrase (semantic latent kept, syntax sampled /
dropped), controlled generation and syntax transfer over `sentence \\t reference` files.  Output file names follow
the reference (`<data>.<desc>.pred/.tgt/.ori/.gold/.input/.ss/.st/.sp`, `<desc>.score`).

Three lines of the reference do not run as written and are taken as meant: `Dataset.from_raw_file(..., e_type=
'plain')` (its table only has "Plain"), `self.vae.genewwrating` (generating), and the unigram statistic (see
tools.unigram_distribution)."""
import os

import torch

from ..structs import Dataset
from ..utils.model_utils import update_tracker
from . import bleu
from .evaluation import evaluate
from .tools import (avg_length, con_to_plain, get_bleu_score, predict_to_plain, recovery_ref, split_examples,
                    to_plain, unigram_divergence, write_append_result, write_result)


class VaeEvaluator(object):
    def __init__(self, model, out_dir, batch_size=10, train_batch_size=32, **kwargs):
        self.vae, self.out_dir = model, out_dir
        self.eval_batch_size, self.train_batch_size = batch_size, train_batch_size
        os.makedirs(out_dir, exist_ok=True)

    def _path(self, *parts):
        return os.path.join(self.out_dir, ".".join(parts))


class SyntaxVaeEvaluator(VaeEvaluator):
    def __call__(self, *args, **kwargs):
        pass

    def _eval_mode(self):
        was = self.vae.training
        self.vae.eval()
        return was

    def _pair_sets(self, eval_dir, eval_list):
        return [Dataset.from_raw_file(os.path.join(eval_dir, f), e_type="Plain") for f in eval_list]

    def evaluate_reconstruction(self, examples, eval_desc, eval_step=None, write_down=True):
        was = self._eval_mode()
        step = eval_step if eval_step is not None else 2 * self.vae.args.x0
        res = evaluate(examples, self.vae, sort_key="src", eval_tgt="src", batch_size=self.eval_batch_size,
                       out_dir=self._path(eval_desc, "files"))
        track = {}
        with torch.no_grad():
            for batch in Dataset(examples).batch_iter(batch_size=self.train_batch_size, shuffle=False):
                track = update_tracker(self.vae.get_loss(batch, step), track)
        if write_down:
            write_result(predict_to_plain(res["predict"]), fname=self._path(eval_desc, "pred"))
            # (the reference formats this file with reference_to_plain, one nesting level too deep for what
            # prepare_ref returns: it writes the characters of each first word; the gold sentences are meant)
            write_result(predict_to_plain(res["reference"]), fname=self._path(eval_desc, "gold"))
            write_result(track, fname=self._path(eval_desc, "score"))
            with open(self._path(eval_desc, "score"), "a") as f:
                f.write("{}:{}".format(getattr(self.vae.args, "eval_mode", "BLEU"), res["accuracy"]))
        self.vae.train(was)
        return track, res

    def evaluate_generation(self, corpus_examples=None, sample_size=50, eval_desc="gen", write_down=True):
        was = self._eval_mode()
        train = [e.src for e in corpus_examples] if corpus_examples is not None else None
        with torch.no_grad():
            sample_corpus = to_plain(self.vae.generating(sample_size, batch_size=self.eval_batch_size))
        write_result(docs=sample_corpus, fname=self._path(eval_desc, "text"))
        res = None
        if write_down:
            res = {"UnigramKL": unigram_divergence(train, sample_corpus, self.vae.vocab.src) if train else None,
                   "Gold AvgLen": avg_length(train) if train else None, "Gen AvgLen": avg_length(sample_corpus)}
            write_result(docs=res, fname=self._path(eval_desc, "score"))
        print("write result to {}".format(self._path(eval_desc, "text")))
        self.vae.train(was)
        return res

    def _paraphrase(self, eval_dir, eval_list, eval_desc, condition):
        was = self.vae.training  # the reference leaves the mode alone here: sampling follows model.training
        scores = []
        for data, name in zip(self._pair_sets(eval_dir, eval_list), eval_list):
            ori_ex, tgt_ex, pred = [], [], []
            with torch.no_grad():
                for batch in data.batch_iter(batch_size=self.eval_batch_size, shuffle=False):
                    ori, tgt = split_examples(batch)
                    ori_ex.extend(ori)
                    tgt_ex.extend(tgt)
                    pred.extend(self.vae.conditional_generating(condition=condition, examples=ori)["res"])
            ori_ref = recovery_ref(ori_ex, self.vae.vocab.src, keep_origin=True)
            tgt_ref = recovery_ref(tgt_ex, self.vae.vocab.src, keep_origin=True)
            ori_bleu, tgt_bleu = get_bleu_score(ori_ref, pred), get_bleu_score(tgt_ref, pred)
            write_result(predict_to_plain(pred), fname=self._path(name, eval_desc, "pred"))
            write_result(con_to_plain(tgt_ref), fname=self._path(name, eval_desc, "tgt"))
            write_result(con_to_plain(ori_ref), fname=self._path(name, eval_desc, "ori"))
            print("finish {} with ori bleu:{}, tgt bleu:{} !".format(name, ori_bleu, tgt_bleu))
            scores.append([ori_bleu, tgt_bleu])
        write_append_result(docs=["ori bleu:{}, tgt bleu:{} !".format(*s) for s in scores],
                            fname=self._path(eval_desc, "score"))
        self.vae.train(was)
        return scores

    def evaluate_para(self, eval_dir, eval_list, eval_desc="para"):
        return self._paraphrase(eval_dir, eval_list, eval_desc, "sem")

    def evaluate_pure_para(self, eval_dir, eval_list, eval_desc="pure-para"):
        return self._paraphrase(eval_dir, eval_list, eval_desc, "sem-only")

    def _transfer_pass(self, data):
        """eval_adv over a pair set -> (semantic-side examples, syntax-side examples, merged result dict)"""
        sem_ex, syn_ex, out = [], [], {"res": [], "ori syn": [], "ref syn": []}
        with torch.no_grad():
            for batch in data.batch_iter(batch_size=self.eval_batch_size, shuffle=False):
                sem, syn = split_examples(batch)
                sem_ex.extend(sem)
                syn_ex.extend(syn)
                ret = self.vae.eval_adv(sem, syn)
                for k in out:
                    out[k].extend(ret.get(k, []))
        return sem_ex, syn_ex, out

    def evaluate_control(self, eval_dir, eval_list, eval_desc="con"):
        was = self._eval_mode()
        scores = []
        for data, name in zip(self._pair_sets(eval_dir, eval_list), eval_list):
            sem_ex, syn_ex, out = self._transfer_pass(data)
            sem_e = recovery_ref(sem_ex, self.vae.vocab.src, keep_origin=True)
            syn_e = recovery_ref(syn_ex, self.vae.vocab.src, keep_origin=True)
            tgt_bleu = get_bleu_score(references=syn_e, hypothesis=out["res"])
            write_result(predict_to_plain(out["res"]), fname=self._path(name, eval_desc, "pred"))
            write_result(con_to_plain(syn_e), fname=self._path(name, eval_desc, "gold"))
            write_result(con_to_plain(sem_e), fname=self._path(name, eval_desc, "input"))
            print("finish {} with bleu:{} !".format(name, tgt_bleu))
            scores.append(str(tgt_bleu))
        write_result(docs=scores, fname=self._path(eval_desc, "score"))
        self.vae.train(was)
        return scores

    def evaluate_style_transfer(self, eval_dir, eval_list, eval_desc="transfer"):
        """pairs (semantic source, syntactic reference) -> transferred sentence, and the linearised trees the
        model predicts for the source (.ss), the reference (.st) and its own output (.sp); BLEU of the output
        against either side in words and in trees, and of the two sides against each other (the bound)."""
        was = self._eval_mode()
        scores = []
        for data, name in zip(self._pair_sets(eval_dir, eval_list), eval_list):
            f = {k: self._path(name, eval_desc, k) for k in ("pred", "tgt", "ori", "ss", "st", "sp")}
            sem_ex, syn_ex, out = self._transfer_pass(data)
            write_result(con_to_plain(recovery_ref(syn_ex, self.vae.vocab.src)), fname=f["tgt"])
            write_result(con_to_plain(recovery_ref(sem_ex, self.vae.vocab.src)), fname=f["ori"])
            write_result(predict_to_plain(out["res"]), fname=f["pred"])
            word = "{} word : ori bleu:{}, tgt bleu:{} !".format(name, bleu.evaluate_file(f["pred"], f["ori"]),
                                                                 bleu.evaluate_file(f["pred"], f["tgt"]))
            print("finish {} with {}".format(name, word))
            syn_pred = []
            with torch.no_grad():
                for batch in Dataset.from_raw_file(f["pred"], e_type="Plain").batch_iter(self.eval_batch_size):
                    syn_pred.extend(self.vae.eval_syntax(batch))
            write_result(predict_to_plain(out["ori syn"]), fname=f["ss"])
            write_result(predict_to_plain(out["ref syn"]), fname=f["st"])
            write_result(predict_to_plain(syn_pred), fname=f["sp"])
            syn = "{} syn : ori bleu:{}, tgt bleu:{} !".format(name, bleu.evaluate_file(f["sp"], f["ss"]),
                                                               bleu.evaluate_file(f["sp"], f["st"]))
            bound = "{} word bleu:{}, syn bleu:{} !".format(name, bleu.evaluate_file(f["ori"], f["tgt"]),
                                                            bleu.evaluate_file(f["ss"], f["st"]))
            print("finish {} with {}".format(name, syn))
            print("finish {} with {}".format(name, bound))
            scores.append("\n".join([word, syn, bound]))
        write_result(docs=scores, fname=self._path(eval_desc, "score"))
        self.vae.train(was)
        return scores
