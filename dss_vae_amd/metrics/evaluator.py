"""Dev-set evaluators of the training loop (reference: dss_vae/metrics/syntax_vae_evaluator.py:50-161,
metrics/vae_metrics.py:30-60): ELBO tracker over the dev set + reconstruction BLEU through the greedy
decode path.  `score_item` selects the entry the loop compares (yaml key `dev_item`)."""
import os
import time

import torch

from ..utils.model_utils import update_tracker
from .evaluation import batchify_examples, new_evaluate, write_result


def get_dev_item(task_type, bleu=-1):
    """which entry of the evaluation dict selects checkpoints (syntax_vae_evaluator.py:40-47)"""
    if task_type == "EnhanceSyntaxVAE":
        return "ELBO"
    return "TGT BLEU" if bleu < 50.0 else "TGT_ORI_RATE"


class SyntaxVaeEvaluator(object):
    def __init__(self, model, eval_set, eval_lists=None, sort_key="src", eval_tgt="tgt", out_dir="./out", batch_size=20,
                 write_down=False, **kwargs):
        self.model, self.eval_set, self.out_dir, self.batch_size = model, eval_set, out_dir, batch_size
        self.sort_key, self.eval_tgt, self.write_down = sort_key, eval_tgt, write_down
        self.score_item = "BLEU" if model.args.dev_item is None else model.args.dev_item

    def eval_elbo(self, eval_desc="vae-elbo", eval_step=None):
        model = self.model
        training = model.training
        model.eval()
        step = eval_step if eval_step is not None else 2 * model.args.x0
        track = {}
        batches, _ = batchify_examples(self.eval_set, batch_size=self.batch_size, sort=False)
        with torch.no_grad():
            for batch in batches:
                track = update_tracker(model.get_loss(batch, step), track)
        if self.write_down:
            os.makedirs(self.out_dir, exist_ok=True)
            write_result(track, fname=os.path.join(self.out_dir, eval_desc + ".score"))
        model.train(training)
        return track

    def eval_reconstruct(self, eval_desc="vae-reconsturct"):
        model = self.model
        training = model.training
        model.eval()
        res = new_evaluate(self.eval_set, model, sort_key=self.sort_key, eval_tgt=self.eval_tgt, batch_size=self.batch_size,
                           out_dir=os.path.join(self.out_dir, eval_desc), write=self.write_down)
        model.train(training)
        return {"BLEU": res["accuracy"], "EVAL TIME": res["use_time"],
                "EVAL SPEED": len(self.eval_set) / max(res["use_time"], 1e-9)}

    def eval_paraphrase(self, eval_desc="paraphrase"):
        """decode the dev sources, BLEU against the paraphrase targets and against the inputs themselves
        (syntax_vae_evaluator.py:66-91)"""
        from . import bleu
        model = self.model
        training = model.training
        model.eval()
        res = new_evaluate(self.eval_set, model, sort_key=self.sort_key, eval_tgt=self.eval_tgt, batch_size=self.batch_size,
                           out_dir=os.path.join(self.out_dir, eval_desc), write=self.write_down)
        tgt_bleu = res["accuracy"]
        ori_bleu = bleu.evaluate_lines(res["pred_lines"], [res["input_lines"]])
        model.train(training)
        return {"TGT BLEU": tgt_bleu, "ORI BLEU": ori_bleu, "TGT_ORI_RATE": tgt_bleu / max(ori_bleu, 1e-9),
                "EVAL TIME": res["use_time"], "EVAL SPEED": len(self.eval_set) / max(res["use_time"], 1e-9)}

    def __call__(self, eval_desc="syntax-vae", step=None, **kwargs):
        t0 = time.time()
        args = self.model.args
        if getattr(args, "task_type", None) == "SyntaxPara":  # syntax_vae_evaluator.py:154-160
            track = self.eval_paraphrase()
            self.score_item = get_dev_item(args.task_type, track["TGT BLEU"]) if args.dev_item is None else args.dev_item
            return track
        track = self.eval_elbo(eval_step=step)
        track.update(**self.eval_reconstruct())
        track["EVAL TIME"] = time.time() - t0
        return track


SyntaxEvaluator = SyntaxVaeEvaluator  # the reference's name for this class (syntax_vae_evaluator.py:50)


def get_evaluator(eval_choice, model, eval_set, **kwargs):
    """every VAE variant of the hot path is scored the same way (ELBO terms + reconstruction BLEU)"""
    return SyntaxVaeEvaluator(model, eval_set, **kwargs)
