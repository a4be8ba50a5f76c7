"""Test-mode driver (reference: dss_vae/bin/test.py:126-187 `test_vae`): the checkpoint of an experiment ->
model on the MI355X -> one of the evaluation passes of metrics/vae_metrics.py, written under
<model_dir>/<model>[.<exp>]/Test.  The pair-file directory the reference hard-codes is an argument here; the
tree-edit-distance report of `testing()` (zss, nltk) and the interactive prompt loop are out of scope."""
import os

from ..metrics.vae_metrics import SyntaxVaeEvaluator
from ..structs import Dataset
from ..utils.model_utils import get_eval_info, load_model

MODES = {0: "reconstruct dev / test + generation statistics", 1: "generation statistics", 2: "paraphrase",
         3: "controlled generation", 4: "syntax transfer", 5: "paraphrase with sampled syntax"}


def test_vae(main_args, model_args, input_mode=0, para_eval_dir=None, para_eval_list=("dev.para.txt",),
             trans_eval_list=("trans.length.txt", "trans.random.txt"), model=None):
    if input_mode not in MODES:
        raise ValueError("input_mode %r: one of %s (the interactive loop of the reference is not provided)"
                         % (input_mode, sorted(MODES)))
    if input_mode >= 2 and para_eval_dir is None:
        raise ValueError("input_mode %d reads `sentence \\t reference` files: pass para_eval_dir" % input_mode)
    if model is None:
        model = load_model(main_args, model_args, check_dir=False)
    out_dir = get_eval_info(main_args=main_args, model_args=model_args, mode="Test")
    model.eval()
    evaluator = SyntaxVaeEvaluator(model=model, out_dir=out_dir, batch_size=model_args.eval_bs,
                                   train_batch_size=main_args.batch_size)
    para_eval_list, trans_eval_list = list(para_eval_list), list(trans_eval_list)
    ret = {"out_dir": out_dir, "mode": MODES[input_mode]}
    if input_mode in (0, 1):
        train_exam = Dataset.from_bin_file(main_args.train_file).examples
        test_exam = Dataset.from_bin_file(main_args.test_file).examples
        if input_mode == 0:
            print("========dev reconstructor========")
            ret["dev"] = evaluator.evaluate_reconstruction(Dataset.from_bin_file(main_args.dev_file).examples, "dev")
            print("========test reconstructor=======")
            ret["test"] = evaluator.evaluate_reconstruction(test_exam, "test")
        print("========generating samples=======")
        ret["gen"] = evaluator.evaluate_generation(corpus_examples=train_exam, sample_size=len(test_exam), eval_desc="gen")
    elif input_mode == 2:
        print("========generating paraphrase========")
        ret["para"] = evaluator.evaluate_para(eval_dir=para_eval_dir, eval_list=para_eval_list)
    elif input_mode == 3:
        print("========supervised generation========")
        ret["con"] = evaluator.evaluate_control(eval_dir=para_eval_dir, eval_list=para_eval_list)
    elif input_mode == 4:
        print("========style transfer========")
        present = [f for f in trans_eval_list if os.path.exists(os.path.join(para_eval_dir, f))]
        if present:
            ret["unmatch"] = evaluator.evaluate_style_transfer(para_eval_dir, present, eval_desc="unmatch")
        ret["match"] = evaluator.evaluate_style_transfer(para_eval_dir, para_eval_list, eval_desc="match")
    elif input_mode == 5:
        print("========random syntax select========")
        ret["pure-para"] = evaluator.evaluate_pure_para(eval_dir=para_eval_dir, eval_list=para_eval_list)
    print("finish")
    return ret
