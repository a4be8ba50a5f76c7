"""Decode-and-score glue (reference: dss_vae/metrics/evaluation.py:19-125, metrics/tools.py:17-164):
batches sorted by -len, model.predict per batch (greedy decode on the MI355X), input/pred/gold text
files, corpus BLEU."""
import os
import time

import numpy as np
import torch

from . import bleu


def write_result(docs, fname):
    with open(fname, "w") as f:
        if isinstance(docs, list):
            for doc in docs:
                f.write(" ".join(doc) if isinstance(doc, list) else doc)
                f.write("\n")
        elif isinstance(docs, dict):
            for key, val in docs.items():
                if val is not None:
                    f.write("%s:%f" % (key, torch.mean(val) if isinstance(val, torch.Tensor) else val))
                    f.write("\n")


def predict_to_plain(decode_results):
    return [" ".join(hyp[0][0]) for hyp in decode_results]


def prepare_ref(examples, eval_tgt, vocab):
    return [[[e.src]] for e in examples] if eval_tgt == "src" else [[[e.tgt]] for e in examples]


def batchify_examples(examples, sort_key="src", batch_size=None, sort=True):
    if batch_size is None:
        batch_size = 50 if len(examples) > 50 else len(examples)
    if sort:
        origin_ids = sorted(range(len(examples)), key=lambda i: -len(getattr(examples[i], sort_key)))
    else:
        origin_ids = np.arange(len(examples))
    examples = sorted(examples, key=lambda e: -len(getattr(e, sort_key)))
    return [examples[i:i + batch_size] for i in range(0, len(examples), batch_size)], origin_ids


def decode(examples, model):
    was_training = model.training
    model.eval()
    with torch.no_grad():
        res = model.predict(examples)
    if was_training:
        model.train()
    return res


def decode_batches(batches, model):
    """decode(b, model) for every batch, in order; with a model that offers predict_async / finish_predict the
    loop is software-pipelined (batch k+1 is on the GPU while the host turns the ids of batch k into words)"""
    if not (hasattr(model, "predict_async") and hasattr(model, "finish_predict")):
        return [r for b in batches for r in decode(b, model)]
    was_training = model.training
    model.eval()
    out, pending = [], None
    with torch.no_grad():
        for b in batches:
            handle = model.predict_async(b)
            if pending is not None:
                out.extend(model.finish_predict(pending))
            pending = handle
        if pending is not None:
            out.extend(model.finish_predict(pending))
    if was_training:
        model.train()
    return out


def prediction(examples, model, sort_key="src", eval_src="src", out_dir=None, batch_size=None, write=True):
    """write=False keeps input/pred lines in the result ("input_lines" / "pred_lines") and touches no file:
    under data parallel only rank 0 writes, the other replicas score what they decoded in memory"""
    if batch_size is None:
        batch_size = 50 if "eval_bs" not in model.args else model.args.eval_bs
    batches, ids = batchify_examples(examples, sort_key=sort_key, batch_size=batch_size)
    t0 = time.time()
    cache = decode_batches(batches, model)
    use_time = time.time() - t0
    pred = [None] * len(cache)
    for new_id, ori in enumerate(ids):
        pred[ori] = cache[new_id]
    out_dir = "./" if out_dir is None else out_dir
    res = {"predict": pred, "oracle_accuracy": 0.0, "use_time": use_time,
           "input_file": os.path.join(out_dir, "input.txt"), "pred_file": os.path.join(out_dir, "pred.txt"),
           "input_lines": predict_to_plain(prepare_ref(examples, eval_src, model.vocab)),
           "pred_lines": predict_to_plain(pred)}
    if write:
        os.makedirs(out_dir, exist_ok=True)
        write_result(res["input_lines"], res["input_file"])
        write_result(res["pred_lines"], res["pred_file"])
    return res


def new_evaluate(examples, model, sort_key="src", eval_src="src", eval_tgt="src", ret_dec_result=False, batch_size=None,
                 out_dir=None, write=True):
    res = prediction(examples, model, sort_key, eval_src, out_dir, batch_size, write=write)
    res["reference"] = prepare_ref(examples, eval_tgt, model.vocab)
    out_dir = "./" if out_dir is None else out_dir
    res["gold_file"] = os.path.join(out_dir, "gold.txt")
    res["gold_lines"] = predict_to_plain(res["reference"])
    if write:
        write_result(res["gold_lines"], res["gold_file"])
        res["accuracy"] = bleu.evaluate_file(res["pred_file"], res["gold_file"])
    else:
        res["accuracy"] = bleu.evaluate_lines(res["pred_lines"], [res["gold_lines"]])
    return (res, res["predict"]) if ret_dec_result else res


def evaluate(examples, model, sort_key="src", eval_src="src", eval_tgt="src", return_decode_result=False,
             batch_size=None, out_dir=None, write=True):
    """metrics/evaluation.py:119-184: like new_evaluate but in corpus-order batches (each sorted by -len), so the
    references are collected in decode order instead of un-permuting the predictions"""
    if batch_size is None:
        batch_size = 50 if "eval_bs" not in model.args else model.args.eval_bs
    from ..structs import Dataset
    ref_examples, pred = [], []
    t0 = time.time()
    batches = []
    for batch in Dataset(examples).batch_iter(batch_size, shuffle=False):
        batch.sort(key=lambda e: -len(getattr(e, sort_key)))
        ref_examples.extend(batch)
        batches.append(batch)
    pred = decode_batches(batches, model)
    res = {"reference": prepare_ref(ref_examples, eval_tgt, model.vocab), "predict": pred, "oracle_accuracy": 0.0,
           "use_time": time.time() - t0}
    out_dir = "./" if out_dir is None else out_dir
    res["input_file"], res["pred_file"], res["gold_file"] = (os.path.join(out_dir, n) for n in
                                                             ("input.txt", "pred.txt", "gold.txt"))
    res["input_lines"] = predict_to_plain(prepare_ref(ref_examples, eval_src, model.vocab))
    res["pred_lines"], res["gold_lines"] = predict_to_plain(pred), predict_to_plain(res["reference"])
    if write:
        os.makedirs(out_dir, exist_ok=True)
        for key in ("input", "pred", "gold"):
            write_result(res[key + "_lines"], res[key + "_file"])
        res["accuracy"] = bleu.evaluate_file(res["pred_file"], res["gold_file"])
    else:
        res["accuracy"] = bleu.evaluate_lines(res["pred_lines"], [res["gold_lines"]])
    return (res, pred) if return_decode_result else res
