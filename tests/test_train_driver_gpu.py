"""Training driver and evaluator glue (N3) on the MI355X: files in the reference's
formats -> training() with dev evaluation, checkpoint and scalars; the 3-pass adversarial iteration
(bin/train.py:90-126) against the CPU oracle driven by torch.optim.Adam."""
import copy
import json
import os
import sys
from argparse import Namespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle  # noqa: E402
from parity_util import build_pair, make_args, make_examples, product_examples, product_vocab  # noqa: E402

pytestmark = pytest.mark.gpu

TINY = dict(embed_size=32, hidden_size=32, enc_embed_dim=32, enc_hidden_dim=32, dec_embed_dim=32, dec_hidden_dim=32,
            latent_size=16, src_max_time_step=12, tgt_max_time_step=30, dec_ed=0.0, dec_rd=0.0, tgt_wd=False)


def test_training_driver_end_to_end(tmp_path):
    from dss_vae_amd.bin.train import training
    from dss_vae_amd.structs import Dataset
    from dss_vae_amd.structs.io import load_model_bin
    n_src, n_tgt = 120, 30
    vocab = product_vocab(n_src, n_tgt)
    vocab.save(str(tmp_path / "vocab.bin"))
    Dataset(product_examples(make_examples(40, n_src, n_tgt, src_len=9, tree_len=27, fixed=False, seed=1))).save(
        str(tmp_path / "train.bin"))
    Dataset(product_examples(make_examples(12, n_src, n_tgt, src_len=9, tree_len=27, fixed=False, seed=2))).save(
        str(tmp_path / "dev.bin"))
    main_args = Namespace(train_file=str(tmp_path / "train.bin"), dev_file=str(tmp_path / "dev.bin"),
                          vocab=str(tmp_path / "vocab.bin"), eval_lists=None, logdir=str(tmp_path / "logs"),
                          model_dir=str(tmp_path / "models"), exp_name="t", task_type="SyntaxVAE2", start_iter=0,
                          batch_size=8, log_every=2, dev_every=3, lr=0.0005, lr_decay=0.8, patience=1, clip_grad=5.0,
                          reset_optimizer=False, mode="train", seed=4)
    model_args = make_args(warm_up=0, eval_bs=5, dev_item="ELBO", **TINY)
    out = training(main_args, model_args, max_iters=7)
    assert out["train_iter"] == 7 and len(out["history"]) == 2
    assert os.path.exists(out["model_file"]) and os.path.exists(out["model_file"] + ".optim.bin")
    ck = load_model_bin(out["model_file"])
    assert set(ck) == {"args", "vocab", "state_dict"} and ck["vocab"].src["w1"] == vocab.src["w1"]
    assert set(ck["state_dict"]) == set(out["model"].state_dict())
    tags = {json.loads(line)["tag"] for line in open(os.path.join(out["log_dir"], "scalars.jsonl"))}
    for t in ("SyntaxVAE2/Train/Loss", "SyntaxVAE2/Train/KL Loss", "Optimize/lr", "SyntaxVAE2/Valid/ELBO",
              "SyntaxVAE2/Valid/BLEU", "SyntaxVAE2/Valid/Best ELBO"):
        assert t in tags, (t, sorted(tags))
    rec = os.path.join(os.path.dirname(out["model_file"]), "out", "vae-reconsturct")
    for f in ("input.txt", "pred.txt", "gold.txt"):
        assert len(open(os.path.join(rec, f)).read().splitlines()) == 12
    base = os.path.dirname(out["model_file"])
    assert os.path.exists(os.path.join(base, "base_config.yaml")) and os.path.exists(os.path.join(base, "model_config.yaml"))
    # test-mode entry of a driver: configs + checkpoint of the experiment -> model on the GPU -> decode
    from dss_vae_amd.utils.model_utils import load_configs, load_model
    pre_main, pre_model = load_configs(base)
    assert pre_model.model_select == "SyntaxVAE2" and pre_main.batch_size == 8
    m2 = load_model(pre_main, pre_model, check_dir=False)
    sd2 = m2.state_dict()
    for k, v in ck["state_dict"].items():
        assert torch.equal(sd2[k].cpu(), v.cpu()), k
    m2.eval()
    dev = Dataset.from_bin_file(str(tmp_path / "dev.bin")).examples[:4]
    with torch.no_grad():
        res = m2.predict(dev)
    assert len(res) == 4 and all(isinstance(r[0][0], list) for r in res)
    # bin/test.py::test_vae over the experiment's checkpoint: mode 0 (dev / test reconstruction + generation
    # statistics) and mode 4 (syntax transfer over a pair file)
    from dss_vae_amd.bin.test import test_vae
    pre_main.test_file = pre_main.dev_file
    rep = test_vae(pre_main, pre_model, input_mode=0)
    for f in ("dev.pred", "dev.gold", "dev.score", "test.pred", "gen.text", "gen.score"):
        assert os.path.exists(os.path.join(rep["out_dir"], f)), f
    assert rep["out_dir"].endswith(os.path.join("SyntaxVAE2.t", "Test")) and rep["gen"]["Gen AvgLen"] > 0
    (tmp_path / "pairs").mkdir()
    (tmp_path / "pairs" / "dev.para.txt").write_text(
        "".join(" ".join(a.src) + "\t" + " ".join(b.src) + "\n" for a, b in zip(dev, dev[1:] + dev[:1])))
    rep = test_vae(pre_main, pre_model, input_mode=4, para_eval_dir=str(tmp_path / "pairs"), model=m2)
    assert "unmatch" not in rep and rep["match"][0].count("bleu:") == 6
    with pytest.raises(ValueError):
        test_vae(pre_main, pre_model, input_mode=2)


def test_test_time_evaluators(tmp_path):
    """metrics/vae_metrics.py:30-268 around the GPU decode path: reconstruction, prior generation statistics,
    paraphrase (`sem` / `sem-only`), controlled generation and syntax transfer over `sentence \\t reference` files;
    the reference's file names, one line per example, finite scores"""
    from dss_vae_amd.metrics.vae_metrics import SyntaxVaeEvaluator
    from dss_vae_amd.models import MODEL_CLS
    from dss_vae_amd.train import build_model_params_
    n_src, n_tgt = 120, 30
    model = MODEL_CLS["SyntaxVAE2"](args=make_args(eval_bs=5, dev_item="ELBO", **TINY), vocab=product_vocab(n_src, n_tgt))
    build_model_params_(model, seed=5)
    model.cuda().eval()
    exs = product_examples(make_examples(7, n_src, n_tgt, src_len=9, tree_len=27, fixed=False, seed=3))
    pairs = tmp_path / "pairs"
    pairs.mkdir()
    others = exs[1:] + exs[:1]
    (pairs / "dev.txt").write_text("".join(" ".join(a.src) + "\t" + " ".join(b.src) + "\n" for a, b in zip(exs, others)))
    ev = SyntaxVaeEvaluator(model, str(tmp_path / "out"), batch_size=3, train_batch_size=4)
    out = str(tmp_path / "out")

    def lines(name):
        return open(os.path.join(out, name)).read().splitlines()

    track, res = ev.evaluate_reconstruction(exs, "rec")
    assert len(lines("rec.pred")) == len(lines("rec.gold")) == 7 and 0.0 <= res["accuracy"] <= 100.0
    assert sorted(lines("rec.gold")) == sorted(" ".join(e.src) for e in exs)
    assert any(l.startswith("ELBO:") for l in lines("rec.score")) and "Loss" in track
    gen = ev.evaluate_generation(exs, sample_size=6, eval_desc="gen")
    assert len(lines("gen.text")) == 6 and gen["Gen AvgLen"] > 0 and gen["UnigramKL"] >= 0.0
    for fn, desc in ((ev.evaluate_para, "para"), (ev.evaluate_pure_para, "pure-para")):
        (ori_bleu, tgt_bleu), = fn(str(pairs), ["dev.txt"])
        assert 0.0 <= ori_bleu <= 100.0 and 0.0 <= tgt_bleu <= 100.0
        assert len(lines("dev.txt.%s.pred" % desc)) == len(lines("dev.txt.%s.ori" % desc)) == 7
        assert lines("dev.txt.%s.tgt" % desc)[0] == " ".join(others[0].src)
    score, = ev.evaluate_control(str(pairs), ["dev.txt"])
    assert 0.0 <= float(score) <= 100.0 and len(lines("dev.txt.con.pred")) == 7
    assert lines("dev.txt.con.input")[2] == " ".join(exs[2].src) and lines("con.score") == [score]
    report, = ev.evaluate_style_transfer(str(pairs), ["dev.txt"])
    assert report.count("bleu:") == 6
    for suffix in ("pred", "tgt", "ori", "ss", "st", "sp"):
        assert len(lines("dev.txt.transfer." + suffix)) == 7, suffix
    assert not model.training
    # the pipelined decode loop of the evaluators returns what predict() returns batch by batch
    from dss_vae_amd.metrics.evaluation import decode_batches
    batches = [sorted(exs[:3], key=lambda e: -len(e.src)), sorted(exs[3:], key=lambda e: -len(e.src))]
    with torch.no_grad():
        want = model.predict(batches[0]) + model.predict(batches[1])
    assert decode_batches(batches, model) == want


def _dp_driver_worker(rank, world, port, root, q, warm_up=0):
    import faulthandler
    import numpy as np
    import torch.distributed as dist
    faulthandler.dump_traceback_later(150, exit=True)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dss_vae_amd.bin.train import training
        np.random.seed(100 + rank)        # ranks disagree on purpose: the driver must not depend on it
        torch.manual_seed(200 + rank)
        main_args = Namespace(train_file=os.path.join(root, "train.bin"), dev_file=os.path.join(root, "dev.bin"),
                              vocab=os.path.join(root, "vocab.bin"), eval_lists=None, logdir=os.path.join(root, "logs"),
                              model_dir=os.path.join(root, "models"), exp_name="dp", task_type="SyntaxVAE2",
                              start_iter=0, batch_size=4, log_every=2, dev_every=3, lr=0.0005, lr_decay=0.8,
                              patience=1, clip_grad=5.0, reset_optimizer=False, mode="train", seed=None)
        model_args = make_args(warm_up=warm_up, eval_bs=5, dev_item="ELBO", adv_train=True, **TINY)
        out = training(main_args, model_args, max_iters=7)  # 21 sentences / (4 x 2): tail batch of 5 = 3 + 2
        flat = out["optimizer"].flat.detach().double().cpu()
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1]), "replicas diverged: %g" % (both[0] - both[1]).abs().max().item()
        assert out["train_iter"] == 7 and torch.isfinite(flat).all()
        if rank == 0:
            assert os.path.exists(out["model_file"])
            tags = {json.loads(line)["tag"] for line in open(os.path.join(out["log_dir"], "scalars.jsonl"))}
            assert "SyntaxVAE2/Train/Loss" in tags and "SyntaxVAE2/Valid/ELBO" in tags
        dist.barrier()
        faulthandler.cancel_dump_traceback_later()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "FAIL %r" % (e,)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("warm_up", [0, 4], ids=["past_warm_up", "crossing_warm_up"])
def test_training_driver_two_ranks(tmp_path, warm_up):
    """the driver under data parallel (two ranks on the one GPU, gloo): differently seeded ranks end with
    bit-identical replicas after 7 adversarial iterations incl. an unequal tail batch, dev evaluation and
    rank 0's checkpoint / schedule broadcast.  Second case: the -adv term joins the loss at iteration 5
    (train_iter > warm_up), i.e. the autograd graph of the main pass changes under the learned bucket schedule."""
    import socket
    import torch.multiprocessing as mp
    from dss_vae_amd.structs import Dataset
    n_src, n_tgt = 120, 30
    product_vocab(n_src, n_tgt).save(str(tmp_path / "vocab.bin"))
    Dataset(product_examples(make_examples(21, n_src, n_tgt, src_len=9, tree_len=27, fixed=False, seed=1))).save(
        str(tmp_path / "train.bin"))
    Dataset(product_examples(make_examples(6, n_src, n_tgt, src_len=9, tree_len=27, fixed=False, seed=2))).save(
        str(tmp_path / "dev.bin"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_driver_worker, args=(r, 2, port, str(tmp_path), q, warm_up)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=200) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


@pytest.mark.parametrize("model", ["SyntaxVAE2", "DisentangleVAE"])
def test_three_pass_adversarial_iteration_matches_oracle(model):
    """dis-syn step, dis-sem step, main step, twice; torch.optim.Adam skips parameters without a gradient
    and counts steps per parameter -- the flat-buffer optimiser must end at the same parameters."""
    from dss_vae_amd.train import FlatAdam
    n_src, n_tgt, B, Z = 150, 30, 6, 16
    args = make_args(**TINY)
    om, pm = build_pair(model, args, n_src, n_tgt, seed=2, train=True)
    exs = make_examples(B, n_src, n_tgt, src_len=9, tree_len=27, fixed=False, seed=3)
    pexs = product_examples(exs)
    oopt = torch.optim.Adam(om.parameters(), lr=0.0005, betas=(0.9, 0.995))
    popt = FlatAdam(pm, lr=0.0005, betas=(0.9, 0.995), clip_grad=5.0)
    g = torch.Generator().manual_seed(9)
    for it in range(2):
        noise = {"eps_syn": torch.randn(B, Z, generator=g), "eps_sem": torch.randn(B, Z, generator=g)}
        om.noise, pm.noise = dict(noise), dict(noise)
        step = 6000 + it
        for key in ("dis syn", "dis sem"):
            ol = om.get_loss(exs, step, is_dis=True)[key]
            oopt.zero_grad()
            ol.backward()
            torch.nn.utils.clip_grad_norm_(om.parameters(), 5.0)
            oopt.step()
            pl = pm.get_loss(pexs, step, is_dis=True)[key]
            assert abs(float(pl) - float(ol)) <= 1e-4 * abs(float(ol)), (it, key, float(pl), float(ol))
            popt.zero_grad(key)
            pl.backward()
            popt.step()
        oret = om.get_loss(exs, step)
        oopt.zero_grad()
        torch.mean(oret["Loss"]).backward()
        torch.nn.utils.clip_grad_norm_(om.parameters(), 5.0)
        oopt.step()
        pret = pm.get_loss(pexs, step)
        assert abs(float(pret["Loss"]) - float(oret["Loss"])) <= 2e-4 * abs(float(oret["Loss"]))
        popt.zero_grad("main")
        torch.mean(pret["Loss"]).backward()
        popt.step()
    osd, psd = om.state_dict(), pm.state_dict()
    for k in osd:
        a, b = psd[k].cpu().double(), osd[k].double()
        assert torch.allclose(a, b, rtol=1e-4, atol=4e-5), (k, (a - b).abs().max().item())


def test_dev_evaluator_matches_oracle(tmp_path):
    """SURVEY 8f N3 at the oracle's level (metrics/syntax_vae_evaluator.py:114-161, utils/model_utils.py:16-30):
    `SyntaxEvaluator()` on a small ragged dev set -- every tracked loss term equals the oracle's `get_loss` terms of
    the same batches (each batch's mean, concatenated in batch order, as `update_tracker` does), `pred.txt` equals
    the oracle's greedy words sentence by sentence, and BLEU equals metrics/bleu.py applied to those words."""
    from dss_vae_amd.metrics import bleu
    from dss_vae_amd.metrics.evaluation import batchify_examples
    from dss_vae_amd.metrics.evaluator import SyntaxEvaluator
    n_src, n_tgt = 120, 30
    args = make_args(eval_bs=5, dev_item="ELBO", x0=50, **TINY)
    om, pm = build_pair("SyntaxVAE2", args, n_src, n_tgt, seed=11, train=False)
    o_exs = make_examples(13, n_src, n_tgt, src_len=9, tree_len=27, fixed=False, seed=21)
    import random
    random.Random(3).shuffle(o_exs)                      # a dev file is not sorted by length
    p_exs = product_examples(o_exs)
    ev = SyntaxEvaluator(pm, p_exs, sort_key="src", eval_tgt="src", batch_size=5, out_dir=str(tmp_path / "out"),
                         write_down=True)
    out = ev()
    # ---- ELBO tracker
    batches, _ = batchify_examples(o_exs, batch_size=5, sort=False)
    assert [len(b) for b in batches] == [5, 5, 3]
    want = {}
    with torch.no_grad():
        for b in batches:
            for k, v in om.get_loss(b, 2 * args.x0).items():
                if isinstance(v, torch.Tensor):
                    want.setdefault(k, []).append(float(v.mean()))
                else:
                    want[k] = want.get(k, 0.0) + v
    for k, w in want.items():
        got = out[k]
        if isinstance(w, list):
            assert isinstance(got, torch.Tensor) and got.numel() == 3, k
            for a, b in zip(got.tolist(), w):
                assert abs(a - b) <= 1e-4 * max(abs(b), 1e-6), (k, got.tolist(), w)
        else:
            assert abs(got - w) <= 1e-6 * max(abs(w), 1.0), k
    assert out[ev.score_item] is out["ELBO"]
    scores = dict(l.split(":") for l in open(os.path.join(str(tmp_path / "out"), "vae-elbo.score")).read().splitlines())
    assert abs(float(scores["ELBO"]) - sum(want["ELBO"]) / 3) <= 1e-4 * abs(sum(want["ELBO"]) / 3)
    # ---- reconstruction: greedy words and BLEU
    rec = os.path.join(str(tmp_path / "out"), "vae-reconsturct")
    pred = open(os.path.join(rec, "pred.txt")).read().splitlines()
    gold = open(os.path.join(rec, "gold.txt")).read().splitlines()
    assert gold == [" ".join(e.src) for e in o_exs]
    o_pred = []
    with torch.no_grad():
        for e in o_exs:
            toks = om.predict_tokens([e])
            o_pred.append(" ".join(oracle.ids_to_words(toks[:, 0].tolist(), om.vocab.src)))
    assert pred == o_pred
    assert abs(out["BLEU"] - bleu.evaluate_lines(o_pred, [gold])) < 1e-9
