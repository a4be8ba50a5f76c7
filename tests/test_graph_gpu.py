"""The training step behind a HIP graph (train.GraphedTrainer) and the device-resident id cache
(structs/id_cache.py) on the MI355X.

Bar: a replayed step computes what the eager step computes -- same loss dictionary (1e-5 relative: the same kernels
on the same numbers; split-K atomics reorder sums) and the same parameters after N optimisation steps -- while the
values that change per step (KL weight, word-dropout rate, Adam's bias corrections and learning rate, dropout
seeds) are read from the graph's device slots.  The id cache's device-cut batches are bit-identical to the
host-tensorised ones of the reference's `to_input_variable` restatement (utils/nn_funcs.py:143-174)."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from parity_util import make_args, make_examples, product_examples, product_vocab, rel_err  # noqa: E402

pytestmark = pytest.mark.gpu

MID = dict(embed_size=60, hidden_size=60, enc_embed_dim=60, enc_hidden_dim=60, dec_embed_dim=60, dec_hidden_dim=60,
           latent_size=24, src_max_time_step=14, tgt_max_time_step=40)
NO_NOISE = dict(dec_ed=0.0, dec_rd=0.0, rnn_drop=0.0, unk_rate=0.0, enc_ed=0.0, enc_rd=0.0)


@pytest.fixture(autouse=True)
def _graphs_allowed():
    """the suite-wide contract check of the fused CE syncs the device at every backward: not capturable"""
    from dss_vae_amd import ops
    old = ops.CHECK_UNIFORM_UPSTREAM
    ops.CHECK_UNIFORM_UPSTREAM = False
    yield
    ops.CHECK_UNIFORM_UPSTREAM = old


def _cache(n, n_src, n_tgt, fixed, seed, src_len=9, tree_len=27):
    from dss_vae_amd.structs.id_cache import IdCache
    vocab = product_vocab(n_src, n_tgt)
    exs = product_examples(make_examples(n, n_src, n_tgt, src_len=src_len, tree_len=tree_len, fixed=fixed, seed=seed))
    np.random.RandomState(seed).shuffle(exs)  # make_examples sorts by length: undo, the iterator has to sort
    return IdCache.from_examples(exs, vocab).cuda(), exs, vocab


@pytest.mark.parametrize("fixed", [True, False])
def test_id_cache_batches_equal_host_tensorised_batches(fixed):
    from dss_vae_amd.structs import Dataset
    from dss_vae_amd.vae.base_vae import DeviceBatch
    cache, exs, vocab = _cache(53, 300, 40, fixed, seed=3)
    # an out-of-vocabulary word maps to <unk> in the cache as in VocabEntry.__getitem__
    exs[5].src[0] = "never-seen"
    from dss_vae_amd.structs.id_cache import IdCache
    cache = IdCache.from_examples(exs, vocab).cuda()
    for world, rank in ((1, 0), (2, 1)):
        ref = Dataset(exs).batch_iter(8 * world, shuffle=True, rng=np.random.RandomState(11))
        got = cache.batch_iter(8, shuffle=True, rng=np.random.RandomState(11), rank=rank, world=world)
        n = 0
        for ref_batch, (n_global, b) in zip(ref, got):
            assert n_global == len(ref_batch)
            if world > 1:
                from dss_vae_amd.parallel import shard_examples
                ref_batch = shard_examples(ref_batch, rank, world)
            a = DeviceBatch(ref_batch, vocab)
            assert b.signature() == a.signature()
            for name in ("src", "tgt_var", "syn_var"):
                ta, tb = getattr(a, name), getattr(b, name)
                assert torch.equal(ta, tb), name
                ka, kb = getattr(ta, "_keep_rows", None), getattr(tb, "_keep_rows", None)
                assert (ka is None) == (kb is None), name
                if ka is not None:
                    assert torch.equal(ka, kb), name
            assert list(a.length) == list(b.length) and torch.equal(a.length.dev, b.length.dev)
            assert list(a.tree_length) == list(b.tree_length) and torch.equal(a.tree_length.dev, b.tree_length.dev)
            n += 1
        assert n == (53 + 8 * world - 1) // (8 * world)


def test_keep_rows_matches_its_definition():
    from dss_vae_amd import ops
    rng = np.random.RandomState(0)
    for B, T in ((1, 5), (7, 9), (300, 33), (2500, 12)):
        lens = rng.randint(0, T + 3, size=B).astype(np.int32)
        want = [b * T + t for b, L in enumerate(lens) for t in range(min(int(L) + 1, T))]
        got = ops.keep_rows(torch.from_numpy(lens).cuda(), T, len(want))
        assert got.tolist() == want
        padded = ops.keep_rows(torch.from_numpy(lens).cuda(), T, len(want) + 37)
        assert padded[:len(want)].tolist() == want and set(padded[len(want):].tolist()) == {want[-1]}


def _build(name, args, vocab, seed=0):
    from dss_vae_amd.models import MODEL_CLS
    from dss_vae_amd.train import FlatAdam, build_model_params_
    m = MODEL_CLS[name](args=copy.deepcopy(args), vocab=vocab)
    build_model_params_(m, seed=seed)
    m.cuda().train()
    return m, FlatAdam(m, lr=0.002, betas=(0.9, 0.995), clip_grad=5.0)


def _run(name, args, batches, steps, graph, adv=False, it0=100):
    from dss_vae_amd import ops
    from dss_vae_amd.networks.rnn_base import GLOBAL_SEEDS
    from dss_vae_amd.train import GraphedTrainer
    torch.manual_seed(5)
    GLOBAL_SEEDS.state = 777
    model, opt = _build(name, args, batches[0][1])
    runner = GraphedTrainer(model, opt, adv, adv, enabled=graph)
    rets = []
    for s in range(steps):
        ret = runner(batches[s % len(batches)][0], it0 + s)
        rets.append({k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in ret.items()})
    ops.check_persistent_errors()
    return rets, opt.flat.clone(), list(opt.steps), runner


@pytest.mark.parametrize("name,adv", [("SyntaxVAE2", False), ("SyntaxVAE2", True), ("SyntaxVAE", False),
                                      ("VanillaVAE", False), ("DisentangleVAE", False)])
def test_replayed_step_equals_eager_step(name, adv):
    """10 optimisation steps around the steep part of the KL schedule (x0 = 104, k = 0.8: the weight goes 0.04 ->
    0.98), latent noise from torch's generator (the same stream eagerly and under replay), no dropout: the loss
    dictionary of every step and the parameters at the end agree; 7 of the 10 steps are replays of one graph"""
    over = dict(MID, **NO_NOISE)
    over.update(x0=104, k=0.8, warm_up=0, tgt_wd=False, model_select=name, task_type=name)
    args = make_args(**over)
    cache, exs, vocab = _cache(64, 300, 40, True, seed=2)
    idx = np.arange(64).reshape(4, 16)
    batches = [(cache.batch(i), vocab) for i in idx]
    eager, p_e, st_e, _ = _run(name, args, batches, 10, graph=False, adv=adv)
    graph, p_g, st_g, runner = _run(name, args, batches, 10, graph=True, adv=adv)
    assert runner.captures == 1 and runner.replays >= 7 and runner.replays + runner.eager == 10
    assert st_e == st_g
    for s, (a, b) in enumerate(zip(eager, graph)):
        assert set(a) == set(b)
        for k in a:
            assert rel_err(b[k], a[k]) <= 2e-5, (s, k, a[k], b[k])
    assert abs(eager[0]["KL Weight"] - eager[-1]["KL Weight"]) > 0.3  # (x kl_factor 0.5)
    # (Adam normalises every step, so the last bits of a gradient -- split-K atomics add in any order -- move a
    # parameter by more than their own size: two eager runs differ as much)
    assert rel_err(p_g, p_e) <= 2e-4


def test_per_step_seeds_and_rates_are_read_on_the_device():
    """a captured dropout + word dropout: every replay draws a new mask from the seed base written in front of it,
    and follows the rate written in front of it"""
    from dss_vae_amd import ops
    vocab = product_vocab(500, 40)
    x = torch.ones(4096, 64, device="cuda")
    ids = torch.randint(4, 500, (512, 40), device="cuda")
    dyn = ops.StepScalars(x.device)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    ops.DYN = dyn
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            rate = dyn.value("unk_rate", 0.0)
            y = ops.dropout(x, 0.25, 123)
            z = ops.unk_replace(ids, rate, vocab.src, seed=9)
    finally:
        ops.DYN = None
    masks, fracs = [], []
    for base, r in ((11, 0.0), (12, 0.5), (12, 0.5), (999, 0.1)):
        dyn.set_seed_base(base)
        dyn.set_float(dyn.slots["unk_rate"], r)
        dyn.upload()
        g.replay()
        torch.cuda.synchronize()
        masks.append((y != 0).clone())
        fracs.append(float((z == vocab.src.unk_id).float().mean().item()))
        keep = float(masks[-1].float().mean().item())
        assert abs(keep - 0.75) < 0.01 and float(y.max().item()) == pytest.approx(1.0 / 0.75)
    assert not torch.equal(masks[0], masks[1]) and torch.equal(masks[1], masks[2]) and not torch.equal(masks[2], masks[3])
    assert fracs[0] == 0.0 and abs(fracs[1] - 0.5) < 0.02 and fracs[1] == fracs[2] and abs(fracs[3] - 0.1) < 0.02


def test_shapes_get_their_own_graphs_and_stale_graphs_are_dropped():
    """ragged batches: a signature (shapes + non-pad row counts) is captured at its third appearance; steps of
    another kind in between that advance Adam's step count of only some parameters make a graph stale (the members
    of one of its ranges no longer share a count): the trainer notices, runs that step eagerly and captures again"""
    from dss_vae_amd.train import GraphedTrainer
    args = make_args(**dict(MID, **NO_NOISE, warm_up=0, tgt_wd=False))
    cache, exs, vocab = _cache(48, 300, 40, False, seed=4)
    order = np.argsort(-cache.src_len, kind="stable")
    b1, b2 = cache.batch(order[0:48:3]), cache.batch(order[1:48:2])
    assert b1.signature() != b2.signature() and b1.signature()[4] is not None
    torch.manual_seed(1)
    model, opt = _build("SyntaxVAE2", args, vocab)
    runner = GraphedTrainer(model, opt)
    it = 10
    for b in (b1, b2, b1, b2, b1, b2, b1, b2, b1, b2):
        it += 1
        runner(b, it)
    assert runner.captures == 2 and len(runner.graphs) == 2 and runner.replays == 5
    # steps of another kind ran in between and advanced only some parameters of one of the graph's Adam ranges
    rec = runner.graphs[runner._signature(b1, it + 1)]
    members = next(m for _, m in rec["dyn"].adam if len(m) > 1)
    opt.steps[members[0]] += 1
    before = (runner.captures, runner.eager)
    it += 1
    ret = runner(b1, it)
    assert runner.eager == before[1] + 1 and len(runner.graphs) == 1 and torch.isfinite(ret["Loss"]).all()
    for _ in range(3):
        it += 1
        runner(b1, it)
    assert runner.captures == before[0] + 1 and len(runner.graphs) == 2


def test_second_backward_without_zero_grad_accumulates():
    """the gradient sink must not hand out a slice that already holds a gradient"""
    from dss_vae_amd.train import FlatAdam
    from dss_vae_amd.vae.base_vae import Linear
    torch.manual_seed(0)
    lin = Linear(24, 12).cuda()
    opt = FlatAdam(lin, lr=0.1, clip_grad=0.0)
    x1, x2 = torch.randn(5, 24, device="cuda"), torch.randn(7, 24, device="cuda")
    opt.zero_grad()
    lin(x1).sum().backward()
    g1 = lin.weight.grad.clone()
    lin(x2).pow(2).sum().backward()
    w = lin.weight.detach()
    want = g1 + (2 * (x2 @ w.t() + lin.bias.detach())).t() @ x2
    assert rel_err(lin.weight.grad, want) < 1e-5
