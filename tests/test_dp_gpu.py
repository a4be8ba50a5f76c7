"""Data-parallel training step of the product path on the MI355X: two ranks share the one GPU of the
test box (gloo moves the buckets; RCCL needs one GPU per rank, which the driver's 8-GPU run has) and
must end, after two optimisation steps on a strided deal of the batch, with the parameters of the
single-process run on the whole batch (every loss term is a sum over sentences / B)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu

B, Z, STEPS = 8, 16, 2


def _setup():
    from parity_util import make_args, make_examples, product_examples, product_vocab
    from dss_vae_amd.models import MODEL_CLS
    from dss_vae_amd.train import build_model_params_
    args = make_args(embed_size=32, hidden_size=32, enc_embed_dim=32, enc_hidden_dim=32, dec_embed_dim=32,
                     dec_hidden_dim=32, latent_size=Z, src_max_time_step=12, tgt_max_time_step=30, dec_ed=0.0,
                     dec_rd=0.0, tgt_wd=False)
    n_src, n_tgt = 200, 40
    model = MODEL_CLS["SyntaxVAE2"](args=args, vocab=product_vocab(n_src, n_tgt))
    build_model_params_(model, seed=3)
    model.cuda().train()
    exs = product_examples(make_examples(B, n_src, n_tgt, src_len=9, tree_len=27, fixed=True, seed=5))
    g = torch.Generator().manual_seed(7)
    eps = [(torch.randn(B, Z, generator=g), torch.randn(B, Z, generator=g)) for _ in range(STEPS)]
    return model, exs, eps


def _run(model, opt, exs, eps, sel, adv=False, n_global=None):
    """adv: the 3-pass adversarial iteration (two discriminator steps + the main step) instead of the main step"""
    from dss_vae_amd.train import train_iteration, train_step
    losses = []
    for s in range(STEPS):
        model.noise = {"eps_syn": eps[s][0][sel], "eps_sem": eps[s][1][sel]}
        if adv:
            ret = train_iteration(model, opt, exs[sel], 6000 + s, True, True, global_batch=n_global)
        else:
            ret = train_step(model, opt, exs[sel], 6000 + s)
        losses.append(float(ret["Loss"]))
    torch.cuda.synchronize()
    return losses


def _worker(rank, world, port, q, n_global=B, adv=False):
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=True)  # a hung rank names its frame instead of stalling the run
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dss_vae_amd.train import FlatAdam
        model, exs, eps = _setup()
        opt = FlatAdam(model, bucket_mb=0.01)
        assert opt.buckets is not None and len(opt.buckets.buckets) > 3
        sel = slice(rank, n_global, world)
        n_local = len(range(*sel.indices(B)))
        losses = _run(model, opt, exs, eps, sel, adv, n_global)
        assert opt.buckets.sent_early > 0, "second step should have overlapped buckets with backward"
        lt = torch.tensor(losses, dtype=torch.float64) * n_local * world / n_global  # sum / B_local -> share of sum / B
        dist.all_reduce(lt)
        if rank == 0:
            ref, exs, eps = _setup()
            ropt = FlatAdam(ref, data_parallel=False)
            rl = _run(ref, ropt, exs, eps, slice(0, n_global), adv)
            for a, b in zip((lt / world).tolist(), rl):
                assert abs(a - b) <= 1e-4 * abs(b), (a, b)
            sd, rsd = model.state_dict(), ref.state_dict()
            for k in rsd:
                assert torch.allclose(sd[k], rsd[k], rtol=1e-4, atol=2e-5), (k, (sd[k] - rsd[k]).abs().max().item())
        dist.barrier()
        faulthandler.cancel_dump_traceback_later()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "FAIL %r" % (e,)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_global,adv", [(B, False), (B - 1, True)],
                         ids=["main_step_equal_shards", "adversarial_iteration_unequal_shards"])
def test_two_rank_step_equals_single_process(n_global, adv):
    """second case: 4 + 3 sentences and the 3-pass iteration -- the main loss (sum / B_local) is weighted by the
    shard size and the discriminator losses (plain sums) by the world size, so that the gradient average is
    the single-process gradient on all 7 sentences"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, n_global, adv)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _graph_worker(rank, world, port, q):
    import faulthandler
    faulthandler.dump_traceback_later(120, exit=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dss_vae_amd import ops
        from dss_vae_amd.train import FlatAdam, GraphedTrainer
        from dss_vae_amd.vae.base_vae import DeviceBatch
        ops.CHECK_UNIFORM_UPSTREAM = False
        results = []
        for graph in (False, True):
            model, exs, _ = _setup()
            opt = FlatAdam(model, bucket_mb=0.01)
            batch = DeviceBatch(exs[rank::world], model.vocab)
            torch.manual_seed(100 + rank)
            runner = GraphedTrainer(model, opt, enabled=graph)
            losses = [float(runner(batch, 6000 + s, global_batch=B)["Loss"]) for s in range(8)]
            ops.check_persistent_errors()
            if graph:
                assert runner.captures == 1 and runner.replays == 5 and len(runner.graphs[next(iter(runner.graphs))]["graphs"]) == 2
            results.append((losses, opt.flat.clone()))
            # replicas hold the same parameters: every rank applied the same averaged gradients
            mine = opt.flat.double().cpu()
            other = mine.clone()
            dist.broadcast(other, 0)
            assert torch.equal(mine, other) or (mine - other).abs().max().item() == 0.0
        (le, pe), (lg, pg) = results
        for a, b in zip(le, lg):
            assert abs(a - b) <= 2e-5 * abs(a), (le, lg)
        assert (pe - pg).abs().max().item() <= 2e-4 * pe.abs().max().item()
        dist.barrier()
        faulthandler.cancel_dump_traceback_later()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "FAIL %r" % (e,)))
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_captured_step_equals_two_rank_eager_step():
    """data parallel behind HIP graphs (train.GraphedTrainer's split mode: [zero_grad + forward + backward] |
    eager all-reduce of the live buckets | [norm + clip + Adam]) against the hook-driven eager schedule on the same
    two ranks: same losses over 8 steps (5 of them replays), same parameters, replicas identical"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_graph_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=170) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
