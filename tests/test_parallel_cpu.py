"""Data-parallel path on CPU with gloo, world_size 2: the flat-buffer bucket scheduler
(dss_vae_amd.parallel.GradBuckets) must give every rank the gradient of the single-process step on the
concatenated batch, whether buckets are sent from the backward hooks (overlap) or at finish()."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from dss_vae_amd.parallel import GradBuckets, shard_examples  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class Toy(torch.nn.Module):
    """shared embedding + three heads; `adv` is frozen half-way through forward (as EnhanceSyntaxVAE
    does with its adversaries), after the shared embedding has been read."""

    def __init__(self):
        super().__init__()
        self.emb = torch.nn.Embedding(50, 8)
        self.a = torch.nn.Linear(8, 16)
        self.b = torch.nn.Linear(16, 4)
        self.adv = torch.nn.Linear(16, 3)
        self.unused = torch.nn.Linear(4, 4)
        self.late = torch.nn.Linear(8, 1)

    def loss(self, ids, use_late=False):
        x = self.emb(ids).sum(1)
        if use_late:  # read early in forward => its gradient arrives at the very end of backward
            x = x + 0.01 * self.late(x)
        h = torch.tanh(self.a(x))
        main = self.b(h).pow(2).sum()
        for p in self.adv.parameters():
            p.requires_grad = False
        adv = self.adv(h).pow(2).sum()
        return (main - 0.1 * adv) / ids.size(0)


def _flatten(model):
    params = list(model.parameters())
    sizes = [(p.numel() + 3) // 4 * 4 for p in params]
    flat = torch.zeros(sum(sizes))
    offsets, o = [], 0
    for p, n in zip(params, sizes):
        p.grad = flat[o:o + p.numel()].view(p.shape)
        offsets.append(o)
        o += n
    return params, offsets, flat


def _worker(rank, world, port, overlap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = Toy()
        params, offsets, flat = _flatten(model)
        gb = GradBuckets(flat, params, offsets, bucket_elems=64, overlap=overlap)
        for i, p in enumerate(params):
            p.register_hook(lambda g, i=i: gb.intercept(i, g))
            p.register_post_accumulate_grad_hook(lambda _p, i=i: gb.mark_ready(i))
        g = torch.Generator().manual_seed(1)
        ids_all = torch.randint(0, 50, (3, 8, 5), generator=g)
        early, skipped = [], []
        for step in range(3):
            flat.zero_()
            gb.reset("main")
            ids = ids_all[step][rank::world]
            model.loss(ids).backward()
            gb.finish()
            early.append(gb.sent_early)
            skipped.append(gb.skipped)
            # single-process reference on the whole batch
            ref = Toy()
            ref.load_state_dict(model.state_dict())
            ref.loss(ids_all[step]).backward()
            for (n, p), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
                want = torch.zeros_like(p) if r.grad is None else r.grad
                assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6), (step, n)
        if overlap:
            assert early[0] == 0 and early[1] > 0 and early[2] > 0, early
        else:
            assert early == [0, 0, 0], early
        # the bucket that holds only parameters without a gradient (adv / unused / late) is learned in the
        # first step and not sent afterwards
        assert skipped[0] == 0 and skipped[1] > 0 and skipped[2] > 0, skipped
        # a gradient that shows up for a bucket that was already sent or skipped (the graph of this step differs
        # from the last one of its kind) is averaged on its own and lands in the right slice
        for rep in range(2):  # second time round the schedule expects it
            flat.zero_()
            gb.reset("main")
            model.loss(ids_all[0][rank::world], use_late=True).backward()
            n_late = len(gb.late)
            gb.finish()
            ref = Toy()
            ref.load_state_dict(model.state_dict())
            ref.loss(ids_all[0], use_late=True).backward()
            for (n, p), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
                want = torch.zeros_like(p) if r.grad is None else r.grad
                assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6), ("late", rep, n)
            if overlap:
                assert (n_late > 0) == (rep == 0), (rep, n_late)
            else:
                assert n_late == 0
        # a different kind of step learns its own schedule
        flat.zero_()
        gb.reset("with_late")
        model.loss(ids_all[0][rank::world], use_late=True).backward()
        gb.finish()
        assert gb.sent_early == 0
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "FAIL %r" % (e,)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_bucketed_gradient_average_world2(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_shard_examples_strided_and_sorted():
    class E(object):
        def __init__(self, n):
            self.src = ["w"] * n

    exs = [E(n) for n in (9, 8, 8, 7, 5, 4, 2)]
    parts = [shard_examples(exs, r, 3) for r in range(3)]
    assert [len(p) for p in parts] == [3, 2, 2]
    assert sorted(len(e.src) for p in parts for e in p) == sorted(len(e.src) for e in exs)
    for p in parts:
        lens = [len(e.src) for e in p]
        assert lens == sorted(lens, reverse=True)
