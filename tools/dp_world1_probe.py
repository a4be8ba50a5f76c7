#!/usr/bin/env python
"""Probe: the whole data-parallel code path (hooks, buckets, side stream, RCCL all-reduce AVG) at
world_size 1 on RCCL, compared with the plain single-process run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import test_dp_gpu as T
from dss_vae_amd.train import FlatAdam
os.environ["DSSVAE_DP_FORCE"] = "1"
model, exs, eps = T._setup()
opt = FlatAdam(model, bucket_mb=0.01)
assert opt.buckets is not None
l1 = T._run(model, opt, exs, eps, slice(None))
print("dp world1", l1, "sent_early", opt.buckets.sent_early, "buckets", len(opt.buckets.buckets), flush=True)
ref, exs, eps = T._setup()
ropt = FlatAdam(ref, data_parallel=False)
l2 = T._run(ref, ropt, exs, eps, slice(None))
print("plain    ", l2, flush=True)
sd, rsd = model.state_dict(), ref.state_dict()
worst = max((sd[k] - rsd[k]).abs().max().item() for k in rsd)
print("worst param diff", worst)
assert worst < 1e-6 and opt.buckets.sent_early > 0
# the captured step's split mode (two graphs around eager RCCL all-reduces) on the same forced 1-rank group
from dss_vae_amd import ops
from dss_vae_amd.train import GraphedTrainer
from dss_vae_amd.vae.base_vae import DeviceBatch
ops.CHECK_UNIFORM_UPSTREAM = False
res = []
for graph in (False, True):
    model, exs, _ = T._setup()
    opt = FlatAdam(model, bucket_mb=0.01)
    assert opt.buckets is not None
    torch.manual_seed(3)
    runner = GraphedTrainer(model, opt, enabled=graph)
    batch = DeviceBatch(exs, model.vocab)
    losses = [float(runner(batch, 6000 + s)["Loss"]) for s in range(8)]
    ops.check_persistent_errors()
    res.append((losses, opt.flat.clone(), runner))
(le, pe, _), (lg, pg, rg) = res
print("rccl eager", le, flush=True)
print("rccl graph", lg, "replays", rg.replays, flush=True)
assert rg.replays == 5 and all(abs(a - b) <= 2e-5 * abs(a) for a, b in zip(le, lg))
assert (pe - pg).abs().max().item() <= 2e-4 * pe.abs().max().item()
dist.destroy_process_group()
print("ok")
