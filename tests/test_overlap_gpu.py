"""The persistent GRU launches need every workgroup resident at once; under data parallel RCCL kernels occupy CUs on
the communication stream while backward is still launching recurrences (parallel.GradBuckets).  Rehearsal on one
GPU: the same training step quiet and beside a side stream that keeps the chip busy with MFMA GEMM workgroups --
results agree and no persistent launch gives up waiting (a late workgroup only delays its peers: spins are
bounded in the seconds, workgroups of other kernels retire in microseconds)."""
import pytest
import torch

from parity_util import make_args, make_examples, product_examples, product_vocab, rel_err

pytestmark = pytest.mark.gpu


def _step(pm, exs, noise, enqueued=None):
    for p in pm.parameters():
        p.grad = None
    pm.noise = dict(noise)
    ret = pm.get_loss(product_examples(exs), 6000)
    torch.mean(ret["Loss"]).backward()
    if enqueued is not None:
        enqueued()
    torch.cuda.synchronize()
    return float(ret["Loss"].detach()), {n: p.grad.detach().clone() for n, p in pm.named_parameters() if p.grad is not None}


def _model(n_src, n_tgt, B):
    from dss_vae_amd.models import MODEL_CLS
    from dss_vae_amd.train import build_model_params_
    args = make_args(dec_ed=0.0, dec_rd=0.0, tgt_wd=False)
    pm = MODEL_CLS["SyntaxVAE2"](args=args, vocab=product_vocab(n_src, n_tgt))
    build_model_params_(pm, seed=3)
    pm.cuda().train()
    exs = make_examples(B, n_src, n_tgt, src_len=30, tree_len=90, fixed=True, seed=11)
    gen = torch.Generator().manual_seed(5)
    noise = {"eps_syn": torch.randn(B, 100, generator=gen), "eps_sem": torch.randn(B, 100, generator=gen)}
    return pm, exs, noise


@pytest.mark.parametrize("workgroups,lds_kb", [(48, 32), (64, 64)])
def test_training_step_beside_a_resident_neighbour(workgroups, lds_kb):
    """the neighbour does not retire: `workgroups` workgroups stay on the chip (as a collective's kernel does while
    it waits for its peers) from before the forward until the host releases them once the whole backward is enqueued
    -- or until their own time limit, should the host block on the device before that (an allocator flush in a long
    test process does).  The step's large persistent launches hold a CU's LDS per workgroup (228 x 150 KB), so their
    last workgroups become resident only when the neighbour leaves: what is shown is that the part of the grid that IS
    resident spins for that long without giving up, that the launch then completes with the same results, and that
    the error word stays clean"""
    from dss_vae_amd import ops
    pm, exs, noise = _model(4000, 300, 256)
    loss_q, g_q = _step(pm, exs, noise)
    ops.check_persistent_errors()
    torch.cuda.synchronize()
    nb = ops.ResidentNeighbour(workgroups=workgroups, lds_bytes=lds_kb * 1024, max_ms=400)
    loss_b, g_b = _step(pm, exs, noise, enqueued=nb.release)
    held = nb.held_ms()
    ops.check_persistent_errors()  # raises if a recurrence gave up waiting for a peer
    assert len(held) == workgroups and min(held) > 1.0, held  # resident while the step ran, not queued behind it
    assert max(held) < 450.0, held  # the host's flag or the neighbour's own limit, whichever came first
    assert abs(loss_b - loss_q) <= 1e-5 * abs(loss_q)
    assert set(g_b) == set(g_q)
    for n in g_q:
        assert rel_err(g_b[n], g_q[n]) < 1e-4, (n, rel_err(g_b[n], g_q[n]))


def test_training_step_beside_a_busy_side_stream():
    from dss_vae_amd import ops
    from dss_vae_amd.models import MODEL_CLS
    from dss_vae_amd.train import build_model_params_
    n_src, n_tgt, B = 4000, 300, 256
    args = make_args(dec_ed=0.0, dec_rd=0.0, tgt_wd=False)
    pm = MODEL_CLS["SyntaxVAE2"](args=args, vocab=product_vocab(n_src, n_tgt))
    build_model_params_(pm, seed=3)
    pm.cuda().train()
    exs = make_examples(B, n_src, n_tgt, src_len=30, tree_len=90, fixed=True, seed=11)
    gen = torch.Generator().manual_seed(5)
    noise = {"eps_syn": torch.randn(B, 100, generator=gen), "eps_sem": torch.randn(B, 100, generator=gen)}
    loss_q, g_q = _step(pm, exs, noise)
    ops.check_persistent_errors()
    # the filler: ~60 ms of 128x128-tile GEMM workgroups (2 per CU, 74 KB of LDS each) on a side stream
    side = torch.cuda.Stream()
    a, b = torch.randn(4096, 4096, device="cuda"), torch.randn(4096, 4096, device="cuda")
    out = torch.empty(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(50):
            ops.mm_nt(a, b, out=out)
    loss_b, g_b = _step(pm, exs, noise)
    side.synchronize()
    ops.check_persistent_errors()  # raises if a recurrence gave up waiting for a peer
    assert abs(loss_b - loss_q) <= 1e-5 * abs(loss_q)
    assert set(g_b) == set(g_q)
    for n in g_q:
        assert rel_err(g_b[n], g_q[n]) < 1e-4, (n, rel_err(g_b[n], g_q[n]))
