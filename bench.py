#!/usr/bin/env python
"""bench.py -- train sentences/sec (fwd+bwd+clip+Adam) of the full DSS-VAE on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY 8d "C2"): full DSS-VAE (registry key SyntaxVAE2 =
EnhanceSyntaxVAE: semantic + syntactic latents, linearised-tree decoder, BoW + adversarial heads),
vocab 30000 / tree vocab 300, batch 256 per GPU, src len 30, tree len 90, default dims of the shipped
yaml (E=H=300, 2-layer bi-GRU, Z=100), dropout as in the yaml, train_iter past warm_up so that the
adversarial term is in the loss.  One step = zero_grad -> get_loss -> mean(Loss).backward() ->
(gradient all-reduce over RCCL when N>1) -> global-norm clip 5.0 -> Adam, all in fp32.
Inputs are tensorised once and are resident in HBM before the timed region.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel of the step, its launches timed
live with HIP events on the stream they run on; `kernels` lists the other hot kernels the same way;
`cpu_baseline` is the CPU oracle (oracle/, the restatement of the reference pinned by the golden
vectors) timed on this box's host cores on a bounded sample of the same workload (N=1 only).

The timed region is the product's training step as `bin/train.py` runs it: `train.GraphedTrainer`, i.e. after two
eager steps the whole iteration is one captured HIP graph, replayed once per step (`--no-graph`: eager launches).
Event records cannot be read back from inside a replayed graph, so the per-kernel durations behind `roofline` /
`kernels` come from a second pass over the same K steps, launched eagerly with the events on (`with_kernel_events`:
its ms/step; `eager`: the same without events), right after the timed region.  `host_enqueue_ms_per_step` is the
host's share of a timed step (every launch of the K steps enqueued, before the final sync), given for every pass.
`driver`: sentences/s of `dss_vae_amd.bin.train.training()` itself -- files in the reference's formats, id cache,
shuffled batches cut on the device, the same captured step -- over a synthetic `train.bin` of the same shape
(`--mode driver`: BASELINE configs[2]'s 100 k sentences per epoch as the one JSON line).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
PEAK_HBM_GBS = 8000.0         # HBM3E spec; ~6.3 TB/s achievable

KERNEL_BOUND = {
    "linear_lse_kernel": "mfma", "linear_dlogits_kernel": "mfma", "gemm_store_kernel": "mfma",
    "gemm_group_kernel": "mfma",
    "gru_fwd_kernel": "hbm", "gru_bwd_kernel": "hbm", "adam_kernel": "hbm",
}


def pmc_traffic():
    """HBM-side bytes per launch per kernel from the rocprofv3 PMC passes of this same command
    (FETCH_SIZE and WRITE_SIZE in separate runs, corrected as MI355X_MICROARCH.md prescribes:
    (2*FETCH_SIZE + WRITE_SIZE) KiB; tools/pmc_summary.py), newest profiles/*pmc_traffic.json."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return {}, None
    with open(files[-1]) as fh:
        return json.load(fh), os.path.relpath(files[-1], ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="SyntaxVAE2", choices=["SyntaxVAE2", "SyntaxVAE", "VanillaVAE", "DisentangleVAE"])
    ap.add_argument("--batch", type=int, default=256, help="sentences per GPU")
    ap.add_argument("--vocab", type=int, default=30000)
    ap.add_argument("--tree-vocab", type=int, default=300)
    ap.add_argument("--src-len", type=int, default=30)
    ap.add_argument("--tree-len", type=int, default=90)
    ap.add_argument("--variable-lengths", action="store_true")
    ap.add_argument("--stress", action="store_true", help="BASELINE configs[3]: hidden 1024, latent 256")
    ap.add_argument("--mode", default="train", choices=["train", "generate", "driver"],
                    help="train = the headline metric; generate = BASELINE configs[4] (prior sampling + syntax "
                         "transfer greedy decode), a secondary line; driver = bin/train.py's training() over a "
                         "synthetic train.bin (BASELINE configs[2]: 100 k sentences per epoch)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of the captured HIP graph")
    ap.add_argument("--driver-sentences", type=int, default=None,
                    help="size of the synthetic train.bin (default: 100000 in --mode driver, 16384 for the short "
                         "driver leg of --mode train; 0 skips that leg)")
    ap.add_argument("--adv-train", action="store_true",
                    help="secondary: the 3-pass adversarial iteration (dis-syn step, dis-sem step, main step)")
    ap.add_argument("--gen-sentences", type=int, default=10000)
    ap.add_argument("--gen-batch", type=int, default=50, help="eval_bs (reference default 50)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product); gloo only to rehearse the N>1 code path on one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--shapes", action="store_true", help="development aid: per-shape kernel table on stderr")
    ap.add_argument("--cpu-batch", type=int, default=256, help="the GPU workload's batch: ~15 s of host work in all")
    ap.add_argument("--cpu-steps", type=int, default=3)
    return ap.parse_args()


def cpu_baseline(a, margs_dict):
    """the oracle (kind "port"), same model / data generator / step order, on the host cores."""
    import copy
    from argparse import Namespace
    import torch
    import oracle
    threads = torch.get_num_threads()
    args = Namespace(**copy.deepcopy(margs_dict))
    src, tgt = oracle.OVocabEntry(), oracle.OVocabEntry()
    for i in range(a.vocab - 4):
        src.add("w%d" % i)
    for i in range(a.tree_vocab - 4):
        tgt.add("t%d" % i)
    om = oracle.ORACLE_MODELS[a.model](args, oracle.OVocab(src, tgt))
    oracle.init_uniform_(om, seed=0)
    om.train()
    from dss_vae_amd import synthetic
    exs = [oracle.OExample(s, t) for s, t in synthetic.word_lists(
        a.cpu_batch, a.vocab, a.tree_vocab, src_len=a.src_len, tree_len=a.tree_len, fixed=not a.variable_lengths)]
    opt = torch.optim.Adam(om.parameters(), lr=0.0005, betas=(0.9, 0.995))
    oracle.oracle_train_step(om, opt, exs, 6000)  # warm-up
    ts = []
    for i in range(a.cpu_steps):
        t0 = time.perf_counter()
        oracle.oracle_train_step(om, opt, exs, 6001 + i)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": a.cpu_batch / med, "unit": "sentences/s", "cores": threads, "kind": "port",
            "sample": "%d timed steps (median) after 1 warm-up of the same model/config at batch %d "
                      "(CPU oracle, torch %s, %d threads; the oracle skips the reference's per-timestep topk + .cpu() "
                      "side work -- 275 calls per pass -- so the reference itself is a little slower than this)"
                      % (a.cpu_steps, a.cpu_batch, torch.__version__, threads),
            "s_per_step": med}


def cpu_generate_baseline(a, margs_dict, n_batches=12):
    """the oracle's prior sampling + greedy decode (kind "port": base_vae.py:69-85 through the CPU restatement) on the
    host cores, a bounded sample of the same workload: n_batches batches of eval_bs sentences, 1 warm-up batch"""
    import copy
    from argparse import Namespace
    import torch
    import oracle
    threads = torch.get_num_threads()
    src, tgt = oracle.OVocabEntry(), oracle.OVocabEntry()
    for i in range(a.vocab - 4):
        src.add("w%d" % i)
    for i in range(a.tree_vocab - 4):
        tgt.add("t%d" % i)
    om = oracle.ORACLE_MODELS[a.model](Namespace(**copy.deepcopy(margs_dict)), oracle.OVocab(src, tgt))
    oracle.init_uniform_(om, seed=0)
    om.eval()
    bs = a.gen_batch

    def one():
        with torch.no_grad():
            return om.decode_to_sentence(om.latent_for_init(om.sample_latent(bs)))[0]

    one()
    t0 = time.perf_counter()
    n = 0
    for _ in range(n_batches):
        n += len(one())
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "sentences/s", "cores": threads, "kind": "port",
            "sample": "%d batches of %d prior samples, greedy decode of %d steps, tokens to words (CPU oracle, torch %s, "
                      "%d threads), after 1 warm-up batch" % (n_batches, bs, margs_dict["src_max_time_step"],
                                                              torch.__version__, threads)}


def generate_bench(a, model, vocab, margs, rank, world, margs_dict=None):
    """BASELINE configs[4]: decoder-only generation.  (1) prior sampling: `generating(n, eval_bs)`
    (base_vae.py:69-85), always max_len = 35 greedy steps, tokens -> words on the host as the reference
    does; (2) syntax transfer: `eval_adv(sem_in, syn_ref)` (enhance_syntax_vae.py:485-497) = two encodes,
    one sentence decode and two 100-step tree decodes per pair batch.  Replicas only: each rank
    generates its share, no collective."""
    import torch
    from dss_vae_amd import synthetic
    model.eval()
    n, bs = a.gen_sentences // world, a.gen_batch
    torch.manual_seed(7 + rank)
    out = {}
    with torch.no_grad():
        import ctypes as C
        from dss_vae_amd import lib as L
        lib = L.load()
        model.generating(bs, bs)  # warm-up
        torch.cuda.synchronize()
        # `value` comes from a pass without instrumentation: the two event records per step that time the projection
        # launches each put ~5 us of idle stream beside the launch (rocprofv3 timeline), a seventh of a 75 us step.
        # The roofline's kernel time comes from a second, instrumented pass over the same work.
        lib.dssvae_decode_profile(0)
        t0 = time.perf_counter()
        res = model.generating(n, bs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["prior_sampling"] = {"sentences": len(res), "seconds": dt, "sentences_per_s": len(res) / dt}
        if bs != 1000:
            # the same sampling at eval_bs 1000 (the large decode batch: tiled GEMM + LSE kernels instead of the
            # decode-batch kernels), reported beside the reference's default batch
            model.generating(1000, 1000)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res_big = model.generating(max(n, 1000), 1000)
            torch.cuda.synchronize()
            dt_big = time.perf_counter() - t0
            out["prior_sampling_eval_bs_1000"] = {"sentences": len(res_big), "seconds": dt_big,
                                                  "sentences_per_s": len(res_big) / dt_big}
        tot_ms, launches = C.c_double(0.0), C.c_int(0)
        if not a.no_kernel_events:
            lib.dssvae_decode_profile(1)  # HIP events around every projection + arg-max launch
            t0 = time.perf_counter()
            model.generating(n, bs)
            torch.cuda.synchronize()
            dt_ev = time.perf_counter() - t0
            L.check(lib.dssvae_decode_profile_read(C.byref(tot_ms), C.byref(launches)), "dssvae_decode_profile_read")
            lib.dssvae_decode_profile(0)
            out["prior_sampling"]["sentences_per_s_with_kernel_events"] = len(res) / dt_ev
            dt = dt_ev  # (share_of_wall below is a share of the instrumented pass)
        if launches.value:
            # dominant kernel of the loop: [h ; e] W_out^T fused with the arg-max.  Measured bound (tools/gen_clock.py,
            # development builds without the loads / without the MFMAs: 40 us of matrix work, 26 us of stream, 43 us
            # together): the fp32 MFMA rate on the decode batch padded to the kernel's row groups.  Algorithmic flops
            # per launch: the UNPADDED batch; algorithmic bytes: W_out once (4 V (H + E)), the activations and
            # partials are noise beside it.
            dec = model.decoder
            H, E, V = dec.hidden_size, dec.input_size, dec.vocab_size
            avg_ms = tot_ms.value / launches.value
            nbytes = 4.0 * V * (H + E)
            group = lib.dssvae_skinny_argmax_rows(bs, V)
            rows = lib.dssvae_skinny_argmax_padded_rows(bs, V)   # (eval_bs 50: 48 rows in MFMA tiles + 2 on the vector ALUs)
            flops, flops_padded = 2.0 * bs * V * (H + E), 2.0 * rows * V * (H + E)
            gbs = nbytes / (avg_ms * 1e-3) / 1e9
            tf, tf_padded = flops / (avg_ms * 1e-3) / 1e12, flops_padded / (avg_ms * 1e-3) / 1e12
            out["roofline"] = {"kernel": "skinny_argmax64_kernel" if group == 64 else "skinny_kernel<ARGMAX>",
                               "bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                               "algorithmic_flops_per_launch": flops, "avg_launch_us": 1e3 * avg_ms,
                               "launches": launches.value, "share_of_wall": tot_ms.value * 1e-3 / dt,
                               "padded_rows": {"rows": rows, "flops_per_launch": flops_padded, "tflops": tf_padded,
                                               "frac": tf_padded / PEAK_F32_MFMA_TFLOPS},
                               "hbm": {"algorithmic_bytes_per_launch": nbytes, "achieved_GBps": gbs,
                                       "frac": gbs / PEAK_HBM_GBS}}
        if a.model == "SyntaxVAE2":
            exs = synthetic.examples(bs, a.vocab, a.tree_vocab, src_len=a.src_len, tree_len=a.tree_len, fixed=False,
                                     seed=99 + rank)
            ref = synthetic.examples(bs, a.vocab, a.tree_vocab, src_len=a.src_len, tree_len=a.tree_len, fixed=False,
                                     seed=199 + rank)
            model.eval_adv(exs, ref)
            torch.cuda.synchronize()
            reps = max(1, n // bs)
            t0 = time.perf_counter()
            for _ in range(reps):
                model.eval_adv(exs, ref)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            out["syntax_transfer"] = {"pairs": reps * bs, "seconds": dt2, "pairs_per_s": reps * bs / dt2}
    line = {"metric": "generation sentences/sec (greedy, prior sampling) DSS-VAE",
            "value": world * out["prior_sampling"]["sentences_per_s"],
            "unit": "sentences/s", "n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: %s decoder-only generation, %d prior samples per GPU at "
                                   "eval_bs %d, max_len %d, vocab %d" % (a.model, n, bs, margs.src_max_time_step,
                                                                         a.vocab)}}
    line.update(out)
    if rank == 0:
        if not a.no_cpu_baseline and world == 1 and margs_dict is not None:
            line["cpu_baseline"] = cpu_generate_baseline(a, margs_dict)
        print(json.dumps(line), flush=True)


def driver_bench(a, margs_dict, rank, world, n_sentences, steps):
    """sentences/s of bin/train.py's training() (the loop a user runs) over a synthetic train.bin of the workload's
    shape: vocab.bin / train.bin / dev.bin in the reference's pickle formats, the id cache, shuffled batches cut on
    the device, the captured step.  The clock starts after 6 iterations (model build, file loading, id conversion and
    the graph capture are one-time costs) and stops, synchronised, after `steps` more."""
    import shutil
    import tempfile
    from argparse import Namespace
    import torch
    import torch.distributed as dist
    from dss_vae_amd import synthetic
    from dss_vae_amd.bin.train import training
    from dss_vae_amd.structs import Dataset
    tmp = os.path.join(tempfile.gettempdir(), "dssvae_bench_%s" % os.environ.get("MASTER_PORT", str(os.getpid())))
    t_files = time.perf_counter()
    if rank == 0:
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        synthetic.vocab(a.vocab, a.tree_vocab).save(os.path.join(tmp, "vocab.bin"))
        Dataset(synthetic.examples(n_sentences, a.vocab, a.tree_vocab, src_len=a.src_len, tree_len=a.tree_len,
                                   fixed=not a.variable_lengths, seed=77)).save(os.path.join(tmp, "train.bin"))
        Dataset(synthetic.examples(64, a.vocab, a.tree_vocab, src_len=a.src_len, tree_len=a.tree_len,
                                   fixed=not a.variable_lengths, seed=78)).save(os.path.join(tmp, "dev.bin"))
    if world > 1:
        dist.barrier()
    t_files = time.perf_counter() - t_files
    skip = 6
    per_epoch = (n_sentences + a.batch * world - 1) // (a.batch * world)
    steps = min(steps, max(1, per_epoch - skip)) if steps else max(1, per_epoch - skip)
    main_args = Namespace(train_file=os.path.join(tmp, "train.bin"), dev_file=os.path.join(tmp, "dev.bin"),
                          vocab=os.path.join(tmp, "vocab.bin"), eval_lists=None, logdir=os.path.join(tmp, "logs"),
                          model_dir=os.path.join(tmp, "models"), exp_name="bench", task_type=a.model, start_iter=6000,
                          batch_size=a.batch, log_every=50, dev_every=10 ** 9, lr=0.0005, lr_decay=0.8, patience=8,
                          clip_grad=5.0, reset_optimizer=False, mode="train", seed=0, bench_skip=skip)
    import contextlib
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sys.stderr):   # the loop's progress lines: stdout carries the one JSON line only
        out = training(main_args, Namespace(**margs_dict), max_iters=skip + steps)
    wall = time.perf_counter() - t0
    tm, runner = out["timed"], out["runner"]
    dt = tm["seconds"]
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
    res = {"sentences_per_s": world * tm["sentences"] / dt, "ms_per_step": 1e3 * dt / tm["steps"], "steps": tm["steps"],
           "train_bin_sentences": n_sentences, "batch_per_gpu": a.batch,
           "graph_replays": runner.replays, "eager_steps": runner.eager, "graphs_captured": runner.captures,
           "one_time_s": {"synthetic_files_written": t_files, "training_call_minus_timed_steps": wall - tm["seconds"]}}
    del out
    torch.cuda.empty_cache()
    if rank == 0:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.gpus > 1 and world == 1:
        # plain `python bench.py --gpus N`: start the N ranks as CHILD processes (torch.distributed.run, one per
        # GPU) before anything here has touched the GPU, pass their output through and exit with their code
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path for the product)"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    from dss_vae_amd import ops, synthetic
    from dss_vae_amd.models import MODEL_CLS
    from dss_vae_amd.train import FlatAdam, GraphedTrainer, build_model_params_, train_iteration
    from dss_vae_amd.vae.base_vae import DeviceBatch

    over = dict(synthetic.STRESS_OVERRIDES) if a.stress else {}
    if a.model in ("VanillaVAE", "DisentangleVAE"):
        over["tgt_wd"] = False  # the reference's tgt_wd path of VanillaVAE does not run (vanilla_vae.py:112)
    margs = synthetic.model_args(model_select=a.model, task_type=a.model, **over)
    margs_dict = dict(vars(margs))
    if a.mode == "driver":
        n_sent = 100000 if a.driver_sentences is None else a.driver_sentences
        res = driver_bench(a, margs_dict, rank, world, n_sent, 0)
        if rank == 0:
            print(json.dumps({
                "metric": "train sentences/sec (fwd+bwd) DSS-VAE", "value": res["sentences_per_s"],
                "unit": "sentences/s", "n_gpus": world, "steps": res["steps"], "warmup": 6,
                "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": "BASELINE configs[2]: bin/train.py training() over a synthetic train.bin of %d "
                                       "sentences (%s, vocab %d, tree vocab %d, src len %d, tree len %d%s), batch %d "
                                       "per GPU, one epoch minus the first 6 iterations"
                                       % (n_sent, a.model, a.vocab, a.tree_vocab, a.src_len, a.tree_len,
                                          " (variable)" if a.variable_lengths else " (fixed)", a.batch),
                           "global_batch": a.batch * world, "parallelism": "dp%d" % world},
                "driver": res}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    vocab = synthetic.vocab(a.vocab, a.tree_vocab)
    model = MODEL_CLS[a.model](args=margs, vocab=vocab)
    build_model_params_(model, seed=0)
    model.cuda().train()
    n_params = sum(p.numel() for p in model.parameters())
    opt = FlatAdam(model, lr=0.0005, betas=(0.9, 0.995), clip_grad=5.0)
    if a.mode == "generate":
        return generate_bench(a, model, vocab, margs, rank, world, margs_dict)
    exs = synthetic.examples(a.batch, a.vocab, a.tree_vocab, src_len=a.src_len, tree_len=a.tree_len,
                             fixed=not a.variable_lengths, seed=1234 + rank)
    batch = DeviceBatch(exs, vocab)
    torch.manual_seed(1 + rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    it = 6000
    runner = GraphedTrainer(model, opt, a.adv_train, a.adv_train, enabled=not a.no_graph)
    for _ in range(a.warmup):
        it += 1
        runner(batch, it)
    extra_warmup = 0
    while runner.enabled and runner.captures == 0 and extra_warmup < 4:
        # the step is captured at its third call: fewer warm-up steps than that are topped up, untimed
        it += 1
        extra_warmup += 1
        runner(batch, it)
    sync()
    ops.check_persistent_errors()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        it += 1
        ret = runner(batch, it)
    host_dt = time.perf_counter() - t0   # every launch of the K steps is enqueued: the host's share of a step
    sync()
    dt = time.perf_counter() - t0
    replays_timed = runner.replays
    ops.check_persistent_errors()

    def eager_pass(with_events):
        nonlocal it
        tm = ops.KernelTimer() if with_events else None
        ops.PROFILE = tm
        sync()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            it += 1
            train_iteration(model, opt, batch, it, a.adv_train, a.adv_train)
        h = time.perf_counter() - t1
        sync()
        d = time.perf_counter() - t1
        ops.PROFILE = None
        ops.check_persistent_errors()
        return tm, d, h

    # the same K steps launched eagerly: with the per-kernel events (the live roofline measurement), and without
    timer, dt_events, host_events = eager_pass(True) if not a.no_kernel_events else (None, None, None)
    _, dt_plain, host_plain = eager_pass(False) if (runner.enabled and not a.no_kernel_events) else (None, None, None)
    if a.shapes:
        sys.stderr.write("gradients adopted in place %d, copied into the flat buffer %d (all steps)\n"
                         % (opt.adopted, opt.copied))
    loss = float(torch.mean(ret["Loss"]).item())
    if world > 1:
        t = torch.tensor([dt, dt_plain or 0.0, dt_events or 0.0], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        dt_plain = float(t[1].item()) if dt_plain is not None else None
        dt_events = float(t[2].item()) if dt_events is not None else None

    out = None
    if rank == 0:
        out = {
            "metric": "train sentences/sec (fwd+bwd) DSS-VAE", "value": a.batch * world * a.steps / dt,
            "unit": "sentences/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "host_enqueue_ms_per_step": 1e3 * host_dt / a.steps,
            "config": {"workload": "BASELINE configs[%d]: full DSS-VAE (%s) %sfwd+bwd+clip+Adam, vocab %d, tree vocab %d, "
                                   "src len %d, tree len %d%s, batch %d per GPU, E=%d H=%d Z=%d, fp32"
                                   % (3 if a.stress else 1, a.model,
                                      "3-pass adversarial iteration (dis-syn, dis-sem, main), each " if a.adv_train else "",
                                      a.vocab, a.tree_vocab, a.src_len, a.tree_len,
                                      " (variable)" if a.variable_lengths else " (fixed)", a.batch, margs.embed_size,
                                      margs.enc_hidden_dim, margs.latent_size),
                       "global_batch": a.batch * world, "parallelism": "dp%d" % world, "params": n_params},
            "final_loss": loss,
            "launch": {"mode": "hip_graph" if replays_timed else "eager", "graph_replays": replays_timed,
                       "eager_steps": runner.eager, "graphs_captured": runner.captures,
                       "extra_warmup": extra_warmup},
        }
        if dt_events:
            out["with_kernel_events"] = {"ms_per_step": 1e3 * dt_events / a.steps,
                                         "value": a.batch * world * a.steps / dt_events, "steps": a.steps,
                                         "host_enqueue_ms_per_step": 1e3 * host_events / a.steps}
        if dt_plain:
            out["eager"] = {"ms_per_step": 1e3 * dt_plain / a.steps,
                            "value": a.batch * world * a.steps / dt_plain, "steps": a.steps,
                            "host_enqueue_ms_per_step": 1e3 * host_plain / a.steps}
        if timer is not None:
            summ = timer.summary()
            if a.shapes:
                for name, detail, n, ms, tf in timer.by_shape():
                    sys.stderr.write("%-22s %-44s x%5.1f/step %8.3f ms/step %7.1f TFLOP/s\n"
                                     % (name, detail, n / a.steps, ms / a.steps, tf))
            default_cfg = (a.model, a.batch, a.vocab, a.tree_vocab, a.src_len, a.tree_len, a.variable_lengths,
                           a.stress) == ("SyntaxVAE2", 256, 30000, 300, 30, 90, False, False)
            traffic, traffic_src = pmc_traffic() if default_cfg else ({}, None)
            kernels = []
            for name, d in summ.items():
                bound = KERNEL_BOUND.get(name, "hbm")
                if bound == "mfma":
                    ach = d["flops"] / (d["total_ms"] * 1e-3) / 1e12
                    peak, unit = PEAK_F32_MFMA_TFLOPS, "TFLOP/s"
                else:
                    ach = d["bytes"] / (d["total_ms"] * 1e-3) / 1e9
                    peak, unit = PEAK_HBM_GBS, "GB/s"
                kernels.append({"kernel": name, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                                "frac": ach / peak,
                                "traffic": traffic.get(name, {}).get("hbm_bytes_per_launch"),
                                "algorithmic_per_launch": (d["flops"] if bound == "mfma" else d["bytes"]) / d["launches"],
                                "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                                "tflops": d["flops"] / (d["total_ms"] * 1e-3) / 1e12,
                                "mfma_frac": d["flops"] / (d["total_ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                "hbm_frac": d["bytes"] / (d["total_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                "us_per_timestep": (1e3 * d["total_ms"] / d["timesteps"]) if d.get("timesteps") else None,
                                "launches_per_step": d["launches"] / a.steps,
                                "avg_launch_ms": d["avg_ms"], "ms_per_step": d["total_ms"] / a.steps})
            kernels.sort(key=lambda k: -k["ms_per_step"])
            if kernels:
                out["roofline"] = kernels[0]
                out["kernels"] = kernels[1:]
                out["timed_kernel_ms_per_step"] = sum(k["ms_per_step"] for k in kernels)
                out["traffic_source"] = traffic_src
    n_drv = 16384 if a.driver_sentences is None else a.driver_sentences
    if n_drv and not a.adv_train and not a.stress and world == 1:
        del runner
        drv = driver_bench(a, margs_dict, rank, world, n_drv, 40)
        if rank == 0:
            out["driver"] = drv
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:
            del model, opt, batch
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(a, margs_dict)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
