"""Training driver (reference: dss_vae/bin/train.py:36-201): same yaml keys, step order,
optional 3-pass adversarial schedule (`adv_train`), dev evaluation every `dev_every` past `warm_up`,
checkpoint on improvement, patience / lr decay.  One process per GPU under torch.distributed.run:
every rank takes a strided share of each batch and gradients are averaged over RCCL.

    python -m dss_vae_amd.bin.train --base_config B.yaml --model_config M.yaml [--max_iters N]

Differences from the reference, host-side only: `max_iters` ends the otherwise endless loop; scalars go
to <log_dir>/scalars.jsonl (and to tensorboardX when it is installed) under the reference's tags."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from .. import ops
from ..metrics import get_evaluator
from ..parallel import world_size
from ..structs.id_cache import IdCache
from ..train import GraphedTrainer
from ..utils.config_funcs import dict_to_args, yaml_to_dict
from ..utils.model_utils import build_model, get_lr_schedule, get_model_info, load_data, update_tracker

sort_key_dict = {"rnn": "src", "att": "src", "src": "src", "tgt": "tgt", "syn": "syn"}


class _Scalars(object):
    def __init__(self, log_dir, enabled):
        self.f, self.tb = None, None
        if not enabled:
            return
        os.makedirs(log_dir, exist_ok=True)
        self.f = open(os.path.join(log_dir, "scalars.jsonl"), "a")
        try:
            from tensorboardX import SummaryWriter
            self.tb = SummaryWriter(log_dir)
        except ImportError:
            pass

    def add_scalar(self, tag, scalar_value, global_step):
        if self.f is not None:
            self.f.write(json.dumps({"tag": tag, "value": float(scalar_value), "step": int(global_step)}) + "\n")
            self.f.flush()
        if self.tb is not None:
            self.tb.add_scalar(tag=tag, scalar_value=scalar_value, global_step=global_step)


def training(main_args, model_args, model=None, max_iters=None):
    if "task_type" in model_args and model_args.task_type is not None:
        main_args.task_type = model_args.task_type
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = world_size()
    dir_ret = get_model_info(main_args=main_args, model_args=model_args, check_dir=rank == 0)
    model, optimizer, vocab = build_model(main_args=main_args, model_args=model_args, model=model,
                                          seed=getattr(main_args, "seed", None))
    train_set, dev_set = load_data(main_args)
    # train.bin as flat ids + offsets, converted once (cached beside the pickle by rank 0) and resident in HBM: a
    # step's batch is a list of example indices, cut into padded id matrices on the device (structs/id_cache.py)
    cache = IdCache.for_file(main_args.train_file, train_set.examples, vocab, write=rank == 0).cuda()
    if world > 1:
        # replicas start from rank 0's parameters whatever the ranks' seeds were, and draw different noise
        dist.broadcast(optimizer.flat, 0)
        torch.manual_seed(torch.initial_seed() + 7919 * rank)
        from ..networks.rnn_base import GLOBAL_SEEDS
        for _ in range(rank):
            GLOBAL_SEEDS.next()
    model_file, out_dir = dir_ret["model_file"], dir_ret["out_dir"]
    writer = _Scalars(dir_ret["log_dir"], rank == 0)
    print("...... Start Training ......")
    train_iter = main_args.start_iter
    train_nums, train_loss = 0.0, 0.0
    epoch, num_trial, patience = 0, 0, 0
    history_scores = []
    task_type = main_args.task_type
    sort_key = sort_key_dict[model_args.sort_key] if "sort_key" in model_args else "src"
    evaluator = get_evaluator(eval_choice=task_type, model=model, eval_set=dev_set.examples,
                              eval_lists=getattr(main_args, "eval_lists", None), sort_key=sort_key, eval_tgt="tgt",
                              batch_size=model_args.eval_bs, out_dir=out_dir, write_down=rank == 0)
    print("Dev ITEM: ", evaluator.score_item)
    adv_training = "adv_train" in model_args and model_args.adv_train
    adv_syn = adv_sem = False
    if adv_training:  # hyper_init, bin/train.py:76-82
        adv_syn = (model_args.adv_syn + model_args.infer_weight * model_args.inf_sem) > 0.0
        adv_sem = (model_args.adv_sem + model_args.infer_weight * model_args.inf_syn) > 0.0
    # the iteration behind a HIP graph per batch shape (train.GraphedTrainer; DSSVAE_GRAPH=0: always eager)
    run_iteration = GraphedTrainer(model, optimizer, adv_syn, adv_sem)
    bench_skip = getattr(main_args, "bench_skip", None)   # bench.py --mode driver: start a clock after this many steps
    timed = None

    while True:
        epoch += 1
        epoch_begin = time.time()
        train_log_dict = {}
        rng = None
        if world > 1:  # all ranks cut the same global batches: rank 0's draw seeds the epoch's shuffle
            seed_t = torch.tensor([int(np.random.randint(0, 2 ** 31 - 1))], device="cuda", dtype=torch.int64)
            dist.broadcast(seed_t, 0)
            rng = np.random.RandomState(int(seed_t.item()))
        for n_global, batch_examples in cache.batch_iter(main_args.batch_size, shuffle=True, rng=rng, rank=rank,
                                                         world=world):
            if n_global < world:
                continue  # every rank must take part in the step's collectives: all ranks drop the same tail
            if bench_skip is not None and train_iter - main_args.start_iter == bench_skip:
                torch.cuda.synchronize()
                timed = [time.perf_counter(), train_iter, train_nums]
            train_iter += 1
            train_nums += len(batch_examples)
            batch_ret = run_iteration(batch_examples, train_iter, global_batch=n_global)
            batch_loss = batch_ret["Loss"]
            train_log_dict = update_tracker(batch_ret, train_log_dict)
            if train_iter % main_args.log_every == 0:
                ops.check_persistent_errors()
                train_loss += batch_loss.detach().sum().item() * main_args.log_every  # sampled: no per-step sync
                print("\r[Iter %d] Train loss=%.5f" % (train_iter, train_loss / train_nums), file=sys.stdout, end=" ")
                keys = sorted(k for k, v in train_log_dict.items() if isinstance(v, torch.Tensor))
                if keys:
                    # one device->host copy for all terms; under data parallel one tiny all-reduce first, so
                    # the logged terms are those of the global batch (same keys on all ranks: same model)
                    vals = torch.stack([torch.mean(train_log_dict[k].detach().float()) for k in keys]).double()
                    if world > 1:
                        dist.all_reduce(vals)
                        vals /= world
                    for key, val in zip(keys, vals.tolist()):
                        writer.add_scalar("{}/Train/{}".format(task_type, key), val, train_iter)
                writer.add_scalar("Optimize/lr", optimizer.param_groups[0]["lr"], train_iter)
                writer.add_scalar("Optimize/trial", num_trial, train_iter)
                writer.add_scalar("Optimize/patience", patience, train_iter)
                train_log_dict = {}
            if train_iter % main_args.dev_every == 0 and train_iter > model_args.warm_up:
                eval_result_dict = evaluator()
                dev_acc = eval_result_dict[evaluator.score_item]
                if isinstance(dev_acc, torch.Tensor):
                    dev_acc = dev_acc.sum().item()
                if world > 1:  # rank 0's score (the one whose files were written) drives every replica's history
                    acc_t = torch.tensor([float(dev_acc)], device="cuda", dtype=torch.float64)
                    dist.broadcast(acc_t, 0)
                    dev_acc = float(acc_t.item())
                print("\r[Iter %d] %s %s=%.3f took %d s" % (train_iter, task_type, evaluator.score_item, dev_acc,
                                                            eval_result_dict["EVAL TIME"]), file=sys.stdout)
                is_better = (history_scores == []) or dev_acc > max(history_scores)
                history_scores.append(dev_acc)
                writer.add_scalar("%s/Valid/Best %s" % (task_type, evaluator.score_item), max(history_scores), train_iter)
                for key, val in eval_result_dict.items():
                    writer.add_scalar("{}/Valid/{}".format(task_type, key),
                                      val.sum().item() if isinstance(val, torch.Tensor) else val, train_iter)
                if rank == 0:
                    model, optimizer, num_trial, patience = get_lr_schedule(
                        is_better=is_better, model_file=model_file, main_args=main_args, patience=patience,
                        num_trial=num_trial, model=model, optimizer=optimizer, reload_model=False)
                if world > 1:  # every rank follows rank 0's schedule decision
                    t = torch.tensor([optimizer.param_groups[0]["lr"], num_trial, patience], device="cuda",
                                     dtype=torch.float64)
                    dist.broadcast(t, 0)
                    optimizer.param_groups[0]["lr"], num_trial, patience = float(t[0]), int(t[1]), int(t[2])
                    # ... including what it restored from the checkpoint (Adam moments and step counts on a
                    # decay): replicas stay bit-identical
                    for buf in (optimizer.flat, optimizer.m, optimizer.v):
                        dist.broadcast(buf, 0)
                    st = torch.tensor(optimizer.steps, device="cuda", dtype=torch.int64)
                    dist.broadcast(st, 0)
                    optimizer.steps = [int(x) for x in st.tolist()]
                model.train()
            if max_iters is not None and train_iter - main_args.start_iter >= max_iters:
                out = {"model": model, "optimizer": optimizer, "train_iter": train_iter, "history": history_scores,
                       "model_file": model_file, "log_dir": dir_ret["log_dir"], "runner": run_iteration}
                if timed is not None:
                    torch.cuda.synchronize()
                    out["timed"] = {"seconds": time.perf_counter() - timed[0], "steps": train_iter - timed[1],
                                    "sentences": train_nums - timed[2]}
                return out
        print("\r[Epoch %d] epoch elapsed %ds" % (epoch, time.time() - epoch_begin), file=sys.stdout)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base_config", required=True)
    ap.add_argument("--model_config", required=True)
    ap.add_argument("--exp_name", default=None)
    ap.add_argument("--mode", default="train")
    ap.add_argument("--max_iters", type=int, default=None)
    a = ap.parse_args()
    main_args = dict_to_args(yaml_to_dict(a.base_config)["base_configs"])
    model_args = dict_to_args(yaml_to_dict(a.model_config)["model_configs"])
    main_args.mode = a.mode
    if a.exp_name is not None:
        main_args.exp_name = a.exp_name
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    training(main_args, model_args, max_iters=a.max_iters)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
