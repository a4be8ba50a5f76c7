"""Training-loop helpers (reference: dss_vae/utils/model_utils.py:16-30,52-97,108-196): loss tracker,
model directory layout, build_model (U(-0.08,0.08) re-init, Adam betas (0.9,0.995)), checkpoint +
patience / lr-decay schedule.  Differences from the reference, all host-side: no interactive prompt
when a log directory exists; the optimiser is the flat-buffer FlatAdam (same update rule)."""
import math
import os
import sys

import torch

from .config_funcs import args_to_dict, dict_to_yaml


def update_tracker(ret_loss, tracker):
    for key, val in ret_loss.items():
        if isinstance(val, torch.Tensor):
            cur = val.mean().data.unsqueeze(0)
            tracker[key] = torch.cat((tracker[key], cur)) if key in tracker else cur
        else:
            tracker[key] = tracker[key] + val if key in tracker else val
    return tracker


def get_model_info(main_args, model_args, check_dir=True):
    name = ".".join([model_args.model_select] + ([main_args.exp_name] if main_args.exp_name is not None else []))
    task_dir = os.path.join(main_args.model_dir, main_args.task_type)
    model_dir = os.path.join(task_dir, name)
    os.makedirs(model_dir, exist_ok=True)
    log_dir = os.path.join(main_args.logdir, "%s.%s" % (main_args.task_type, name))
    if check_dir and getattr(main_args, "mode", "train").lower().startswith("train"):
        dict_to_yaml(os.path.join(model_dir, "base_config.yaml"), {"base_configs": args_to_dict(main_args)})
        dict_to_yaml(os.path.join(model_dir, "model_config.yaml"), {"model_configs": args_to_dict(model_args)})
    return {"model_dir": model_dir, "log_dir": log_dir, "model_file": os.path.join(model_dir, "model.bin"),
            "out_dir": os.path.join(model_dir, "out")}


def build_model(main_args, model_args, model=None, seed=None):
    from ..models import init_create_model
    from ..structs import Vocab
    from ..train import FlatAdam, build_model_params_
    if model is None:
        vocab = Vocab.from_bin_file(main_args.vocab)
        model = init_create_model(model_args.model_select, args=model_args, vocab=vocab)
        build_model_params_(model, seed=seed)
    model.train()
    model.cuda()
    optimizer = FlatAdam(model, lr=main_args.lr, betas=(0.9, 0.995), clip_grad=main_args.clip_grad)
    return model, optimizer, model.vocab


def load_data(main_args):
    from ..structs import Dataset
    return Dataset.from_bin_file(main_args.train_file), Dataset.from_bin_file(main_args.dev_file)


def get_lr_schedule(is_better, model, optimizer, main_args, patience, num_trial, model_file, reload_model=True):
    if is_better:
        patience = 0
        print("save model to [%s]" % model_file, file=sys.stdout)
        model.save(model_file)
        torch.save(optimizer.state_dict(), model_file + ".optim.bin")
    elif patience < main_args.patience:
        patience += 1
        print("hit patience %d" % patience, file=sys.stdout)
    if patience == main_args.patience:
        num_trial += 1
        print("hit #%d trial" % num_trial, file=sys.stdout)
        lr = optimizer.param_groups[0]["lr"] * main_args.lr_decay
        print("decay learning rate to %f" % lr, file=sys.stdout)
        if reload_model:
            from ..structs.io import load_model_bin
            print("load previously best model", file=sys.stdout)
            model.load_state_dict(load_model_bin(model_file)["state_dict"])
        if not getattr(main_args, "reset_optimizer", False):
            print("restore parameters of the optimizers", file=sys.stdout)
            optimizer.load_state_dict(torch.load(model_file + ".optim.bin", weights_only=True))
        else:
            # the reference builds a fresh Adam here (utils/model_utils.py:80-82): moments and step counts restart
            print("reset optimizer", file=sys.stdout)
            optimizer.m.zero_()
            optimizer.v.zero_()
            optimizer.steps = [0] * len(optimizer.steps)
        for group in optimizer.param_groups:
            group["lr"] = lr
        patience = 0
    if optimizer.param_groups[0]["lr"] <= 1e-6:
        print("early stop!", file=sys.stdout)
    return model, optimizer, num_trial, patience


# ---- the rest of utils/model_utils.py's surface -------------------------------------------------
def lr_transformer(args, i, lr0=0.1):
    """utils/model_utils.py:33-34"""
    return lr0 * 10 / math.sqrt(args.d_model) * min(1 / math.sqrt(i), i / (args.warmup * math.sqrt(args.warmup)))


def get_lr_anneal(args, train_iter):
    """utils/model_utils.py:37-39"""
    lr_end = 1e-5
    return max(0, (args.lr - lr_end) * (args.anneal_steps - train_iter) / args.anneal_steps) + lr_end


def update_lr(args, opt, train_iter):
    """utils/model_utils.py:42-49.  The reference calls its two schedule functions without `args` (a TypeError
    for every schedule but "fixed"); here they get it."""
    if args.lr_schedule == "fixed":
        opt.param_groups[0]["lr"] = args.lr
    elif args.lr_schedule == "anneal":
        opt.param_groups[0]["lr"] = get_lr_anneal(args, train_iter + 1)
    elif args.lr_schedule == "transformer":
        opt.param_groups[0]["lr"] = lr_transformer(args, train_iter + 1)
    return opt


def get_eval_info(main_args, model_args, mode="Train"):
    """utils/model_utils.py:100-110: <model_dir>/<model>[.<exp>]/<mode>, created"""
    parts = [model_args.model_select] + ([main_args.exp_name] if main_args.exp_name is not None else [])
    model_dir = os.path.join(main_args.model_dir, ".".join(parts), mode)
    os.makedirs(model_dir, exist_ok=True)
    return model_dir


def load_configs(exp_dir):
    """utils/model_utils.py:199-204: the two yaml files get_model_info left beside a checkpoint"""
    from .config_funcs import dict_to_args, yaml_to_dict
    return (dict_to_args(yaml_to_dict(os.path.join(exp_dir, "base_config.yaml"))["base_configs"]),
            dict_to_args(yaml_to_dict(os.path.join(exp_dir, "model_config.yaml"))["model_configs"]))


def load_model(main_args, model_args, check_dir=True, use_cuda=None):
    """utils/model_utils.py:213-221: the checkpoint of an experiment -> model (on the MI355X unless use_cuda
    is False)"""
    from ..models import load_static_model
    use_cuda = getattr(model_args, "cuda", True) if use_cuda is None else use_cuda
    model_file = get_model_info(main_args=main_args, model_args=model_args, check_dir=check_dir)["model_file"]
    print("Load From:\t{}".format(model_file))
    model = load_static_model(model=model_args.model_select, model_path=model_file)
    if use_cuda:
        model.cuda()
    return model
