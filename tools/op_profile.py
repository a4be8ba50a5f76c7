#!/usr/bin/env python
"""development aid: torch.profiler table of one training step (which host-side ops launch the small
kernels between the HIP launches).  usage: python tools/op_profile.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dss_vae_amd import synthetic  # noqa: E402
from dss_vae_amd.models import MODEL_CLS  # noqa: E402
from dss_vae_amd.train import FlatAdam, build_model_params_, train_iteration  # noqa: E402
from dss_vae_amd.vae.base_vae import DeviceBatch  # noqa: E402


def main():
    margs = synthetic.model_args(model_select="SyntaxVAE2", task_type="SyntaxVAE2")
    vocab = synthetic.vocab(30000, 300)
    model = MODEL_CLS["SyntaxVAE2"](args=margs, vocab=vocab)
    build_model_params_(model, seed=0)
    model.cuda().train()
    opt = FlatAdam(model, lr=0.0005, betas=(0.9, 0.995), clip_grad=5.0)
    batch = DeviceBatch(synthetic.examples(256, 30000, 300, src_len=30, tree_len=90, fixed=True, seed=1234), vocab)
    for i in range(3):
        train_iteration(model, opt, batch, 6001 + i)
    torch.cuda.synchronize()
    shapes = "--shapes" in sys.argv  # per input shape, by device time: which tensors the torch glue ops touch
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=shapes) as prof:
        train_iteration(model, opt, batch, 6010)
        torch.cuda.synchronize()
    if "--counts" in sys.argv:  # which aten ops are launched most often (the tiny-kernel tail of the step)
        evs = sorted(prof.key_averages(), key=lambda e: -e.count)
        for e in evs[:45]:
            print("%-50s x%-5d self cuda %9.1f us" % (e.key[:50], e.count, e.self_device_time_total))
    elif shapes:
        print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=60,
                                                                 max_name_column_width=40, max_shapes_column_width=70))
    else:
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))


if __name__ == "__main__":
    main()
