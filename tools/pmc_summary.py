#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes per kernel.

usage: pmc_summary.py OUT.json COUNTER=dir_or_csv [COUNTER=dir_or_csv ...]

For every counter pass (one rocprofv3 run each: FETCH_SIZE and WRITE_SIZE do not fit in one pass on
gfx950) average the counter over the dispatches of each of our kernels.  `hbm_bytes_per_launch` =
(2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and FETCH_SIZE reads half the bytes of
wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM)."""
import csv
import glob
import json
import os
import re
import sys

NAMES = ["gemm_store_kernel", "gemm_group_kernel", "dssvae_zero2d_kernel", "linear_lse_kernel", "linear_dlogits_kernel", "gru_fwd_kernel", "gru_bwd_kernel",
         "adam_kernel", "sumsq_kernel", "softmax_grad_kernel", "colsum_kernel", "lse_combine_kernel",
         "picked_logit_kernel", "onehot_sub_kernel", "embed_scatter_kernel", "embed_gather_kernel",
         "reparam_kl_fwd_kernel", "reparam_kl_bwd_kernel", "greedy_decode_kernel", "skinny_kernel<true", "skinny_kernel<false",
         "skinny_gru_kernel", "skinny_argmax64_kernel", "combine_gather_kernel", "ce_rowscale_kernel", "ce_wgrad_cols_kernel", "ce_onehot_fixup_kernel", "ce_count_kernel",
         "ce_row_nll_kernel", "ce_group_sum_kernel", "dropout_apply_kernel", "dssvae_zero_many_kernel"]
PAT = re.compile("|".join(NAMES))
csv.field_size_limit(1 << 30)


def load(path):
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"),
                                                          recursive=True)
    per = {}
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                # (the training path's LSE kernel is reported with the other instantiations: bench.py times both as one)
                m = PAT.search(row["Kernel_Name"].replace("linear_lse_norm_kernel", "linear_lse_kernel"))
                if not m:
                    continue
                d = per.setdefault((m.group(0), row["Counter_Name"]), {})
                # one row per (dispatch, counter instance): sum instances of a dispatch
                d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    return per


def main():
    out_path, specs = sys.argv[1], sys.argv[2:]
    res = {}
    for spec in specs:
        _, path = spec.split("=", 1)
        for (kern, cname), disp in load(path).items():
            k = res.setdefault(kern, {})
            k[cname] = sum(disp.values()) / len(disp)
            k["launches_" + cname] = len(disp)
    for kern, k in res.items():
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            k["hbm_bytes_per_launch"] = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in k and k.get("SQ_BUSY_CU_CYCLES"):
            # matrix-pipe utilisation while the kernel has CUs busy (4 SIMDs per CU); share of wave cycles spent waiting
            k["mfma_pipe_utilisation"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * k["SQ_BUSY_CU_CYCLES"])
        if "SQ_WAIT_INST_ANY" in k and k.get("SQ_WAVE_CYCLES"):
            k["wave_cycles_waiting"] = k["SQ_WAIT_INST_ANY"] / k["SQ_WAVE_CYCLES"]
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    for kern in sorted(res):
        print(kern, json.dumps(res[kern], sort_keys=True))


if __name__ == "__main__":
    main()
