"""bag_of_word_loss (reference: dss_vae/utils/loss_funcs.py:117-129) for callers that already hold the
per-sentence log-softmax row.  The models do not come through here: their BoW heads fuse the 30k-vocabulary
projection with the loss (vae/syntax_vae.py::BridgeMLP -> ops.linear_ce with T targets per row), so neither the
[B,V] row nor the reference's [B,T,V] expanded copy is ever written."""
import torch


def bag_of_word_loss(input_log_softmax, input_var, criterion=None):
    """input_log_softmax [B,1,V] or [B,V]; input_var [B,T] int64; criterion: an nn.NLLLoss (its ignore_index
    and reduction are honoured).  Same value as criterion(expand(row).view(-1,V), input_var.view(-1)) without
    the expanded copy: the row is gathered at the T targets."""
    row = input_log_softmax.squeeze(1) if input_log_softmax.dim() == 3 else input_log_softmax
    ignore = getattr(criterion, "ignore_index", -100)
    reduction = getattr(criterion, "reduction", "mean")
    keep = input_var.ne(ignore)
    picked = row.gather(1, input_var.clamp(min=0)) * keep.to(row.dtype)
    if reduction == "sum":
        return -picked.sum()
    if reduction == "mean":
        return -picked.sum() / keep.sum().clamp(min=1).to(row.dtype)
    return (-picked).view(-1)
