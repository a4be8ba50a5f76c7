"""The loss dictionary as rows of coefficients (vae/enhance_syntax_vae.py `_Assembly`): the host-side algebra, with
the device op replaced by its definition (out_j = sum_k coef[j][k] * sum(x_k)).  The op itself is held to fp64 on the
GPU (tests/test_kernels_gpu.py::test_weighted_sums_loss_assembly), the model's dictionary to the golden vectors."""
import torch

from dss_vae_amd import ops
from dss_vae_amd.vae.enhance_syntax_vae import _Assembly, _Row


def _definition(coef, xs, coef_dyn=None, dyn=None):
    sums = torch.stack([x.double().sum() for x in xs])
    c = torch.tensor(coef, dtype=torch.float64)
    if coef_dyn is not None and dyn is not None:
        c = c + torch.tensor(coef_dyn, dtype=torch.float64) * float(dyn)
    return tuple(c @ sums)


def test_rows_compose_linearly(monkeypatch):
    monkeypatch.setattr(ops, "weighted_sums", _definition)
    g = torch.Generator().manual_seed(0)
    rec, tree, bow, kl = (torch.randn(7, generator=g), torch.randn(7, generator=g), torch.randn(7, generator=g),
                          torch.randn((), generator=g))
    asm = _Assembly()
    nll = asm.term((-1.0, rec), 1.0)
    mul = asm.term((-1.0, tree), 0.5) + asm.term((1.0, bow), 2.0)
    off = asm.term(None, 3.0)                       # a head that is switched off contributes nothing
    assert off == _Row() and len(asm.parts) == 3
    klr = asm.term((1.0, kl), 1.0) * (1.0 / 7)
    out = asm.run({"nll": nll, "mul": mul, "loss": klr * 0.25 + nll * (1.0 / 7) + mul * (1.0 / 7) + off,
                   "neg": mul * -1.0})
    want = {"nll": -rec.double().sum(), "mul": -0.5 * tree.double().sum() + 2.0 * bow.double().sum()}
    want["loss"] = 0.25 * kl.double() / 7 + want["nll"] / 7 + want["mul"] / 7
    want["neg"] = -want["mul"]
    assert set(out) == set(want)
    for k in want:
        assert abs(float(out[k]) - float(want[k])) < 1e-12, k
    # a row keeps only the components it reads: the backward returns None for the others
    assert set(nll) == {0} and set(mul) == {1, 2}


def test_per_step_factor_stays_symbolic_under_capture(monkeypatch):
    """while a step is being captured (ops.DYN set) a multiplication by the step's KL weight (an ops.Dyn) moves the
    coefficients into the part the kernel multiplies by the device value; the numbers are the same either way"""
    monkeypatch.setattr(ops, "weighted_sums", _definition)
    g = torch.Generator().manual_seed(1)
    rec, kl = torch.randn(5, generator=g), torch.randn((), generator=g)

    def build(weight):
        asm = _Assembly()
        nll = asm.term((-1.0, rec), 1.0) * (1.0 / 5)
        kl_loss = asm.term((1.0, kl), 1.0) * (1.0 / 5)
        item = kl_loss * weight
        return asm, {"KL Item": item, "Loss": item + nll, "NLL": nll}

    asm, rows = build(0.37)
    plain = asm.run(rows)
    monkeypatch.setattr(ops, "DYN", object())          # "capturing"
    asm, rows = build(ops.Dyn(0.37, 5))
    assert rows["KL Item"].dyn is not None and set(rows["KL Item"]) == {("d", 1)}
    assert set(rows["Loss"]) == {("d", 1), 0} and rows["NLL"].dyn is None
    dyn = asm.run(rows)
    for k in plain:
        assert abs(float(plain[k]) - float(dyn[k])) < 1e-12, k
