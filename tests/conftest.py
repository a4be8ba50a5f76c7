import os
import sys
from argparse import Namespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=True)


def golden_args(g):
    return Namespace(**g["args"])


CASES = ["SyntaxVAE2_tiny", "SyntaxVAE2_mid", "SyntaxVAE_tiny", "SyntaxVAE_mid", "VanillaVAE_tiny",
         "VanillaVAE_mid", "DisentangleVAE_tiny", "DisentangleVAE_mid", "DisentangleVAE_plain"]


@pytest.fixture(params=CASES)
def golden_case(request):
    return request.param, load_golden(request.param)


@pytest.fixture(autouse=True)
def _verify_uniform_upstream_contract(request):
    """every GPU test runs the fused CE schedule with its contract verified (all rows of a linear_ce call receive
    the same upstream gradient): a model change that breaks it fails here instead of training on wrong gradients"""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from dss_vae_amd import ops
    old = ops.CHECK_UNIFORM_UPSTREAM
    ops.CHECK_UNIFORM_UPSTREAM = True
    yield
    ops.CHECK_UNIFORM_UPSTREAM = old
