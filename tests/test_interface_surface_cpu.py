"""Interface surface of the drop-in: every public method of the reference's VAE classes and sub-modules,
and every function of the utility modules the path's callers use, exists here under the same name.  Runs only where
the reference is present (the build container); on the GPU box it is skipped -- the reference does not travel."""
import importlib
import inspect
import os
import sys
import types

import pytest
import torch.nn as nn

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "dss_vae")), reason="reference not present")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLASSES = [("dss_vae.vae.syntax_vae", "SyntaxVAE", "dss_vae_amd.vae.syntax_vae"),
           ("dss_vae.vae.enhance_syntax_vae", "EnhanceSyntaxVAE", "dss_vae_amd.vae.enhance_syntax_vae"),
           ("dss_vae.vae.vanilla_vae", "VanillaVAE", "dss_vae_amd.vae.vanilla_vae"),
           ("dss_vae.vae.disentangle_vae", "DisentangleVAE", "dss_vae_amd.vae.disentangle_vae"),
           ("dss_vae.encoder.rnn_encoder", "RNNEncoder", "dss_vae_amd.encoder.rnn_encoder"),
           ("dss_vae.decoder.rnn_decoder", "RNNDecoder", "dss_vae_amd.decoder.rnn_decoder"),
           ("dss_vae.networks.bridger", "MLPBridger", "dss_vae_amd.networks.bridger"),
           ("dss_vae.structs.vocab", "Vocab", "dss_vae_amd.structs.vocab"),
           ("dss_vae.structs.vocab", "VocabEntry", "dss_vae_amd.structs.vocab"),
           ("dss_vae.structs.dataset", "Dataset", "dss_vae_amd.structs.dataset"),
           ("dss_vae.structs.dataset", "PlainExample", "dss_vae_amd.structs.dataset")]
# members that are construction details of the reference, not interface: the nn.GRU / nn.LSTM class selector
# and its argument check, and a dict the reference also keeps as an instance attribute
NOT_INTERFACE = {"module_cell", "rnn_cell", "valid_args", "id2word"}
MODULES = [("dss_vae.utils.model_utils", "dss_vae_amd.utils.model_utils"),
           ("dss_vae.utils.config_funcs", "dss_vae_amd.utils.config_funcs")]


@pytest.fixture(scope="module")
def reference_on_path():
    saved = list(sys.path), sys.dont_write_bytecode
    sys.dont_write_bytecode = True  # the reference tree is read-only
    stubbed = []
    for name in ("nltk", "nltk.tree"):  # import-time only dependency of dss_vae.structs
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
            stubbed.append(name)
    sys.modules["nltk.tree"].Tree = object
    sys.modules["nltk"].tree = sys.modules["nltk.tree"]
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    yield
    sys.path[:], sys.dont_write_bytecode = saved
    for name in stubbed:
        sys.modules.pop(name, None)
    for name in [m for m in sys.modules if m == "dss_vae" or m.startswith("dss_vae.")]:
        sys.modules.pop(name, None)


def test_public_methods_and_functions_exist(reference_on_path):
    base = set(dir(nn.Module))
    missing = {}
    for ref_mod, cls, our_mod in CLASSES:
        R = getattr(importlib.import_module(ref_mod), cls)
        P = getattr(importlib.import_module(our_mod), cls)
        want = {n for n, _ in inspect.getmembers(R) if not n.startswith("__") and n not in base} - NOT_INTERFACE
        lack = sorted(want - {n for n, _ in inspect.getmembers(P)})
        if lack:
            missing[cls] = lack
    for ref_mod, our_mod in MODULES:
        R, P = importlib.import_module(ref_mod), importlib.import_module(our_mod)
        want = {n for n, v in inspect.getmembers(R, inspect.isfunction) if v.__module__ == ref_mod}
        lack = sorted(want - {n for n, _ in inspect.getmembers(P)})
        if lack:
            missing[ref_mod] = lack
    assert not missing, missing


def test_registry_keys(reference_on_path):
    from dss_vae_amd.models import MODEL_CLS, init_create_model, load_static_model  # noqa: F401
    ref = importlib.import_module("dss_vae.models").MODEL_CLS
    for key in ("VanillaVAE", "DisentangleVAE", "SyntaxVAE", "SyntaxVAE2"):
        assert key in MODEL_CLS and key in ref
        assert MODEL_CLS[key].__name__ == ref[key].__name__


def test_reference_import_paths_resolve_to_this_package():
    """dss_vae_amd.compat.install(): a reference-side caller's import lines work unchanged (in a child process:
    the aliases must not leak into the tests that import the real reference)"""
    import subprocess
    code = r'''
import sys
sys.path.insert(0, sys.argv[1])
import dss_vae_amd.compat as compat
names = compat.install()
from dss_vae.models import MODEL_CLS, init_create_model
from dss_vae.vae.syntax_vae import SyntaxVAE
from dss_vae.vae.enhance_syntax_vae import EnhanceSyntaxVAE
from dss_vae.vae.vanilla_vae import VanillaVAE
from dss_vae.vae.disentangle_vae import DisentangleVAE
from dss_vae.encoder import RNNEncoder
from dss_vae.decoder import RNNDecoder
from dss_vae.networks.bridger import MLPBridger
from dss_vae.networks.embeddings import Embeddings
from dss_vae.structs.vocab import Vocab, VocabEntry
from dss_vae.structs.dataset import Dataset, PlainExample
from dss_vae.utils.nn_funcs import to_input_variable, unk_replace, id2word
from dss_vae.utils.model_utils import build_model, get_lr_schedule
from dss_vae.metrics.syntax_vae_evaluator import SyntaxVaeEvaluator
import dss_vae_amd.models, dss_vae_amd.vae.syntax_vae, dss_vae_amd.structs.vocab
assert MODEL_CLS is dss_vae_amd.models.MODEL_CLS
assert MODEL_CLS["SyntaxVAE"] is SyntaxVAE is dss_vae_amd.vae.syntax_vae.SyntaxVAE
assert MODEL_CLS["SyntaxVAE2"] is EnhanceSyntaxVAE and MODEL_CLS["VanillaVAE"] is VanillaVAE
assert MODEL_CLS["DisentangleVAE"] is DisentangleVAE
assert Vocab is dss_vae_amd.structs.vocab.Vocab
assert "dss_vae.models" in names and "dss_vae.vae.base_vae" in names
compat.uninstall()
assert "dss_vae" not in sys.modules
'''
    subprocess.run([sys.executable, "-c", code, ROOT], check=True,
                   env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
