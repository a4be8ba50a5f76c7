"""On-disk formats of the reference: ``vocab.bin`` = pickled Vocab, ``train/dev/test.bin``
= pickled *list* of PlainExample (structs/generate_dataset.py:146-157), ``model.bin`` =
torch.save({'args': Namespace, 'vocab': Vocab, 'state_dict'}) (vae/base_vae.py:206-220).

Reading: a restricted unpickler that resolves the reference's class paths
(``dss_vae.structs.vocab.Vocab`` ...) onto this package's classes and refuses every other global, so a
data file cannot run code.  Writing: the same class paths as the reference, so its loaders read the
files this package writes."""
import io
import pickle
from argparse import Namespace

from .dataset import PlainExample
from .vocab import Vocab, VocabEntry

_CLASSES = {
    ("dss_vae.structs.vocab", "Vocab"): Vocab, ("dss_vae.structs.vocab", "VocabEntry"): VocabEntry,
    ("dss_vae.structs.dataset", "PlainExample"): PlainExample,
    ("dss_vae_amd.structs.vocab", "Vocab"): Vocab, ("dss_vae_amd.structs.vocab", "VocabEntry"): VocabEntry,
    ("dss_vae_amd.structs.dataset", "PlainExample"): PlainExample,
    ("argparse", "Namespace"): Namespace,
}
_REFERENCE_PATH = {Vocab: ("dss_vae.structs.vocab", "Vocab"), VocabEntry: ("dss_vae.structs.vocab", "VocabEntry"),
                   PlainExample: ("dss_vae.structs.dataset", "PlainExample")}


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return _CLASSES[(module, name)]
        except KeyError:
            raise pickle.UnpicklingError("refusing global %s.%s in a DSS-VAE data file" % (module, name))


def load_bin(path):
    with open(path, "rb") as f:
        return _Unpickler(f).load()


def loads_bin(data):
    return _Unpickler(io.BytesIO(data)).load()


import contextlib


@contextlib.contextmanager
def reference_class_paths():
    """While active, Vocab / VocabEntry / PlainExample present themselves under the reference's module paths
    (``dss_vae.structs.*``), so that pickle (and torch.save, which pickles non-tensor objects) writes the class
    paths the reference's own loaders resolve."""
    import sys
    import types
    saved_mod, saved_attr = {}, {}
    try:
        for cls, (module, name) in _REFERENCE_PATH.items():
            parts = module.split(".")
            for k in range(1, len(parts) + 1):
                mn = ".".join(parts[:k])
                if mn not in sys.modules:
                    saved_mod[mn] = None
                    sys.modules[mn] = types.ModuleType(mn)
            mod = sys.modules[module]
            saved_attr[(module, name)] = (getattr(mod, name, None), cls.__module__)
            setattr(mod, name, cls)
            cls.__module__ = module
        yield
    finally:
        for (module, name), (old, old_module) in saved_attr.items():
            cls = _CLASSES[(module, name)]
            cls.__module__ = old_module
            if old is None:
                try:
                    delattr(sys.modules[module], name)
                except AttributeError:
                    pass
            else:
                setattr(sys.modules[module], name, old)
        for mn in saved_mod:
            sys.modules.pop(mn, None)


def dump_bin(obj, path, reference_layout=True):
    """reference_layout=True writes exactly what the reference's pickle.dump writes (class paths under
    ``dss_vae.structs``), so the reference's ``pickle.load`` finds its own classes."""
    with (reference_class_paths() if reference_layout else contextlib.nullcontext()):
        with open(path, "wb") as f:
            pickle.dump(obj, f)


def save_model_bin(obj, path):
    """torch.save of {'args', 'vocab', 'state_dict'} with the vocabulary pickled under the reference's class
    paths: the reference's ``torch.load(model.bin)`` finds ``dss_vae.structs.vocab.Vocab``"""
    import torch
    with reference_class_paths():
        torch.save(obj, path)


def load_model_bin(path, map_location="cpu"):
    """{'args', 'vocab', 'state_dict'} written by BaseVAE.save (here or by the reference); tensors through
    torch's weights-only loader, the Namespace / Vocab objects through the allow-list above."""
    import torch
    allow = [Namespace, Vocab, VocabEntry, PlainExample,
             (Vocab, "dss_vae.structs.vocab.Vocab"), (VocabEntry, "dss_vae.structs.vocab.VocabEntry")]
    try:
        with torch.serialization.safe_globals(allow):
            return torch.load(path, map_location=map_location, weights_only=True)
    except (TypeError, ValueError):  # torch without (class, "path") pairs in safe_globals
        with torch.serialization.safe_globals(allow[:4]):
            return torch.load(path, map_location=map_location, weights_only=True)
