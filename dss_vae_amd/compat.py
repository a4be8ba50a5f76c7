"""Reference import paths for this package: after ``dss_vae_amd.compat.install()`` the statements a caller of
the reference already has -- ``from dss_vae.models import MODEL_CLS`` (dss_vae/models/__init__.py:12-34),
``from dss_vae.vae.syntax_vae import SyntaxVAE``, ``from dss_vae.encoder import RNNEncoder``,
``from dss_vae.structs.vocab import Vocab`` ... -- resolve to the MI355X implementations, unchanged.

Nothing of the reference is copied or imported: ``dss_vae`` and its sub-modules are registered in
``sys.modules`` as aliases of the corresponding ``dss_vae_amd`` modules (the same module objects, so class
identity, registries and pickles agree between the two names)."""
import importlib
import pkgutil
import sys

ALIAS = "dss_vae"
# reference module names whose counterpart here is named differently
_RENAMED = {
    "metrics.syntax_vae_evaluator": "metrics.evaluator",   # SyntaxVaeEvaluator / SyntaxEvaluator
    "metrics.bleu_scorer": "metrics.bleu",
}


def install(force=False):
    """Register the aliases; returns the list of alias names.  Refuses to shadow a real ``dss_vae`` package that
    is already imported unless force=True."""
    import dss_vae_amd
    cur = sys.modules.get(ALIAS)
    if cur is not None and cur is not dss_vae_amd and not force:
        raise RuntimeError("a different `dss_vae` package is already imported (%r); call install() first, or "
                           "install(force=True) to replace it" % (getattr(cur, "__file__", cur),))
    names = [ALIAS]
    sys.modules[ALIAS] = dss_vae_amd
    prefix = dss_vae_amd.__name__ + "."
    for info in pkgutil.walk_packages(dss_vae_amd.__path__, prefix):
        tail = info.name[len(prefix):]
        if tail.split(".")[0] in ("csrc", "lib"):  # native sources / the ctypes binding: not reference modules
            continue
        sys.modules[ALIAS + "." + tail] = importlib.import_module(info.name)
        names.append(ALIAS + "." + tail)
    for ref_tail, own_tail in _RENAMED.items():
        sys.modules[ALIAS + "." + ref_tail] = importlib.import_module(prefix + own_tail)
        names.append(ALIAS + "." + ref_tail)
    return names


def uninstall():
    for name in [n for n in sys.modules if n == ALIAS or n.startswith(ALIAS + ".")]:
        mod = sys.modules[name]
        if getattr(mod, "__name__", "").startswith("dss_vae_amd"):
            del sys.modules[name]
