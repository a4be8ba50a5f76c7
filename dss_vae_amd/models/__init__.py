"""Model registry (reference: dss_vae/models/__init__.py:12-34).  Only the DSS-VAE family is on the
MI355X hot path; the other research models of the reference repo are out of scope."""
from ..vae.disentangle_vae import DisentangleVAE
from ..vae.enhance_syntax_vae import EnhanceSyntaxVAE
from ..vae.syntax_vae import SyntaxVAE
from ..vae.vanilla_vae import VanillaVAE

MODEL_CLS = {
    "VanillaVAE": VanillaVAE,
    "DisentangleVAE": DisentangleVAE,
    "SyntaxVAE": SyntaxVAE,
    "SyntaxVAE2": EnhanceSyntaxVAE,
}


def init_create_model(model: str, **kwargs):
    if model not in MODEL_CLS:
        raise ValueError("Invalid model class '{}' provided. Only {} are supported now.".format(
            model, list(MODEL_CLS.keys())))
    return MODEL_CLS[model](**kwargs)


def load_static_model(model: str, model_path: str):
    """models/__init__.py:36-43"""
    if model not in MODEL_CLS:
        raise ValueError("Invalid model class '{}' provided. Only {} are supported now.".format(
            model, list(MODEL_CLS.keys())))
    return MODEL_CLS[model].load(model_path)
