"""Evaluator glue around the MI355X decode path (reference: dss_vae/metrics/)."""
from .evaluator import SyntaxVaeEvaluator, get_evaluator  # noqa: F401
