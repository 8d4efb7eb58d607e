"""`simvg.apis` of the reference (`simvg/apis/__init__.py`): set_random_seed, train_model, evaluate_model + metrics."""
from .train import set_random_seed, train_model                                   # noqa: F401
from .test import evaluate_model, accuracy, grec_evaluate_f1_nacc                 # noqa: F401
