"""`simvg.core` of the reference: optimizer / scheduler registries (the criteria live inside the head here)."""
from .optimizer import OPTIMIZERS, build_optimizer, FlatAdam   # noqa: F401
from .scheduler import SCHEDULERS, build_scheduler              # noqa: F401
