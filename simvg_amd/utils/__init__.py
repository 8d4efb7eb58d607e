"""`simvg.utils` of the reference (`simvg/utils/__init__.py:1-3`)."""
from .distributed import init_dist, is_main, reduce_mean, get_dist_info                      # noqa: F401
from .logger import get_root_logger                                                           # noqa: F401
from .checkpoint import load_checkpoint, save_checkpoint, load_pretrained_checkpoint, \
    is_paral_state, de_parallel, log_loaded_info                                             # noqa: F401
