"""`simvg/utils/logger.py:5-18` of the reference (mmcv `get_logger`): one named logger ('SimVG'), stream handler on
every rank, file handler on rank 0, non-zero ranks silenced to ERROR."""
import logging

from .distributed import get_dist_info

_initialized = {}


def get_root_logger(log_file=None, log_level=logging.INFO):
    name = "SimVG"
    logger = logging.getLogger(name)
    rank, _ = get_dist_info()
    if name in _initialized:
        if log_file is not None and rank == 0 and log_file not in _initialized[name]:
            fh = logging.FileHandler(log_file, "w")
            fh.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
            fh.setLevel(log_level)
            logger.addHandler(fh)
            _initialized[name].add(log_file)
        return logger
    logger.propagate = False
    handlers = [logging.StreamHandler()]
    files = set()
    if rank == 0 and log_file is not None:
        handlers.append(logging.FileHandler(log_file, "w"))
        files.add(log_file)
    fmt = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    for h in handlers:
        h.setFormatter(fmt)
        h.setLevel(log_level)
        logger.addHandler(h)
    logger.setLevel(log_level if rank == 0 else logging.ERROR)
    _initialized[name] = files
    return logger
