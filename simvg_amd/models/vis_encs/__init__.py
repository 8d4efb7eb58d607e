from .beit.beit3 import BEIT3

__all__ = ["BEIT3"]
