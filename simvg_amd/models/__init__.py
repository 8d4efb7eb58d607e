"""`simvg.models` of the reference, MI355X-native: same registries, same build_* helpers, same class names
(`simvg/models/__init__.py:1-8`)."""
from .builder import (VIS_ENCODERS, LAN_ENCODERS, FUSIONS, HEADS, MODELS, Registry,
                      build_model, build_vis_enc, build_lan_enc, build_fusion, build_head)
from .det_seg import *     # noqa: F401,F403
from .heads import *       # noqa: F401,F403
from .vis_encs import *    # noqa: F401,F403
