from .builder import (VIS_ENCODERS, LAN_ENCODERS, FUSIONS, HEADS, MODELS, Registry,
                      build_model, build_vis_enc, build_lan_enc, build_fusion, build_head)
from .vis_encs import *  # noqa: F401,F403  (registers BEIT3)
