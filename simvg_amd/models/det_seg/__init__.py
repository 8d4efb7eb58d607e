from .base import BaseModel, OneStageModel
from .mix_detr_mb import MIXDETRMB
