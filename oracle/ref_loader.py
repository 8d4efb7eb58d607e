"""Load the REAL reference hot-path files from /root/reference (dev container only).

TEST INFRASTRUCTURE. Executes the 13 reference files of SURVEY.md Appendix D verbatim
(`importlib`, from where they lie under /root/reference -- nothing is copied) on top of
`oracle/leaf.py`, which stands in for the third-party leaf packages the reference imports
but this image lacks (mmcv, torchscale, detrex, detectron2, timm, fairscale ...).

Used by `oracle/make_golden.py` to (a) validate `oracle/simvg_cpu.py` (the CPU restatement
that travels to the GPU box) and (b) emit the golden fixtures under tests/golden/.
/root/reference does not exist on the GPU box: nothing in tests -m gpu / smoke / bench
imports this module.
"""
import importlib.util
import os
import sys
import types

import torch

from . import leaf

REF_ROOT = os.environ.get("SIMVG_REFERENCE_ROOT", "/root/reference")

_REF_FILES = [  # (dotted module name, path relative to REF_ROOT)
    ("simvg.models.builder", "simvg/models/builder.py"),
    ("simvg.models.utils", "simvg/models/utils.py"),
    ("simvg.core.criterion.criterion", "simvg/core/criterion/criterion.py"),
    ("simvg.models.heads.utils", "simvg/models/heads/utils.py"),
    ("simvg.models.heads.tgqs_kd_detr_head.transformer",
     "simvg/models/heads/tgqs_kd_detr_head/transformer.py"),
    ("simvg.models.heads.tgqs_kd_detr_head.tgqs_kd_detr_head",
     "simvg/models/heads/tgqs_kd_detr_head/tgqs_kd_detr_head.py"),
    ("simvg.models.vis_encs.beit.utils", "simvg/models/vis_encs/beit/utils.py"),
    ("simvg.models.vis_encs.beit.beit3_base", "simvg/models/vis_encs/beit/beit3_base.py"),
    ("simvg.models.vis_encs.beit.modeling_utils", "simvg/models/vis_encs/beit/modeling_utils.py"),
    ("simvg.models.vis_encs.beit.beit3", "simvg/models/vis_encs/beit/beit3.py"),
    ("simvg.models.det_seg.base", "simvg/models/det_seg/base.py"),
    ("simvg.models.det_seg.one_stage", "simvg/models/det_seg/one_stage.py"),
    ("simvg.models.det_seg.mix_detr_mb", "simvg/models/det_seg/mix_detr_mb.py"),
]


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "simvg"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        if parent not in sys.modules:
            _mod(parent)
        setattr(sys.modules[parent], child, m)
    return m


def _pkg(name):
    m = _mod(name)
    m.__path__ = []
    return m


_loaded = None


def load():
    """Returns the `simvg.models` module of the real reference (registries + build_model)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    L = leaf
    # ---- third-party stand-ins (names exactly as imported by the 13 files) -----------------
    for p in ["mmcv", "timm", "timm.models", "fairscale", "torchscale", "torchscale.architecture",
              "torchscale.component", "torchscale.component.xmoe", "detrex", "detrex.layers",
              "detrex.modeling", "detrex.modeling.matcher", "detectron2", "mmdet", "pycocotools"]:
        _pkg(p)
    _mod("mmcv.utils", Registry=L.Registry)
    _mod("mmcv.runner", BaseModule=L.BaseModule, auto_fp16=L.auto_fp16, force_fp32=L.force_fp32)
    _mod("timm.models.layers", trunc_normal_=L.trunc_normal_)
    _mod("timm.utils", get_state_dict=lambda m, *a, **k: m.state_dict())
    _mod("torchmetrics", Metric=object)
    _mod("fairscale.nn", checkpoint_wrapper=lambda m, *a, **k: m, wrap=lambda m, *a, **k: m)
    sys.modules["fairscale"].nn = sys.modules["fairscale.nn"]
    _mod("torchscale.architecture.config", EncoderConfig=L.EncoderConfig)
    _mod("torchscale.architecture.utils", init_bert_params=L.init_bert_params)
    _mod("torchscale.component.embedding", PositionalEmbedding=L.PositionalEmbedding,
         TextEmbedding=L.TextEmbedding, VisionEmbedding=L.VisionEmbedding)
    _mod("torchscale.component.multiway_network", MutliwayEmbedding=L.MutliwayEmbedding,
         MultiwayWrapper=L.MultiwayWrapper, MultiwayNetwork=L.MultiwayNetwork,
         set_split_position=L.set_split_position)
    _mod("torchscale.component.droppath", DropPath=L.DropPath)
    _mod("torchscale.component.feedforward_network", FeedForwardNetwork=L.FeedForwardNetwork,
         make_experts=L.make_experts)
    _mod("torchscale.component.multihead_attention", MultiheadAttention=L.TSMultiheadAttention)
    _mod("torchscale.component.relative_position_bias", RelativePositionBias=object)
    _mod("torchscale.component.xmoe.moe_layer", MOELayer=object)
    _mod("torchscale.component.xmoe.routing", Top1Gate=object, Top2Gate=object)
    box = dict(box_cxcywh_to_xyxy=L.box_cxcywh_to_xyxy, box_xyxy_to_cxcywh=L.box_xyxy_to_cxcywh,
               box_iou=L.box_iou, generalized_box_iou=L.generalized_box_iou)
    sys.modules["detrex.layers"].__dict__.update(
        FFN=L.FFN, BaseTransformerLayer=L.BaseTransformerLayer, MultiheadAttention=L.DxMultiheadAttention,
        TransformerLayerSequence=L.TransformerLayerSequence, **box)
    _mod("detrex.layers.box_ops", **box)
    _mod("detrex.layers.position_embedding", PositionEmbeddingSine=L.PositionEmbeddingSine,
         PositionEmbeddingLearned=L.PositionEmbeddingLearned)
    _mod("detrex.modeling.matcher.matcher", HungarianMatcher=L.HungarianMatcher)
    _mod("detrex.utils", get_world_size=L.get_world_size,
         is_dist_avail_and_initialized=L.is_dist_avail_and_initialized)
    _mod("detectron2.structures", Boxes=L.Boxes, ImageList=L.ImageList, Instances=L.Instances)
    _mod("detectron2.modeling", detector_postprocess=L.detector_postprocess)
    _mod("mmdet.core", BitmapMasks=object)
    _mod("pycocotools.mask")
    # ---- empty reference packages (their real __init__ pulls the whole zoo) -----------------
    for p in ["simvg", "simvg.models", "simvg.core", "simvg.core.criterion", "simvg.models.heads",
              "simvg.models.heads.tgqs_kd_detr_head", "simvg.models.vis_encs",
              "simvg.models.vis_encs.beit", "simvg.models.det_seg"]:
        _pkg(p)
    _mod("simvg.models.lan_encs", LSTM=type("LSTM", (), {}))
    _mod("simvg.core.criterion.distill_criterion", DistillCriterion=object)  # dead "soft" path
    # Q3: tgqs_kd_detr_head.py:389 hard-codes .cuda(); identity on a CPU-only host.
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
    # ---- execute the reference files verbatim ------------------------------------------------
    for name, rel in _REF_FILES:
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        parent, _, child = name.rpartition(".")
        setattr(sys.modules[parent], child, m)
        spec.loader.exec_module(m)
        if name == "simvg.models.builder":  # what simvg/models/__init__.py:1-2 re-exports
            for k in ["VIS_ENCODERS", "LAN_ENCODERS", "FUSIONS", "HEADS", "MODELS", "build_model",
                      "build_vis_enc", "build_lan_enc", "build_fusion", "build_head"]:
                setattr(sys.modules["simvg.models"], k, getattr(m, k))
    _loaded = sys.modules["simvg.models"]
    return _loaded


_apis_loaded = None


def load_apis():
    """Executes the reference's scheduler / optimizer registries, metrics and EMA verbatim (dev container only):
    -> dict(scheduler=module, optimizer=module, test=module, utils=module).  Stand-ins: mmcv Registry (leaf),
    mmdet bbox_overlaps and torchvision box_area (leaf restatements), logging / dist helpers (trivial)."""
    global _apis_loaded
    if _apis_loaded is not None:
        return _apis_loaded
    load()
    L = leaf
    import logging
    for p in ["torchvision", "torchvision.ops", "mmdet.core.bbox", "mmdet.core.bbox.iou_calculators"]:
        _pkg(p)
    _mod("torchvision.ops.boxes", box_area=L.tv_box_area)
    _mod("mmdet.core.bbox.iou_calculators.iou2d_calculator", bbox_overlaps=L.mmdet_bbox_overlaps)
    _mod("simvg.datasets", extract_data=lambda inputs: {k: v.data[0] for k, v in inputs.items()})   # DataContainer unwrap
    _mod("simvg.utils", get_root_logger=lambda *a, **k: logging.getLogger("SimVG-ref"), reduce_mean=lambda t: t,
         is_main=lambda: True)
    _pkg("simvg.apis")
    # torch >= 2.7 removed the `verbose=` argument the reference still passes (always False): accept and drop it
    import inspect
    import torch.optim.lr_scheduler as _ls
    for _cls in (_ls.LambdaLR, _ls.CosineAnnealingLR, _ls.CosineAnnealingWarmRestarts):
        if "verbose" not in inspect.signature(_cls.__init__).parameters:
            def _wrap(orig):
                def __init__(self, *a, verbose=False, **k):
                    orig(self, *a, **k)
                return __init__
            _cls.__init__ = _wrap(_cls.__init__)
    out = {}
    for key, name, rel in [("scheduler", "simvg.core.scheduler", "simvg/core/scheduler.py"),
                           ("optimizer", "simvg.core.optimizer", "simvg/core/optimizer.py"),
                           ("test", "simvg.apis.test", "simvg/apis/test.py"),
                           ("utils", "simvg.models.utils", "simvg/models/utils.py")]:
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        out[key] = m
    _apis_loaded = out
    return out


_pipes_loaded = None


def load_pipelines():
    """Executes the reference's image transforms (simvg/datasets/pipelines/transforms.py: Resize, Normalize, Pad,
    LargeScaleJitter) verbatim on top of the restated mmcv / OpenCV leaves of oracle/pipeline_cpu.py."""
    global _pipes_loaded
    if _pipes_loaded is not None:
        return _pipes_loaded
    load()
    from . import pipeline_cpu as P
    mm = sys.modules["mmcv"]
    for k in ("imrescale", "imresize", "imnormalize", "impad", "impad_to_multiple", "rescale_size", "is_list_of"):
        setattr(mm, k, getattr(P, k))
    sys.modules["mmcv.utils"].to_2tuple = P.to_2tuple
    mm.utils = sys.modules["mmcv.utils"]
    ds = sys.modules.get("simvg.datasets")
    if ds is None:
        ds = _pkg("simvg.datasets")
    ds.__path__ = []
    _mod("simvg.datasets.builder", PIPELINES=leaf.Registry("PIPELINES"), DATASETS=leaf.Registry("DATASETS"))
    _pkg("simvg.datasets.pipelines")
    name = "simvg.datasets.pipelines.transforms"
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, "simvg/datasets/pipelines/transforms.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    _pipes_loaded = m
    return m


def model_cfg(vit_type="base", num_queries=1, img_size=640, patch_size=32, branch_loss_weight=None):
    """cfg.model of configs/single/ViT-{base,large}/refcoco/refcoco_onestage.py:68-105
    (grefcoco: num_queries=10), with pretrain=None (no checkpoint in this image).  branch_loss_weight: the head's loss
    branches; default = ViT-B one-stage's; {"decoder": 1.0} = *_twostage_1 / pretrian-mixed / finetune_* (e.g.
    configs/single/ViT-large/refcoco/refcoco_twostage_1.py:98); ViT-L one-stage uses token 1.0 / distill 0.4
    (configs/single/ViT-large/refcoco/refcoco_onestage.py:96)."""
    import copy
    blw = copy.deepcopy(branch_loss_weight) if branch_loss_weight is not None else \
        {"decoder": 1.0, "balanced_distill": {"token": 2.0, "distill": 1.0}}
    return dict(
        type="MIXDETRMB",
        vis_enc=dict(type="BEIT3", img_size=img_size, patch_size=patch_size, vit_type=vit_type,
                     drop_path_rate=0.1, vocab_size=64010, freeze_layer=-1,
                     vision_embed_proj_interpolate=True, pretrain=None),
        lan_enc=None, fusion=None,
        head=dict(type="TextGuidedQuerySelectKDDETRHead", num_queries=num_queries, text_max_token=20,
                  in_channels=768 if vit_type == "base" else 1024, embed_dim=256, decoder_freeze=False,
                  num_classes=1, aux_loss=True, num_encoder_layers=6, num_decoder_layers=3,
                  only_decoder=True, text_embed_aug=False,
                  branch_loss_weight=blw,
                  distill_type="hard_weighted", prepare_target_mode="score_iou_weighted",
                  share_predicthead=False, num_token_mlp_layers=1, mlp_aux_loss=False,
                  text_guided_query_generation=True, num_tgqg_layers=2),
    )
