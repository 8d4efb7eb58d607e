"""CPU (dev container only): the binding INTEGRATION.md section 1 documents, applied to the REFERENCE's own registry.

The reference has no FFI for this path; its plugin boundary is the mmcv-style registry of `simvg/models/builder.py:4-36`.
INTEGRATION.md shows the few lines a maintainer adds to `simvg/models/__init__.py` to route `MIXDETRMB`, `BEIT3` and
`TextGuidedQuerySelectKDDETRHead` through `simvg_amd`.  This test takes THAT code block out of the document, executes it
against the registries of the reference's `builder.py` (executed verbatim from /root/reference through `oracle/ref_loader`),
and builds the model with the reference's `build_model`: the three classes that come out are `simvg_amd`'s, constructed from
the reference's own config dict, with the reference model's state-dict keys and shapes."""
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hook_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    hook = [b for b in blocks if "register_module" in b and "simvg_amd.models" in b]
    assert len(hook) == 1, "INTEGRATION.md section 1 must hold exactly one registry hook block"
    return hook[0]


def test_documented_registry_hook_builds_the_hip_classes_through_the_reference_builder():
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("reference tree not present (GPU box)")
    ref_loader.load()
    ref_models = sys.modules["simvg.models"]                 # carries the registries + build_* of builder.py:4-36
    ref_builder = sys.modules["simvg.models.builder"]
    ns = {k: getattr(ref_models, k) for k in ("MODELS", "VIS_ENCODERS", "HEADS", "LAN_ENCODERS", "FUSIONS")}
    for k, reg in ns.items():
        assert reg is getattr(ref_builder, k), k             # the very registry objects the reference's build_model reads
    # the reference's own classes are registered under these names before the hook
    before = {n: ns[r].get(n) for r, n in (("MODELS", "MIXDETRMB"), ("VIS_ENCODERS", "BEIT3"), ("HEADS", "TextGuidedQuerySelectKDDETRHead"))}
    assert all(c is not None and c.__module__.startswith("simvg.") for c in before.values()), before
    try:
        exec(compile(_hook_source(), "INTEGRATION.md#1", "exec"), ns)
        import simvg_amd.models as amd
        for r, n in (("MODELS", "MIXDETRMB"), ("VIS_ENCODERS", "BEIT3"), ("HEADS", "TextGuidedQuerySelectKDDETRHead")):
            assert ns[r].get(n) is getattr(amd, n), (r, n)
        # the reference's build_model on the reference's config dict (refcoco_onestage shape) now constructs simvg_amd's classes
        cfg = ref_loader.model_cfg("base", 1, 640)
        model = ref_builder.build_model(cfg)
        assert type(model) is amd.MIXDETRMB and type(model.vis_enc) is amd.BEIT3
        assert type(model.head) is amd.TextGuidedQuerySelectKDDETRHead
        # ... with the reference model's state dict: same keys, same shapes (612 entries for ViT-B)
        from oracle import simvg_cpu as O, weights as W
        ref_sd = W.reference_init_state_dict(O.make_cfg("base", 1, 640), 0)
        sd = model.state_dict()
        assert len(sd) == 612 and set(sd) == set(ref_sd), set(sd) ^ set(ref_sd)
        assert all(tuple(sd[k].shape) == tuple(ref_sd[k].shape) for k in sd)
        model.load_state_dict(ref_sd, strict=True)
        # the mmdet-style entry points the reference's loops call
        for m in ("forward_train", "forward_test", "forward"):
            assert callable(getattr(model, m))
        # no CPU fallback behind the boundary: CPU tensors are refused, loudly
        from simvg_amd._lib import SimvgHipError
        with pytest.raises(SimvgHipError):
            model.eval()(torch.zeros(1, 3, 640, 640), torch.zeros(1, 20, dtype=torch.int64),
                         [dict(img_shape=(640, 640, 3), pad_shape=(640, 640, 3), ori_shape=(640, 640, 3), scale_factor=[1.0] * 4)],
                         return_loss=False, text_attention_mask=torch.zeros(1, 20, dtype=torch.int64), with_bbox=True, with_mask=False,
                         rescale=False)
    finally:                                                 # leave the reference registry as the other oracle users expect it
        for (r, n), c in zip((("MODELS", "MIXDETRMB"), ("VIS_ENCODERS", "BEIT3"), ("HEADS", "TextGuidedQuerySelectKDDETRHead")),
                             before.values()):
            ns[r].register_module(name=n, force=True, module=c)
