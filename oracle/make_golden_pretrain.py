"""Fixture for SURVEY.md 8 row f-2 on the GPU: `BEIT3(pretrain=...)` through the REFERENCE's own loader.

TEST INFRASTRUCTURE.  Run in the dev container only (needs /root/reference):
    python -m oracle.make_golden_pretrain

A synthetic `beit3_base_patch16_224`-shaped checkpoint (ViT-B, 14x14 + 3 position rows, 16x16 patch kernel, the 64 010-row
text table; seeded, regenerated on the GPU box by `synthetic_pretrain_checkpoint`) is loaded by the reference's
`BEIT3.load_model_and_may_interpolate` (`/root/reference/simvg/models/vis_encs/beit/beit3.py:92-174`, executed verbatim through
oracle/ref_loader.py) into the SimVG geometry (640 px, patch 32, `vision_embed_proj_interpolate=True`: bicubic 14x14 -> 20x20
positions and 16x16 -> 32x32 kernel).  Stored: a digest of the state dict the reference ends up with (every key: shape, sum,
|sum|, 16 evenly spaced entries; the two interpolated tensors 4096 entries) and the reference's forward on it for a seeded
two-pair batch; the restatement `oracle.simvg_cpu.beit3_forward` is asserted equal on the same state dict.  The GPU test
(tests/test_pretrain_gpu.py) imports the same file with `simvg_amd.checkpoint.load_beit3_pretrain` into the arena-backed HIP
encoder and must reproduce both."""
import os
import sys
import tempfile

import torch

from . import ref_loader, simvg_cpu as O, weights as W
from .make_golden import _even_idx

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SRC = dict(vit="base", img_size=224, patch_size=16, seed=51)         # the "pretrained" file's geometry
DST = dict(img_size=640, patch_size=32, iseed=61, B=2)               # SimVG's
WIDE = ("beit3.encoder.embed_positions.A.weight", "beit3.vision_embed.proj.weight")


def synthetic_pretrain_checkpoint(path):
    """the seeded stand-in for beit3_base_patch16_224.pth: {"model": {beit3.* keys}} -- same function on both boxes"""
    cfg = O.make_cfg(SRC["vit"], 1, SRC["img_size"], patch_size=SRC["patch_size"])
    sd = W.golden_state_dict(cfg, SRC["seed"])
    src = {k[len("vis_enc."):]: v for k, v in sd.items() if k.startswith("vis_enc.")}
    torch.save({"model": src}, path)
    return src


def digest(sd):
    out = {}
    for k, v in sd.items():
        t = v.detach().float().reshape(-1)
        n = 4096 if k in WIDE else 16
        ix = _even_idx(t.numel(), n) if t.numel() >= n else torch.arange(t.numel())
        out[k] = dict(shape=tuple(v.shape), sum=float(t.double().sum()), abssum=float(t.double().abs().sum()),
                      idx=ix.int(), vals=t[ix].clone())
    return out


def main():
    torch.manual_seed(0)
    ref_loader.load()
    mod = sys.modules["simvg.models.vis_encs.beit.beit3"]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "beit3_base_patch16_224.pth")
        src = synthetic_pretrain_checkpoint(path)
        ref = mod.BEIT3(img_size=DST["img_size"], patch_size=DST["patch_size"], vit_type=SRC["vit"], vocab_size=64010,
                        vision_embed_proj_interpolate=True, pretrain=path)
    ref.eval()
    rsd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    # what the import must have done (sanity on the fixture itself)
    assert tuple(rsd[WIDE[0]].shape) == (403, 768) and tuple(rsd[WIDE[1]].shape) == (768, 3, 32, 32)
    assert torch.equal(rsd[WIDE[0]][:3], src[WIDE[0]][:3])
    assert torch.equal(rsd["beit3.encoder.layers.7.ffn.B.fc2.weight"], src["beit3.encoder.layers.7.ffn.B.fc2.weight"])
    cfg = O.make_cfg(SRC["vit"], 1, DST["img_size"], patch_size=DST["patch_size"])
    batch = W.synthetic_batch(cfg, DST["B"], DST["iseed"])
    with torch.no_grad():
        img_feat, text_feat, cls_feat = ref(batch["img"], batch["ref_expr_inds"], batch["text_attention_mask"])
        sd = {"vis_enc." + k: v for k, v in rsd.items()}
        i2, t2, c2 = O.beit3_forward(sd, cfg, batch["img"], batch["ref_expr_inds"], batch["text_attention_mask"])
    for a, b, what in [(i2, img_feat, "img_feat"), (t2, text_feat, "text_feat"), (c2, cls_feat, "cls_feat")]:
        err = float((a - b).abs().max())
        assert err <= 2e-5 * max(1.0, float(b.abs().max())), (what, err)
        print(f"[pretrain] restatement == reference on {what}: {err:.2e}")
    ix = _even_idx(img_feat.numel(), 8192)
    fx = dict(src=SRC, dst=DST, digest=digest(rsd), cls_feat=cls_feat.clone(), text_feat=text_feat.clone(),
              img_feat=dict(idx=ix, vals=img_feat.reshape(-1)[ix].clone(), max=float(img_feat.abs().max())),
              torch_version=torch.__version__)
    path = os.path.join(GOLDEN_DIR, "pretrain_base_p16_to_p32.pt")
    torch.save(fx, path)
    print(f"[pretrain] wrote {path} ({os.path.getsize(path) / 1024:.1f} kB)")


if __name__ == "__main__":
    main()
