"""`BEIT3(pretrain=path)` -- the BEiT-3 checkpoint import with bicubic position / patch-kernel
interpolation -- on the HIP model.  The fixture (`tests/golden/pretrain_base_p16_to_p32.pt`, oracle/make_golden_pretrain.py)
was recorded from the REFERENCE's loader (`/root/reference/simvg/models/vis_encs/beit/beit3.py:92-174`) executed on a seeded
beit3_base_patch16_224-shaped file: a digest of the state dict it produced and the reference's forward on it.

CPU part (not gpu): the host-side import reproduces the reference loader's state dict key for key.
GPU part: the imported weights sit in the encoder's arena (fp32 master views + refreshed 16-bit copies) and the HIP forward
reproduces the reference's features -- exactly in the fp32 parity mode, within the 16-bit engine's stated tolerance otherwise;
a second import into the SAME (already used) model is picked up by the next forward (16-bit weight refresh after a load)."""
import os

import pytest
import torch

DEV = "cuda"


def _fixture(golden):
    return golden("pretrain_base_p16_to_p32")


def _make_file(tmp_path):
    from oracle.make_golden_pretrain import synthetic_pretrain_checkpoint
    path = os.path.join(tmp_path, "beit3_base_patch16_224.pth")
    synthetic_pretrain_checkpoint(path)
    return path


def _build(path, dst):
    from simvg_amd.models import build_vis_enc
    return build_vis_enc(dict(type="BEIT3", img_size=dst["img_size"], patch_size=dst["patch_size"], vit_type="base",
                              vocab_size=64010, vision_embed_proj_interpolate=True, pretrain=path))


def _check_digest(sd, dg):
    assert set(sd) == set(dg), set(sd) ^ set(dg)
    for k, d in dg.items():
        v = sd[k].detach().float().cpu()
        assert tuple(v.shape) == tuple(d["shape"]), k
        got = v.reshape(-1)[d["idx"].long()]
        assert torch.allclose(got, d["vals"], atol=1e-6, rtol=0), (k, float((got - d["vals"]).abs().max()))
        assert abs(float(v.double().sum()) - d["sum"]) <= 1e-6 * max(1.0, d["abssum"]), k
        assert abs(float(v.double().abs().sum()) - d["abssum"]) <= 1e-6 * max(1.0, d["abssum"]), k


def test_pretrain_import_equals_reference_loader_digest(golden, tmp_path):
    fx = _fixture(golden)
    enc = _build(_make_file(tmp_path), fx["dst"])
    _check_digest(enc.state_dict(), fx["digest"])


def _features(enc, batch, B, T):
    out = enc.encode(batch["img"].to(DEV), batch["ref_expr_inds"].to(DEV), batch["text_attention_mask"].to(DEV))
    return [t.detach().float().cpu() for t in enc.split_output(out, B, T)]


@pytest.mark.gpu
def test_pretrain_import_runs_on_the_hip_encoder(golden, tmp_path):
    from oracle import simvg_cpu as O, weights as W
    from simvg_amd import _lib
    fx = _fixture(golden)
    dst = fx["dst"]
    path = _make_file(tmp_path)
    enc = _build(path, dst).to(DEV).eval()
    _check_digest(enc.state_dict(), fx["digest"])                 # the arena views hold what the reference loader produced
    cfg = O.make_cfg("base", 1, dst["img_size"], patch_size=dst["patch_size"])
    batch = W.synthetic_batch(cfg, dst["B"], dst["iseed"])
    B, T = dst["B"], cfg.max_token
    ref_img = fx["img_feat"]
    scale = ref_img["max"]
    with torch.no_grad():
        for prec, tol in (("fp32", 2e-5), ("lowp", 1.5e-2 if _lib.lowp_format() == "bf16" else 4e-3)):
            enc.set_precision(prec)
            img_feat, text_feat, cls_feat = _features(enc, batch, B, T)
            e_img = float((img_feat.reshape(-1)[ref_img["idx"]] - ref_img["vals"]).abs().max()) / scale
            e_txt = float((text_feat - fx["text_feat"]).abs().max()) / float(fx["text_feat"].abs().max())
            e_cls = float((cls_feat - fx["cls_feat"]).abs().max()) / float(fx["cls_feat"].abs().max())
            print(f"[pretrain import, {prec}] features vs the reference's: img {e_img:.2e} text {e_txt:.2e} cls {e_cls:.2e} (of max|ref|)")
            assert max(e_img, e_txt, e_cls) <= tol, (prec, e_img, e_txt, e_cls)
        # a second load into the model that has already run: the 16-bit compute copies must follow the arena
        enc.set_precision("lowp")
        before = _features(enc, batch, B, T)[2]
        from simvg_amd.checkpoint import load_beit3_pretrain
        sd = torch.load(path)["model"]
        sd = {k: (v * 1.25 if k.endswith("fc2.weight") else v) for k, v in sd.items()}
        path2 = os.path.join(tmp_path, "scaled.pth")
        torch.save({"model": sd}, path2)
        load_beit3_pretrain(enc, path2)
        after = _features(enc, batch, B, T)[2]
        assert float((after - before).abs().max()) > 1e-2, "the forward after a second import still ran the old 16-bit weights"
        load_beit3_pretrain(enc, path)
        again = _features(enc, batch, B, T)[2]
        assert torch.equal(again, before)
