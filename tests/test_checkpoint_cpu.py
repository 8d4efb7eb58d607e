"""CPU: checkpoint formats.  The BEiT-3 import is checked against the REAL reference's
`load_model_and_may_interpolate` where /root/reference is available (dev container), otherwise self-consistency."""
import os

import pytest
import torch


def _fake_beit3_ckpt(D=128, H=2, F_=256, L=2, patch=16, grid=4, vocab=300, seed=0):
    from oracle import simvg_cpu as O, weights as W
    cfg = O.make_cfg("tiny", 1, patch * grid, patch_size=patch, vocab_size=vocab)
    sd = W.golden_state_dict(cfg, seed)
    return {k[len("vis_enc."):]: v for k, v in sd.items() if k.startswith("vis_enc.")}


def _enc(img, patch, vocab=300, interp=True):
    from simvg_amd.models import build_vis_enc
    return build_vis_enc(dict(type="BEIT3", img_size=img, patch_size=patch, vocab_size=vocab, pretrain=None,
                              vision_embed_proj_interpolate=interp, drop_path_rate=0.0,
                              encoder_cfg=dict(embed_dim=128, heads=2, ffn_dim=256, layers=2)))


def test_beit3_pretrain_import_interpolates(tmp_path):
    from simvg_amd.checkpoint import load_beit3_pretrain
    src = _fake_beit3_ckpt(patch=16, grid=4)          # "beit3_base_patch16": 4x4 positions, 16x16 kernel
    path = os.path.join(tmp_path, "beit3.pth")
    torch.save({"model": src}, path)
    enc = _enc(img=160, patch=32)                      # target: 5x5 positions, 32x32 kernel
    missing, unexpected = load_beit3_pretrain(enc, path)
    assert not missing and not unexpected
    got = enc.state_dict()
    pos = src["beit3.encoder.embed_positions.A.weight"]
    assert torch.equal(got["beit3.encoder.embed_positions.A.weight"][:3], pos[:3])      # extra rows kept
    tok = pos[3:].reshape(1, 4, 4, -1).permute(0, 3, 1, 2)
    ref = torch.nn.functional.interpolate(tok, size=(5, 5), mode="bicubic", align_corners=False).permute(0, 2, 3, 1).reshape(25, -1)
    assert torch.allclose(got["beit3.encoder.embed_positions.A.weight"][3:], ref, atol=1e-6)
    refw = torch.nn.functional.interpolate(src["beit3.vision_embed.proj.weight"], size=(32, 32), mode="bicubic", align_corners=False)
    assert torch.allclose(got["beit3.vision_embed.proj.weight"], refw, atol=1e-6)
    assert torch.equal(got["beit3.encoder.layers.1.ffn.B.fc2.weight"], src["beit3.encoder.layers.1.ffn.B.fc2.weight"])


def test_beit3_pretrain_import_matches_reference(tmp_path):
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("reference tree not present (GPU box)")
    import sys
    from oracle.leaf import EncoderConfig
    from simvg_amd.checkpoint import load_beit3_pretrain
    src = _fake_beit3_ckpt(patch=16, grid=4)
    path = os.path.join(tmp_path, "beit3.pth")
    torch.save({"model": src}, path)
    ref_loader.load()
    mod = sys.modules["simvg.models.vis_encs.beit.beit3"]
    orig = mod._get_base_config
    mod._get_base_config = lambda img_size, patch_size, drop_path_rate, vocab_size: EncoderConfig(
        img_size=img_size, patch_size=patch_size, vocab_size=vocab_size, multiway=True, layernorm_embedding=False,
        normalize_output=True, no_output_layer=True, drop_path_rate=0.0, encoder_embed_dim=128,
        encoder_attention_heads=2, encoder_ffn_embed_dim=256, encoder_layers=2)
    try:
        ref = mod.BEIT3(img_size=160, patch_size=32, vit_type="base", vocab_size=300, vision_embed_proj_interpolate=True,
                        pretrain=path)
    finally:
        mod._get_base_config = orig
    enc = _enc(img=160, patch=32)
    load_beit3_pretrain(enc, path)
    rsd, gsd = ref.state_dict(), enc.state_dict()
    assert set(rsd) == set(gsd)
    for k in rsd:
        assert torch.allclose(rsd[k], gsd[k], atol=1e-6), k


def test_simvg_checkpoint_roundtrip(tmp_path):
    from simvg_amd.checkpoint import load_checkpoint, save_checkpoint
    enc = _enc(img=128, patch=32)
    opt = torch.optim.Adam(enc.parameters(), lr=1e-3)
    p = save_checkpoint(str(tmp_path), 3, enc, opt, d_acc=1.0, best_d_acc=2.0, is_best_det=True)
    assert os.path.exists(os.path.join(tmp_path, "det_best.pth"))
    enc2 = _enc(img=128, patch=32)
    ep, best, _, ok = load_checkpoint(enc2, p, resume=True, optimizer=torch.optim.Adam(enc2.parameters(), lr=1e-3))
    assert ok and ep == 3 and best == 2.0
    for (k, a), (_, b) in zip(enc.state_dict().items(), enc2.state_dict().items()):
        assert torch.equal(a, b), k
    ck = torch.load(p)
    ck["state_dict"] = {"module." + k: v for k, v in ck["state_dict"].items()}     # DDP-saved file
    torch.save(ck, p)
    assert load_checkpoint(enc2, p)[3]
