"""Checkpoint formats of the reference, for the MI355X model (host-side file munging only).

* `load_beit3_pretrain`  -- `BEIT3.load_model_and_may_interpolate` (reference `vis_encs/beit/beit3.py:92-174` +
  `vis_encs/beit/utils.py:326-372`): key search "model|module", bicubic interpolation of
  `beit3.encoder.embed_positions.A.weight` (the 3 extra rows are kept, 14x14 -> 20x20) and, when
  `vision_embed_proj_interpolate`, of `beit3.vision_embed.proj.weight` (16x16 -> 32x32); non-strict load that
  ignores `relative_position_index`.
* `save_checkpoint` / `load_checkpoint` -- reference `simvg/utils/checkpoint.py:82-148`: dict with `state_dict`
  (+ optimizer / scheduler / bookkeeping), `module.` prefix handling, strict load with non-strict fallback.
The state_dict schema of the MI355X model equals the reference's, so files are interchangeable.
"""
import os
import shutil

import torch
import torch.nn.functional as F


def _interpolate_beit3(enc, ckpt):
    key = "beit3.encoder.embed_positions.A.weight"
    if key in ckpt:
        pos = ckpt[key]
        emb = pos.shape[-1]
        num_patches = enc.np
        num_extra = (num_patches + 1) + 2 - num_patches          # num_position_embeddings() + 2 - num_patches = 3
        orig = int((pos.shape[-2] - num_extra) ** 0.5)
        new = int(num_patches ** 0.5)
        if orig != new:
            extra = pos[:num_extra].unsqueeze(0)
            tok = pos[num_extra:].reshape(-1, orig, orig, emb).permute(0, 3, 1, 2).float()
            tok = F.interpolate(tok, size=(new, new), mode="bicubic", align_corners=False)
            tok = tok.permute(0, 2, 3, 1).flatten(1, 2)
            ckpt[key] = torch.cat((extra, tok), dim=1).squeeze(0)
    pk = "beit3.vision_embed.proj.weight"
    if pk in ckpt and enc.vision_embed_proj_interpolate:
        tgt = tuple(enc.beit3.vision_embed.proj.weight.shape)
        if tuple(ckpt[pk].shape) != tgt:
            ckpt[pk] = F.interpolate(ckpt[pk].float(), size=tgt[-2:], mode="bicubic", align_corners=False)
    return ckpt


def load_beit3_pretrain(enc, ckpt_path, model_key="model|module"):
    checkpoint = torch.load(ckpt_path, map_location="cpu")
    ckpt = None
    for k in model_key.split("|"):
        if k in checkpoint:
            ckpt = checkpoint[k]
            break
    if ckpt is None:
        ckpt = checkpoint
    ckpt = _interpolate_beit3(enc, dict(ckpt))
    own = enc.state_dict()
    usable = {k: v for k, v in ckpt.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}
    missing = [k for k in own if k not in usable and "relative_position_index" not in k]
    unexpected = [k for k in ckpt if k not in own]
    enc.load_state_dict(usable, strict=False)
    if missing:
        print(f"Weights of {enc.__class__.__name__} not initialized from pretrained model: {missing[:8]}{'...' if len(missing) > 8 else ''}")
    if unexpected:
        print(f"Weights from pretrained model not used in {enc.__class__.__name__}: {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
    return missing, unexpected


def save_checkpoint(work_dir, epoch, model, optimizer=None, scheduler=None, d_acc=0.0, miou=0.0, best_d_acc=0.0,
                    best_miou=0.0, model_ema=None, is_best_det=False, is_best_segm=False, save_interval=-1):
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in model.state_dict().items()}
    ckpt = dict(epoch=epoch, d_acc=d_acc, miou=miou, best_d_acc=best_d_acc, best_miou=best_miou, amp=None,
                state_dict={k: v.detach().cpu() for k, v in sd.items()},
                optimizer=optimizer.state_dict() if optimizer is not None else None,
                scheduler=scheduler.state_dict() if scheduler is not None else None,
                lr=optimizer.param_groups[0]["lr"] if optimizer is not None else None)
    if model_ema is not None:
        ckpt["ema_state_dict"] = model_ema.shadow if hasattr(model_ema, "shadow") else model_ema
    os.makedirs(work_dir, exist_ok=True)
    latest = os.path.join(work_dir, "latest.pth")
    torch.save(ckpt, latest)
    if is_best_det:
        shutil.copyfile(latest, os.path.join(work_dir, "det_best.pth"))
    if is_best_segm:
        shutil.copyfile(latest, os.path.join(work_dir, "segm_best.pth"))
    if save_interval > 0 and (epoch + 1) % save_interval == 0:
        shutil.copyfile(latest, os.path.join(work_dir, f"epoch_{epoch + 1}.pth"))
    return latest


def load_checkpoint(model, path, optimizer=None, scheduler=None, resume=False, use_ema=False):
    """-> (start_epoch, best_d_acc, best_miou, strict_ok).  `load_from` semantics unless resume=True."""
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt["ema_state_dict"] if (use_ema and "ema_state_dict" in ckpt) else ckpt.get("state_dict", ckpt)
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
    try:
        model.load_state_dict(sd, strict=True)
        ok = True
    except RuntimeError:
        model.load_state_dict(sd, strict=False)
        ok = False
    start_epoch = -1
    if resume:
        if optimizer is not None and ckpt.get("optimizer") is not None:
            optimizer.load_state_dict(ckpt["optimizer"])
        if scheduler is not None and ckpt.get("scheduler") is not None:
            scheduler.load_state_dict(ckpt["scheduler"])
        start_epoch = ckpt.get("epoch", -1)
    return start_epoch, ckpt.get("best_d_acc", 0.0), ckpt.get("best_miou", 0.0), ok
