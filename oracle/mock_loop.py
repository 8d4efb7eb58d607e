"""A CPU-sized stand-in model + loader with the interface `train_model` / `evaluate_model` expect, shared by the golden
generator (which drives the REFERENCE's loops with it) and the CPU tests (which drive this build's loops with it).
TEST INFRASTRUCTURE ONLY."""
import torch


class DC:
    """minimal mmcv.parallel.DataContainer: `.data` is a list with one entry per GPU"""

    def __init__(self, data, cpu_only=False):
        self.data, self.cpu_only = [data], cpu_only


class MockVG(torch.nn.Module):
    """img [B,3,8,8] + ids -> two box predictions (decoder / token branch) by a linear map; L1 losses."""

    def __init__(self, grec=False, nq=4):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.vis_enc_proj = torch.nn.Linear(192, 16)
        self.head = torch.nn.Linear(16, 4 * (nq if grec else 1))
        self.head_tok = torch.nn.Linear(16, 4 * (nq if grec else 1))
        self.score = torch.nn.Linear(16, nq)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        self.register_buffer("empty_weight", torch.tensor([1.0, 0.1]))
        self.grec, self.nq = grec, nq

    def _boxes(self, raw, base):
        # the loader hides the target box in the first 4 pixels: prediction = that + a learned offset of up to +-9 px
        c = base + 9.0 * torch.tanh(raw)
        return torch.cat([torch.min(c[..., :2], c[..., 2:]), torch.max(c[..., :2], c[..., 2:]) + 1.0], -1)

    def forward(self, img, ref_expr_inds, img_metas, text_attention_mask=None, gt_bbox=None, return_loss=True,
                rescale=False, with_bbox=False, with_mask=False):
        f = torch.tanh(self.vis_enc_proj(img.flatten(1)) + ref_expr_inds.float().mean(1, keepdim=True) * 1e-3)
        B = img.shape[0]
        base = img[:, 0, 0, :4] * 60.0
        if self.grec:
            bq = base[:, None, :]
            b0, b1 = self._boxes(self.head(f).view(B, self.nq, 4), bq), self._boxes(self.head_tok(f).view(B, self.nq, 4), bq)
            sc = self.score(f).sigmoid()
            preds = [dict(pred_bboxes=[dict(scores=sc[i], boxes=b[i]) for i in range(B)], pred_masks=None) for b in (b0, b1)]
        else:
            b0, b1 = self._boxes(self.head(f), base), self._boxes(self.head_tok(f), base)
            preds = [dict(pred_bboxes=b0.detach(), pred_masks=None), dict(pred_bboxes=b1.detach(), pred_masks=None)]
        if not return_loss:
            return preds
        if self.grec:
            tgt = torch.stack([g[0] for g in gt_bbox])
            l0, l1 = (b0[:, 0] - tgt).abs().mean() / 60, (b1[:, 0] - tgt).abs().mean() / 60
        else:
            tgt = torch.stack(list(gt_bbox))
            l0, l1 = (b0 - tgt).abs().mean() / 60, (b1 - tgt).abs().mean() / 60
        losses = dict(loss_dec=l0, loss_token=l1, loss_total=l0 + 2.0 * l1)
        return losses, preds


def batches(n, B, seed, grec=False, wrap=False):
    """n batches; wrap=True -> DataContainer-wrapped like the reference's collate output, else plain tensors."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        img = torch.randn(B, 3, 8, 8, generator=g)
        ids = torch.randint(0, 100, (B, 6), generator=g)
        pad = torch.zeros(B, 6, dtype=torch.int64)
        xy = torch.rand(B, 2, generator=g) * 30
        wh = 5 + torch.rand(B, 2, generator=g) * 25
        gt = torch.cat([xy, xy + wh], 1)
        img[:, 0, 0, :4] = gt / 60.0
        metas = [dict(filename=f"m{i}", target=[dict(category_id=1 if (i % 3) else -1)]) for i in range(B)]
        if grec:
            gts = [gt[i:i + 1] for i in range(B)]
        if wrap:
            d = dict(img=DC(img), ref_expr_inds=DC(ids), text_attention_mask=DC(pad), img_metas=DC(metas, cpu_only=True),
                     gt_bbox=DC(gts if grec else [gt[i] for i in range(B)]))
        else:
            d = dict(img=img, ref_expr_inds=ids, text_attention_mask=pad, img_metas=metas, gt_bbox=gts if grec else gt)
        out.append(d)
    return out


class Loader(list):
    class _S:
        def set_epoch(self, e):
            pass
    sampler = _S()


class Cfg(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def make_cfg(dataset="RefCOCOUNC"):
    return Cfg(distributed=False, use_fp16=False, grad_norm_clip=0.15, ema=True, dataset=dataset, log_interval=2)
