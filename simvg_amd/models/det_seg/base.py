"""BaseModel / OneStageModel -- train/test dispatch of the reference's model API.

Mirror of `simvg/models/det_seg/base.py:5-27` and `one_stage.py:6-26` without mmcv: `forward(img,
ref_expr_inds, img_metas, return_loss=True, **kw)` writes `img_meta['batch_input_shape']` and dispatches to
`forward_train` / `forward_test`; the constructor builds vis_enc / lan_enc / head / fusion from cfg dicts.
"""
import torch.nn as nn

from .. import builder


class BaseModel(nn.Module):
    def __init__(self):
        super().__init__()
        self.fp16_enabled = False

    def add_batch_input_shape(self, img, img_metas):
        batch_input_shape = tuple(img.size()[-2:])
        for img_meta in img_metas:
            img_meta["batch_input_shape"] = batch_input_shape

    def forward(self, img, ref_expr_inds, img_metas, return_loss=True, **kwargs):
        self.add_batch_input_shape(img, img_metas)
        if return_loss:
            return self.forward_train(img, ref_expr_inds, img_metas, **kwargs)
        return self.forward_test(img, ref_expr_inds, img_metas, **kwargs)


@builder.MODELS.register_module()
class OneStageModel(BaseModel):
    def __init__(self, word_emb, num_token, vis_enc, lan_enc, head, fusion):
        super().__init__()
        self.vis_enc = builder.build_vis_enc(vis_enc)
        if lan_enc is not None:
            self.lan_enc = builder.build_lan_enc(lan_enc, {"word_emb": word_emb, "num_token": num_token})
        if head is not None:
            self.head = builder.build_head(head)
        if fusion is not None:
            self.fusion = builder.build_fusion(fusion)
