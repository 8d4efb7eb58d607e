# ViT-B/32 SimVG mix pre-training recipe on synthetic pairs: the model / optimizer / scheduler keys and values of the
# reference's configs/mix/ViT-base/pretrian-mixed.py (decoder branch only -- branch_loss_weight={"decoder": 1.0}: no token
# branch in training or evaluation, loss dict = {loss_dgt, loss_total}; bs 32 per GPU; x0.1 at epochs 21 and 27), with the
# `Mixed` dataset (RefCOCO/+/g + ReferIt + Flickr30k annotation files, absent from this image) replaced by generated pairs.
_base_ = ["../../_base_/synthetic_refcoco.py", "../../_base_/misc.py"]

data = dict(samples_per_gpu=32)

model = dict(
    type="MIXDETRMB",
    vis_enc=dict(
        type="BEIT3",
        img_size=640,
        patch_size=32,
        vit_type="base",
        drop_path_rate=0.1,
        vocab_size=64010,
        freeze_layer=-1,
        vision_embed_proj_interpolate=True,
        pretrain=None,          # e.g. "pretrain_weights/beit3_base_patch16_224.zip"
    ),
    lan_enc=None,
    fusion=None,
    head=dict(
        type="TextGuidedQuerySelectKDDETRHead",
        num_queries=1,
        text_max_token=20,
        in_channels=768,
        embed_dim=256,
        decoder_freeze=False,
        num_classes=1,
        aux_loss=True,
        num_encoder_layers=6,
        num_decoder_layers=3,
        only_decoder=True,
        text_embed_aug=False,
        branch_loss_weight={"decoder": 1.0},
        distill_type="hard_weighted",
        prepare_target_mode="score_iou_weighted",
        share_predicthead=False,
        num_token_mlp_layers=1,
        mlp_aux_loss=False,
        text_guided_query_generation=True,
        num_tgqg_layers=2,
    ),
)

grad_norm_clip = 0.15
use_fp16 = False
ema = False

lr = 0.0005
optimizer_config = dict(
    type="Adam",
    lr=lr,
    lr_vis_enc=lr / 10.0,
    lr_lan_enc=lr,
    betas=(0.9, 0.98),
    eps=1e-9,
    weight_decay=0,
    amsgrad=True,
)
scheduler_config = dict(type="MultiStepLRWarmUp", warmup_epochs=3, decay_steps=[21, 27], decay_ratio=0.1, max_epoch=30)
log_interval = 50
