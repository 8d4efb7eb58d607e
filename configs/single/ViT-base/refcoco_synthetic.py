# ViT-B/32 SimVG (decoder + token branch with dynamic weight-balance distillation) -- the BASELINE.json workload.
# Model / optimizer / scheduler keys and values are those of the reference's single-dataset ViT-base RefCOCO recipe
# (bs 64 per GPU, Adam amsgrad lr 5e-4 with lr/10 on the encoder, clip 0.15, warm-up 3 epochs, x0.1 at epoch 25).
_base_ = ["../../_base_/synthetic_refcoco.py", "../../_base_/misc.py"]

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
        pretrain=None,          # e.g. "pretrain_weights/beit3_base_patch16_224.zip" (16x16 -> 32x32 kernels interpolated)
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
        branch_loss_weight={"decoder": 1.0, "balanced_distill": {"token": 2.0, "distill": 1.0}},
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
scheduler_config = dict(type="MultiStepLRWarmUp", warmup_epochs=3, decay_steps=[25], decay_ratio=0.1, max_epoch=30)
log_interval = 50
