# tiny end-to-end config for tests/test_tools_gpu.py: reference schema, synthetic data, 2-layer encoder geometry
_base_ = ["./_base_/misc.py"]
dataset = "RefCOCOUNC"
max_token = 20
img_size = 96
data = dict(
    samples_per_gpu=8,
    workers_per_gpu=0,
    train=dict(type="SyntheticRefDataset", which_set="train", length=48, img_size=img_size, max_token=max_token, seed=1),
    val=dict(type="SyntheticRefDataset", which_set="val", length=16, img_size=img_size, max_token=max_token, seed=2),
    testA=dict(type="SyntheticRefDataset", which_set="testA", length=8, img_size=img_size, max_token=max_token, seed=3),
    testB=dict(type="SyntheticRefDataset", which_set="testB", length=8, img_size=img_size, max_token=max_token, seed=4),
)
model = dict(
    type="MIXDETRMB",
    vis_enc=dict(type="BEIT3", img_size=img_size, patch_size=32, vit_type="base", drop_path_rate=0.1, vocab_size=64010,
                 freeze_layer=-1, vision_embed_proj_interpolate=True, pretrain=None,
                 encoder_cfg=dict(embed_dim=128, heads=2, ffn_dim=256, layers=2)),
    lan_enc=None,
    fusion=None,
    head=dict(type="TextGuidedQuerySelectKDDETRHead", num_queries=1, text_max_token=max_token, in_channels=128, embed_dim=256,
              decoder_freeze=False, num_classes=1, aux_loss=True, num_encoder_layers=6, num_decoder_layers=3, only_decoder=True,
              text_embed_aug=False, branch_loss_weight={"decoder": 1.0, "balanced_distill": {"token": 2.0, "distill": 1.0}},
              distill_type="hard_weighted", prepare_target_mode="score_iou_weighted", share_predicthead=False,
              num_token_mlp_layers=1, mlp_aux_loss=False, text_guided_query_generation=True, num_tgqg_layers=2),
)
grad_norm_clip = 0.15
ema = True
deterministic = True
save_interval = -1
start_evaluate_epoch = 0
resume_from = None
load_from = None
finetune_from = None
lr = 0.0005
optimizer_config = dict(type="Adam", lr=lr, lr_vis_enc=lr / 10.0, lr_lan_enc=lr, betas=(0.9, 0.98), eps=1e-9, weight_decay=0, amsgrad=True)
scheduler_config = dict(type="MultiStepLRWarmUp", warmup_epochs=1, decay_steps=[2], decay_ratio=0.1, max_epoch=2)
log_interval = 2
