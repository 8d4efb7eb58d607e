dataset = "RefCOCOUNC"
data_root = "./data/"
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375])
train_pipeline = [dict(type="Load", max_token=15), dict(type="Resize", img_scale=(512, 512), keep_ratio=False)]
data = dict(
    samples_per_gpu=64,
    workers_per_gpu=4,
    train=dict(type=dataset, which_set="train", annsfile=data_root + "annotations/x.json", pipeline=train_pipeline),
    val=dict(type=dataset, which_set="val", annsfile=data_root + "annotations/x.json", pipeline=train_pipeline),
)
