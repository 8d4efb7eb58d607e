_base_ = ["../_base_/data.py", "../_base_/misc.py"]
max_token = 20
img_size = 640
train_pipeline = [dict(type="Load", max_token=max_token), dict(type="Resize", img_scale=(img_size, img_size), keep_ratio=False)]
data = dict(samples_per_gpu=16, train=dict(pipeline=train_pipeline), val=dict(_delete_=True, type="Other", which_set="val"))
ema = False
lr = 0.0005
optimizer_config = dict(type="Adam", lr=lr, lr_vis_enc=lr / 10.0, lr_lan_enc=lr, betas=(0.9, 0.98), eps=1e-9, amsgrad=True)
model = dict(type="MIXDETRMB", head=dict(num_queries=1, branch_loss_weight={"decoder": 1.0}))
