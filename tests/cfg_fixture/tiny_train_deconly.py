# tiny_train.py with the head of the reference's *_twostage_1 / pre-training / fine-tuning configs: decoder branch only
_base_ = ["./tiny_train.py"]
model = dict(head=dict(branch_loss_weight={"_delete_": True, "decoder": 1.0}))
