# run-wide defaults; same keys as the reference's configs/_base_/misc.py
ema = True
ema_factor = 0.999
use_fp16 = False
seed = 6666
evaluate_interval = 1
deterministic = True
log_level = "INFO"
log_interval = 50
save_interval = -1
resume_from = None
load_from = None
finetune_from = None
start_evaluate_epoch = 0
