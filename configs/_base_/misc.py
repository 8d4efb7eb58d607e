# run-wide defaults read by tools/train.py, tools/test.py and simvg_amd/apis (the keys of the reference's misc base)
seed, deterministic = 6666, True
ema, ema_factor = True, 0.999
use_fp16 = False                        # accepted for compatibility: the engine computes in bf16 MFMA with fp32 master weights
log_level, log_interval = "INFO", 50
evaluate_interval, start_evaluate_epoch = 1, 0
save_interval = -1                      # epoch_N.pth every N epochs (<= 0: never); latest / best checkpoints are always written
resume_from = load_from = finetune_from = None
