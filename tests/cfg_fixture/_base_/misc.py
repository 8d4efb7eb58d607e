ema = True
ema_factor = 0.999
use_fp16 = False
seed = 6666
evaluate_interval = 1
log_interval = 50
evaluate_interval = 1
here = "{{ fileBasenameNoExtension }}"
