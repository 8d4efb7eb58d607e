#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kq && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kq -o p -- python bench.py --queries 10 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-test --no-extras > /tmp/kq.log 2>&1
python - <<'PY'
import csv, glob, re
f = glob.glob('/tmp/kq/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = max(1, sum(int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']) // 2)
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f"kernel time per step {tot/steps/1e6:.2f} ms over {steps} steps")
keys = ("gemm_f32", "attn_small", "ln_fwd_kernel<float, 1>", "ln_bwd_kernel<float, float, 1>", "match", "criterion", "soft_targets", "at::", "rocclr", "postprocess")
for r in rows:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
    if any(k in n for k in keys) and int(r['TotalDurationNs'])/steps > 20e3:
        print(f"{n[:100]:100s} {int(r['Calls'])/steps:6.1f}/step {float(r['AverageNs'])/1e3:8.1f} us {int(r['TotalDurationNs'])/steps/1e6:7.3f} ms/step")
PY
