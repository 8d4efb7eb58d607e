#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --vit large --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-test 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','model_tflops_per_gpu')}, d['roofline']['frac'])"
rm -rf /tmp/kl && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kl -o p -- python bench.py --vit large --batch 32 --steps 6 --warmup 2 --no-cpu-baseline --no-forward-test --no-extras > /tmp/kl.log 2>&1
python - <<'PY'
import csv, glob, re
f = glob.glob('/tmp/kl/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:22]:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
    print(f"{n[:95]:95s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f} %")
PY
