#!/bin/bash
# per-kernel time of a short bench run: bash tools/dev/kstat.sh [grep pattern]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kstat && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstat -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-test --no-extras > /tmp/kstat.log 2>&1
tail -1 /tmp/kstat.log | cut -c1-200
python - "$1" <<'PY'
import csv, glob, sys, re
f = glob.glob('/tmp/kstat/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
steps = max(1, sum(int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']) // 2)      # two Adam launches per step
print(f"sum of kernel time per step: {tot / steps / 1e6:.3f} ms ({steps} steps incl. set-up and warm-up)")
pat = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else None
for r in rows[:40] if pat is None else rows:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
    if pat is None or re.search(pat, n):
        print(f"{n[:90]:90s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us {int(r['TotalDurationNs'])/steps/1e6:7.3f} ms/step")
PY
