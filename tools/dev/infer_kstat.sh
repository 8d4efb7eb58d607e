#!/bin/bash
# kernel time of forward_test at batch size $1 (default 1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=${1:-1}
python tools/dev/infer_loop.py $B 100
rm -rf /tmp/ik && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ik -o p -- python tools/dev/infer_loop.py $B 100 > /tmp/ik.log 2>&1
mkdir -p gpurun_out/infer && cp $(find /tmp/ik -name "*kernel_stats.csv" | head -1) gpurun_out/infer/b${B}_kernel_stats.csv
python - <<'PY'
import csv, glob, re
f = glob.glob('/tmp/ik/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f"sum of kernel time per call: {tot / 105e6:.3f} ms; launches per call: {sum(int(r['Calls']) for r in rows) / 105:.0f}")
for r in rows[:28]:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
    print(f"{n[:100]:100s} {int(r['Calls'])/105:6.1f}/call {float(r['AverageNs'])/1e3:8.1f} us {int(r['TotalDurationNs'])/105e6:7.3f} ms/call")
PY
