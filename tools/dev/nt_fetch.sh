#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for gn in 0 6 4; do
  echo "== SIMVG_NT_GN=$gn"
  SIMVG_NT_GN=$gn timeout 100 python tools/dev/gemm_bench.py 20 nt | grep -E "qkv|fc1"
  rm -rf /tmp/ntf
  SIMVG_NT_GN=$gn timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/ntf -o p -- python tools/dev/gemm_bench.py 1 nt > /tmp/ntf.log 2>&1
  python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/ntf/*counter_collection.csv")
per = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    if "256sq_w16" not in r["Kernel_Name"]: continue
    per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
v = list(per.values())
# 31 launches per shape (qkv first, then fc1)
print("fetch x2 MB per launch: qkv %.1f  fc1 %.1f" % (v[30] * 2 * 1024 / 1e6, v[-1] * 2 * 1024 / 1e6))
PY
done
