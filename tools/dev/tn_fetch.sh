#!/bin/bash
# fabric reads of the wgrad kernel for two tile walks: tools/dev/tn_fetch.sh
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
for gk in -1 3; do
  echo "== SIMVG_TN_GK=$gk"
  SIMVG_TN_GK=$gk timeout 100 python tools/dev/gemm_bench.py 20 tn | grep -E "^tn"
  rm -rf /tmp/tnf
  SIMVG_TN_GK=$gk timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/tnf -o p -- python tools/dev/gemm_bench.py 1 tn > /tmp/tnf.log 2>&1
  python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/tnf/*counter_collection.csv")
per = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    if "gemm_tn" not in r["Kernel_Name"]: continue
    per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
v = list(per.values())
print("fetch x2 MB per launch (qkv, out, fc1, fc2):", " ".join("%.1f" % (v[i * 31 + 30] * 2 * 1024 / 1e6) for i in range(4)))
PY
done
