#!/bin/bash
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/calib_$set
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/calib_$set -o p -- python tools/dev/pmc_calib.py > /tmp/calib_$set.log 2>&1
done
tail -1 /tmp/calib_FETCH_SIZE.log
python - <<'PY'
import csv, glob, collections
for s in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/calib_{s}/*counter_collection.csv")
    if not f:
        print("no counters for", s); continue
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        if not any(t in k for t in ("sumsq", "gemm_nt", "gemm_tn")): continue
        per.setdefault((k, r["Dispatch_Id"]), 0.0)
        per[(k, r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, d), v in per.items():
        print(f"{s:10s} {k:40s} dispatch {d:>5s}: {v * 1024 / 1e6:9.1f} MB raw")
PY
