#!/bin/bash
# L2 hit rate and fabric traffic per kernel for any command: tools/dev/pmc_l2.sh <command...>
# (each pass is cut off after 300 s: the TCC pass over a whole bench.py run did not finish in 25 minutes -- use a short command)
export TMPDIR=/tmp
rm -rf /tmp/pmc_l2a /tmp/pmc_l2b
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_l2a -o p -- "$@" > /tmp/pmc_l2a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d /tmp/pmc_l2b -o p -- "$@" > /tmp/pmc_l2b.log 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for d in ("/tmp/pmc_l2a", "/tmp/pmc_l2b"):
    f = glob.glob(d + "/*counter_collection.csv")
    if not f:
        print("no counters in", d); continue
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:56]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, d)].add(r["Dispatch_Id"])
rows = []
for k, c in agg.items():
    n = max(len(disp[(k, "/tmp/pmc_l2b")]), 1)
    hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    mb = (c.get("FETCH_SIZE", 0) * 2 + c.get("WRITE_SIZE", 0)) * 1024 / n / 1e6      # gfx950: FETCH_SIZE counts half
    rows.append((mb * n, k, n, hit / max(hit + miss, 1), mb))
for tot, k, n, hr, mb in sorted(rows, reverse=True)[:28]:
    print(f"{k:56s} launches {n:5d}  L2 hit {hr:5.2f}  fabric MB/launch {mb:9.1f}  total GB {tot / 1e3:7.2f}")
PY
