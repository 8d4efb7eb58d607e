#!/bin/bash
# fabric traffic of the wgrad launches by PMC (separate passes, --kernel-trace only): tools/dev/pmc_wgrad.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pmcw$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcw$i -o p -- python tools/dev/wgrad_bench.py > /tmp/pmcw$i.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmcw$i/*counter_collection.csv")
if not f:
    print("no counter file for pass $i"); print(open("/tmp/pmcw$i.log").read()[-600:])
else:
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "wgrad" not in k and "gemm_tn" not in k: continue
        key = (k[:60], r["Grid_Size"], r["Counter_Name"])
        a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    for (k, g, c), (n, v) in agg.items():
        print(f"pass $i {k} grid {g} {c}: {v / n:.4g} per launch ({n} launches)")
PY
done
