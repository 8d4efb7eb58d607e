#!/bin/bash
# LDS / wait counters per kernel for any command: tools/dev/pmc_lds.sh <name-filter> <command...>
filt=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/pmc_lds
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_lds -o p -- "$@" > /tmp/pmc_lds.log 2>&1
python - "$filt" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pmc_lds/*counter_collection.csv")
if not f:
    print(open("/tmp/pmc_lds.log").read()[-800:]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if sys.argv[1] not in k: continue
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    idx = max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1)
    print(f"{k:60s} LDS conflict/active {c.get('SQ_LDS_BANK_CONFLICT', 0) / idx:6.3f}  mfma_busy/wave_cycles {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):6.3f}  "
          f"wait_any {c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):5.2f}  wait_lds {c.get('SQ_WAIT_INST_LDS', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):5.3f}")
PY
