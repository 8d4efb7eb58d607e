#!/bin/bash
# PMC passes (own runs, --pmc only) over tools/dev/attn_one.py; SQ counters 8 per pass.  Usage: tools/dev/attn_pmc.sh <tag> [env...]
# Writes gpurun_out/pmc_<tag>_*.csv summaries (kernel, counter, mean per dispatch).
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  env "$@" rocprofv3 --pmc $PMC -d /tmp/pmc_$name -o out --output-format csv -- python $R/tools/dev/attn_one.py > /tmp/pmc_$name.log 2>&1
  python - "$name" <<'PY'
import csv, glob, sys, collections
name = sys.argv[1]
files = glob.glob(f"/tmp/pmc_{name}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" not in k: continue
        acc[(k[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{name},{k},{c},{sum(v)/len(v):.0f},{len(v)}")
PY
}
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU SQ_CYCLES"; do
  export PMC
  n=$(echo $PMC | cut -d' ' -f1)
  run ${TAG}_$n "$@" >> $R/gpurun_out/pmc_$TAG.csv
done
cat $R/gpurun_out/pmc_$TAG.csv
