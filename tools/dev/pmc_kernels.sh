#!/bin/bash
# issue / busy PMC counters of every kernel of `$@` (separate passes per counter group); prints per-kernel averages
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pk_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pk_$i -o p -- "$@" > /tmp/pk_$i.log 2>&1 || tail -3 /tmp/pk_$i.log
done
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("/tmp/pk_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"(\w+_kernel\w*)(<[^>]*>)?", k)
        if not m or "at::" in k: continue
        a = agg[m.group(0)][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in agg.items():
    print(k)
    for c, (v, n) in sorted(d.items()):
        print(f"   {c:28s} {v / n:16.0f}  (n={n})")
PY
