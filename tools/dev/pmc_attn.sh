#!/bin/bash
# PMC counters of the isolated attention kernels (separate passes per counter group)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_attn
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rm -rf /tmp/pa_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pa_$i -o p -- python tools/dev/attn_bench.py > /tmp/pa_$i.log 2>&1 || tail -3 /tmp/pa_$i.log
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("/tmp/pa_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" not in k: continue
        import re
        k = re.search(r"attn_\w+(<[^>]*>)?", k).group(0)
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in agg.items():
    print(k)
    for c, (v, n) in sorted(d.items()):
        print(f"   {c:28s} {v / n:16.0f}  (n={n})")
PY
