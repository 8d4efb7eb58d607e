#!/bin/bash
# what the wgrad_x kernel's waves wait for: LDS / wait counters of the fc1-shape wgrad (tools/dev/wgrad_profile.py launches it 20 times)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/pmcwl$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcwl$i -o p -- python tools/dev/wgrad_profile.py ${1:-fc1} > /tmp/pmcwl$i.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmcwl$i/*counter_collection.csv")
if not f:
    print("no counter file for pass $i"); print(open("/tmp/pmcwl$i.log").read()[-600:])
else:
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "wgrad_x" not in k: continue
        a = agg.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    for c, (n, v) in agg.items():
        print(f"pass $i {c}: {v / n:.5g} per launch ({n} launches)")
PY
done
