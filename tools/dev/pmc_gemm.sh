#!/bin/bash
# PMC passes for the GEMM micro-benchmark (separate passes: TCC has 4 slots, SQ 8).  Usage: tools/dev/pmc_gemm.sh nt|tn
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-nt}
python tools/dev/gemm_bench.py 5 $W
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc$i -o p -- python tools/dev/gemm_bench.py 1 $W > /tmp/pmc$i.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc$i/*counter_collection.csv")
if not f:
    print("no counter file for pass $i"); print(open("/tmp/pmc$i.log").read()[-600:])
else:
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.OrderedDict()
    for r in rows:
        k = r["Kernel_Name"]
        if "gemm" not in k: continue
        key = (r["Dispatch_Id"], r["Counter_Name"])
        agg[key] = agg.get(key, 0.0) + float(r["Counter_Value"])
    disp = sorted({int(d) for d, _ in agg})
    for d in disp[-4:] if len(disp) > 4 else disp:   # last launches = timed ones of each shape are interleaved; print all gemm dispatches compactly
        pass
    # one line per dispatch (3 per shape: 2 warm-up + 1 timed) -> print every third
    for n, d in enumerate(disp):
        if n % 3 != 2: continue
        print("pass $i dispatch", d, {c: v for (dd, c), v in agg.items() if int(dd) == d})
PY
done
