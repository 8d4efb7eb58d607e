#!/bin/bash
# kernel trace of a short bench run -> gpurun_out/trace/kernel_trace.csv (timestamps), for per-segment analysis
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/trace
rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --roofline-every 100 > gpurun_out/trace/bench.log 2>&1
python - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/tr/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = open("gpurun_out/trace/kernel_trace_compact.csv", "w")
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"]
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)[:90].replace(",", ";")
    out.write(f"{int(r['Start_Timestamp']) - t0},{int(r['End_Timestamp']) - int(r['Start_Timestamp'])},{n}\n")
out.close()
print(len(rows), "kernels")
PY
gzip -f gpurun_out/trace/kernel_trace_compact.csv; ls -la gpurun_out/trace
