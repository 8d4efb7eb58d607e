#!/bin/bash
# every launch of ONE training step with the gradient exchange forced on one RCCL rank, in stream order -> gpurun_out/reduce_trace.txt;
# prints the launches that do not exist in a plain step (copies / fills / framework kernels) with their neighbours
export TMPDIR=/tmp
repo=$(cd "$(dirname "$0")/../.." && pwd)
cd $repo; mkdir -p gpurun_out
rm -rf /tmp/tr_red
SIMVG_FORCE_REDUCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 \
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_red -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-forward-test --no-extras > /tmp/tr_red.log 2>&1
f=$(find /tmp/tr_red -name "*kernel_trace.csv" | head -1)
python - "$f" > gpurun_out/reduce_trace.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n[:70]
names = [short(r["Kernel_Name"]) for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("adam_kernel")]
seg = rows[idx[-3] + 1: idx[-1] + 1]          # two adam launches close a step
t0 = int(seg[0]["Start_Timestamp"])
print(f"# {len(seg)} launches, span {(int(seg[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms")
for k, r in enumerate(seg):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{k:4d} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  s{r.get('Stream_Id', '?'):>3s} q{r.get('Queue_Id', '?'):>3s}  {short(r['Kernel_Name'])}")
PY
head -1 gpurun_out/reduce_trace.txt
grep -c copyBuffer gpurun_out/reduce_trace.txt
