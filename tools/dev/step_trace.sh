#!/bin/bash
# ordered kernel list of ONE steady-state training step: tools/dev/step_trace.sh <tag>   -> gpurun_out/<tag>_trace.txt
tag=$1; shift
export TMPDIR=/tmp
repo=$(cd "$(dirname "$0")/../.." && pwd)
cd $repo; mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > /tmp/tr_$tag.log 2>&1
f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
python - "$f" > gpurun_out/${tag}_trace.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.search(r"(FillFunctor|CUDAFunctor_add|MulFunctor|direct_copy|bernoulli|index|sigmoid|threshold|where|FusedAdam|LpNorm|multiplies)", n)
    if "elementwise" in n or "multi_tensor" in n:
        return "at::" + (m.group(1) if m else n[:50])
    return re.match(r"[A-Za-z0-9_:]+", n).group(0)[:48]
names = [short(r["Kernel_Name"]) for r in rows]
# last step = after the last-but-one adam_kernel
idx = [i for i, n in enumerate(names) if n == "adam_kernel"]
seg = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
out = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append((short(r["Kernel_Name"]), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = max(prev_end, e)
print(f"# {len(seg)} launches, span {(int(seg[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms; columns: kernel, start_us, dur_us, gap_before_us")
for n, s, d, g in out:
    print(f"{n:50s} {s:10.1f} {d:8.1f} {g:7.1f}")
PY
head -3 gpurun_out/${tag}_trace.txt
