"""per-kernel statistics of a rocprofv3 --kernel-trace directory, grouped by (kernel, grid size): calls, mean / min us"""
import csv, glob, re, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        rows[(n, r.get("Grid_Size_X", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for (n, gs), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if pat in n:
        v2 = sorted(v)
        print(f"{len(v):5d} x mean {sum(v)/len(v):8.2f} min {v2[0]:8.2f} p50 {v2[len(v2)//2]:8.2f} us  grid {gs:>8s}  {n[:80]}")
