"""largest idle gaps of a rocprofv3 kernel trace (csv) and the kernels around them"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])[:50]
lo = int(len(rows) * float(sys.argv[2]) if len(sys.argv) > 2 else len(rows) * 0.6)
rows = rows[lo:]
end = int(rows[0]["End_Timestamp"])
gaps = []
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - max(end, int(a["End_Timestamp"]))
    end = max(end, int(a["End_Timestamp"]), int(b["End_Timestamp"]) if False else end)
    end = max(end, int(a["End_Timestamp"]))
    if g > 0:
        gaps.append((g, name(a), name(b), a.get("Stream_Id", "?"), b.get("Stream_Id", "?")))
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("span ms", span / 1e6, "idle ms", sum(g[0] for g in gaps) / 1e6, "kernels", len(rows))
import collections
by = collections.Counter()
for g, a, b, sa, sb in gaps:
    by[(a, b)] += g
for (a, b), g in by.most_common(14):
    print(f"{g / 1e6:8.2f} ms idle between  {a:50s} -> {b}")
