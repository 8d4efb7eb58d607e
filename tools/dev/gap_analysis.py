"""GPU idle-gap analysis from a rocprofv3 kernel trace: python tools/dev/gap_analysis.py kernel_trace.csv
Groups the idle time between consecutive kernels by the name of the kernel that FOLLOWS the gap."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# keep the last 40 % of the trace (steady state)
t0 = ev[0][0] + int((ev[-1][1] - ev[0][0]) * 0.6)
ev = [e for e in ev if e[0] >= t0]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+)", n); return (m.group(1) if m else n)[:40]
gap_by, cnt_by = collections.Counter(), collections.Counter()
busy = 0; idle = 0; big = []
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    busy += e0 - s0
    g = max(0, s1 - e0)
    idle += g
    gap_by[short(n1)] += g; cnt_by[short(n1)] += 1
    if g > 30000: big.append((g / 1e3, short(n0), short(n1)))
span = ev[-1][1] - ev[0][0]
print(f"span {span/1e6:.1f} ms  busy {busy/1e6:.1f} ms  idle {idle/1e6:.1f} ms ({100*idle/span:.1f} %)  kernels {len(ev)}")
for k, g in gap_by.most_common(14):
    print(f"  gap before {k:42s} total {g/1e6:7.3f} ms  n={cnt_by[k]:5d}  avg {g/cnt_by[k]/1e3:6.2f} us")
big.sort(reverse=True)
print("largest gaps (us):", [(round(g), a, b) for g, a, b in big[:12]])
