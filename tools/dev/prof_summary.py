"""Condense a rocprofv3 kernel_stats.csv: python tools/dev/prof_summary.py stats.csv steps [out.csv] [header...]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
def short(n):
    n0 = n
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]{0,60}>)?)", n)
    s = m.group(1) if m else n[:80]
    if "multi_tensor_apply" in n0:
        f = re.search(r"(LerpFunctor|Sqrt|maximum|LpNormFunctor|std::divides|std::multiplies|std::plus|FusedAdam|fused_adam|Adam)", n0)
        s = "at::multi_tensor_apply<" + (f.group(1) if f else "?") + ">"
    elif "elementwise_kernel" in n0:
        f = re.search(r"(FillFunctor|CUDAFunctor_add|MulFunctor|direct_copy|bernoulli|index_kernel|gpu_index|sigmoid|threshold|where)", n0)
        s = "at::elementwise<" + (f.group(1) if f else "?") + ">"
    return s[:90]
agg = {}
tot = 0.0
for r in rows:
    t = int(r["TotalDurationNs"]) / 1e6
    tot += t
    k = short(r["Name"])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += int(r["Calls"]); a[1] += t
lines = [" ".join(sys.argv[4:]), "kernel,calls,total_ms,avg_us,percent,ms_per_step"]
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if t / steps < 0.03: continue
    lines.append(f"{k},{c},{t:.3f},{t/c*1e3:.2f},{100*t/tot:.2f},{t/steps:.3f}")
lines.append(f"# total kernel time {tot:.1f} ms over {steps} steps = {tot/steps:.2f} ms/step")
txt = "\n".join(lines) + "\n"
if len(sys.argv) > 3 and sys.argv[3] != "-":
    open(sys.argv[3], "w").write(txt)
print(txt)
