"""Kernel sequence of ONE training step from a rocprofv3 kernel trace (csv): every launch in stream order with its duration and
the gap to its predecessor; steps are cut at the optimizer's `adam_kernel` launches.  The head = what lies between the encoder's
last forward kernel and its first backward kernel.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline
    python tools/dev/step_trace.py gpurun_out/trace [--step 3] [--all]"""
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"at::native::.*?(\w+Functor|\w+_kernel_cuda|\w+_kernel_impl|\w+Ops)\b", name)
    if name.startswith("at::native"):
        return "at::native " + (m.group(1) if m else name[12:60])
    return name.split("(")[0][:70]


def main():
    d = sys.argv[1]
    want = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 3
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    # two adam launches per step (encoder arena, head arena): a step ends after the second
    ends = cuts[1::2]
    if len(ends) <= want:
        want = len(ends) - 2
    seg = rows[ends[want - 1] + 1: ends[want] + 1]
    t0 = seg[0][0]
    print(f"step {want}: {len(seg)} launches, {(seg[-1][1] - t0) / 1e6:.3f} ms wall, kernel time {sum(e - s for s, e, _ in seg) / 1e6:.3f} ms")
    names = [short(n) for _, _, n in seg]
    # the head: after the encoder's final LayerNorm (the first `ln_fwd_kernel<float, 3>` of the step: fp32 in, fp32 + 16-bit out) up to
    # the encoder backward's first LayerNorm backward (the first `ln_bwd*` launch after the last criterion launch)
    final_ln = next(i for i, n in enumerate(names) if n.startswith("ln_fwd_kernel<float, 3>"))
    last_crit = max(i for i, n in enumerate(names) if n.startswith("criterion_kernel"))
    first_bwd = next(i for i, n in enumerate(names) if i > last_crit and n.startswith("ln_bwd") and (seg[i][1] - seg[i][0]) > 20000)
    head = range(final_ln + 1, first_bwd)
    last_fwd = final_ln - 1
    ht = sum(seg[i][1] - seg[i][0] for i in head)
    print(f"head region: launches {len(head)}, kernel time {ht / 1e3:.1f} us, wall {(seg[first_bwd][0] - seg[last_fwd + 1][1]) / 1e3:.1f} us, "
          f"at::native launches {sum(names[i].startswith('at::native') or 'rocclr' in names[i] for i in head)}")
    agg = {}
    for i in head:
        a = agg.setdefault(names[i], [0, 0])
        a[0] += 1
        a[1] += seg[i][1] - seg[i][0]
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {c:4d} x {t / c / 1e3:7.2f} us = {t / 1e3:8.1f} us  {n}")
    if "--all" in sys.argv:
        prev = seg[0][0]
        for i, (s, e, n) in enumerate(seg):
            mark = "H" if i in head else " "
            print(f"{mark} {i:4d} +{(s - prev) / 1e3:7.2f} us  {(e - s) / 1e3:8.2f} us  {names[i]}")
            prev = e


if __name__ == "__main__":
    main()
