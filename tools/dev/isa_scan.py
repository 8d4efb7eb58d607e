"""Scan the gfx950 ISA of the csrc kernels for loads that are waited for one (or a few) at a time.

    python tools/dev/isa_scan.py [file.hip ...]      (default: every simvg_amd/csrc/*.hip; needs hipcc, no GPU)

Two reports per kernel:
  * inner loops (label `Inner Loop Header` .. back edge) that hold 1-3 global loads and an `s_waitcnt vmcnt(0)`: a run-time trip count
    compiled as load / wait / use per trip -- every trip a memory round trip (round 5: the attention prologues, the decoder FFN
    staging loops, embed_bwd; profiles/r05_sweeps.md sections 12, 14);
  * the number of SHORT load groups (<= 3 loads) that end in `vmcnt(0)` right behind a store or another wait: loads that hipcc kept
    behind a store they may alias with, or loads inside branches / tile epilogues.
Kernels with many workgroups per CU cover such latencies by occupancy; the ones that matter are those with one workgroup per CU or
fewer workgroups than CUs."""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "simvg_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=fast", "-I" + CSRC,
         "-S", "--cuda-device-only"]


def isa(src):
    extra = []
    first = open(src).readline()
    if "simvg-build-flags:" in first:
        extra = first.split("simvg-build-flags:")[1].split()
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    subprocess.run(["hipcc", *FLAGS, *extra, "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def scan(src):
    lines = isa(src)
    kern, seq = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
            seq[kern] = []
        if kern is None:
            continue
        if re.search(r"\b(global|flat|buffer)_load", l) and " lds" not in l:
            seq[kern].append("L")
        elif re.search(r"\b(global|flat|buffer)_store", l):
            seq[kern].append("S")
        elif "vmcnt(0)" in l:
            seq[kern].append("W")
        m = re.match(r"^(\.LBB\d+_\d+):.*Inner Loop Header", l)
        if m:
            lab = m.group(1)
            for j in range(i + 1, min(i + 400, len(lines))):
                if re.search(r"s_c?branch\w*\s+" + re.escape(lab) + r"\b", lines[j]):
                    body = lines[i:j]
                    nl = sum(1 for b in body if re.search(r"(global|buffer|flat)_load", b) and " lds" not in b)
                    if 1 <= nl <= 3 and any("vmcnt(0)" in b for b in body) and len(body) < 120:
                        print(f"  loop  {kern[:80]:80s} {lab}: {len(body)} lines, {nl} load(s) per trip, waited with vmcnt(0)")
                    break
                if re.match(r"^_Z", lines[j]):
                    break
    for k, v in seq.items():
        s = "".join(v)
        short = re.findall(r"(?:(?<=W)|(?<=S))L{1,3}W", s)
        if len(short) >= 6:
            print(f"  waits {k[:80]:80s} {len(short)} short load groups behind a store / wait ({s.count('W')} vmcnt(0), {s.count('L')} loads)")


if __name__ == "__main__":
    for f in sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        print(os.path.basename(f))
        scan(f)
