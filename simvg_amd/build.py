"""Build libsimvg_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsimvg_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=fast",
         "-Wno-unused-result"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, lowp=None):
    """lowp="bf16" (or env SIMVG_LOWP=bf16): the bf16 variant of the same kernels, for A/B measurements -- objects and library
    get a `_bf16` suffix (load it with SIMVG_HIP_LIB=simvg_amd/lib/libsimvg_hip_bf16.so); the default build is fp16."""
    lowp = lowp or os.environ.get("SIMVG_LOWP", "fp16")
    if lowp not in ("fp16", "bf16"):
        raise ValueError("SIMVG_LOWP must be fp16 or bf16")
    suffix = "_bf16" if lowp == "bf16" else ""
    flags = FLAGS + (["-DSIMVG_LOWP_BF16"] if lowp == "bf16" else [])
    # development variants (instrumented kernels, tools/dev): extra compiler flags into a library of their own
    if os.environ.get("SIMVG_EXTRA_FLAGS"):
        flags = flags + os.environ["SIMVG_EXTRA_FLAGS"].split()
        suffix += os.environ.get("SIMVG_LIB_SUFFIX", "_dev")
    lib = os.path.join(LIBDIR, f"libsimvg_hip{suffix}.so")
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h")))
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s)[:-4] + suffix + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([_hipcc(), *flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[simvg_amd.build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if r.stderr.strip() and verbose:
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(lib):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
