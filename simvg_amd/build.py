"""Build libsimvg_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

What is rebuilt is decided by CONTENT, not by file times (a snapshot copy, a checkout or a stray object file newer than its
source would fool mtimes): every object has a side file `<name>.o.sha` holding the sha256 of its source, of every header and of
the compiler flags; the library embeds the sha256 of all sources (`simvg_source_hash()`, csrc/api.hip), which `_lib.load()`
compares with the sources it finds next to it."""
import glob
import hashlib
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


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        h.update(open(p, "rb").read())
    return h.hexdigest()


def source_hash(flags=()):
    """sha256 over csrc/*.hip, csrc/*.h (names + contents) and the variant's extra flags: what the library embeds"""
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(CSRC, "*.h")))
    return _sha(srcs, " ".join(flags))[:32]


def build(force=False, verbose=True, lowp=None):
    """lowp="bf16" (or env SIMVG_LOWP=bf16): the bf16 variant of the same kernels, for A/B measurements -- objects and library
    get a `_bf16` suffix (load it with SIMVG_HIP_LIB=simvg_amd/lib/libsimvg_hip_bf16.so); the default build is fp16."""
    lowp = lowp or os.environ.get("SIMVG_LOWP", "fp16")
    if lowp not in ("fp16", "bf16"):
        raise ValueError("SIMVG_LOWP must be fp16 or bf16")
    suffix = "_bf16" if lowp == "bf16" else ""
    variant = ["-DSIMVG_LOWP_BF16"] if lowp == "bf16" else []
    # development variants (instrumented kernels, tools/dev): extra compiler flags into a library of their own
    if os.environ.get("SIMVG_EXTRA_FLAGS"):
        variant = variant + os.environ["SIMVG_EXTRA_FLAGS"].split()
        suffix += os.environ.get("SIMVG_LIB_SUFFIX", "_dev")
    whole = source_hash(variant)
    flags = FLAGS + variant
    lib = os.path.join(LIBDIR, f"libsimvg_hip{suffix}.so")
    os.makedirs(LIBDIR, exist_ok=True)
    for stray in glob.glob(os.path.join(LIBDIR, "*.o.*")):          # compiler temporaries of an interrupted run
        if not stray.endswith(".o.sha"):
            os.remove(stray)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h")))
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s)[:-4] + suffix + ".o")
        objs.append(o)
        f = list(flags)
        head = open(s).readline()                                   # "// simvg-build-flags: ..." on a source's first line: this file only
        if head.startswith("// simvg-build-flags:"):
            f += head.split(":", 1)[1].split()
        if os.path.basename(s) == "api.hip":                        # the one object that carries the whole-source hash
            f.append(f'-DSIMVG_SOURCE_HASH="{whole}"')
        want = _sha([s] + hdrs, " ".join(f))
        have = open(o + ".sha").read().strip() if os.path.exists(o + ".sha") and os.path.exists(o) else None
        if force or have != want:
            jobs.append(([_hipcc(), *f, "-c", s, "-o", o], o, want))

    def run(job):
        cmd, o, want = job
        if verbose:
            print("[simvg_amd.build]", " ".join(cmd[-4:]), flush=True)
        if o and os.path.exists(o + ".sha"):
            os.remove(o + ".sha")
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if r.stderr.strip() and verbose:
            print(r.stderr)
        if o:
            with open(o + ".sha", "w") as fh:
                fh.write(want)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    link_want = _sha([], whole + " ".join(sorted(os.path.basename(o) for o in objs)))
    link_have = open(lib + ".sha").read().strip() if os.path.exists(lib + ".sha") and os.path.exists(lib) else None
    if jobs or link_have != link_want:
        run(([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], None, None))
        with open(lib + ".sha", "w") as fh:
            fh.write(link_want)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
