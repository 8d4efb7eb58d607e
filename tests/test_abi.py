"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/simvg_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "simvg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(simvg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from simvg_amd import build, _lib
    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/simvg_hip.h but not exported"
    # and the ctypes binding covers the same set
    assert set(_lib.exported_symbols()) == set(names), set(_lib.exported_symbols()) ^ set(names)
    lib2 = _lib.load()
    assert lib2.simvg_version() >= 2
    assert _lib.lowp_format() in ("fp16", "bf16")


def test_argument_errors_are_reported_not_thrown():
    from simvg_amd import _lib
    lib = _lib.load()
    # K not a multiple of 64 -> argument error before any launch (safe without a GPU)
    rc = lib.simvg_gemm_nt(None, 8, None, 0, 8, None, 0, None, 8, 0, None, 0, None, 0, None, 1, 1, 4, 4, 30, 0, 0, 1.0, None)
    assert rc < 0
    assert b"multiple of 64" in lib.simvg_last_error()


def test_gemm_nt_dispatcher_plans_the_baseline_launches_as_documented():
    """`simvg_gemm_nt_plan`: the dispatcher's choice of kernel / tile extent (rounds x rows per tile on 256 CUs) as data, no launch.
    Pins the plan of every gemm_nt form of BASELINE's configurations and of the shapes the GPU kernel tests use to
    reach each kernel (tests/test_kernels_gpu.py::test_gemm_nt)."""
    import re
    from simvg_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "simvg_hip.h")).read()
    ids = {int(v): k for k, v in re.findall(r"#define SIMVG_GEMM_PLAN_(\w+) (\d+)", hdr)}
    assert len(ids) == 11

    def plan(M, N, K, split, f32=0, res=0, rs=0, act=0, aux=0, two=0):
        r = lib.simvg_gemm_nt_plan(M, N, K, split, f32, res, rs, act, aux, two)
        assert r > 0, (r, lib.simvg_last_error())
        return ids[r]
    Mb, Sb = 64 * 421, 64 * 401                   # ViT-B, 64 pairs per step (BASELINE config 2)
    assert plan(Mb, 2304, 768, Sb) == "PERSIST" and plan(Mb, 3072, 768, Sb) == "PERSIST"            # qkv, fc1 forward / dgrad fc2
    assert plan(Mb, 2304, 768, Sb, two=1) == "PERSIST_SPLIT"                                          # precise_training's qkv
    assert plan(Mb, 768, 768, Sb, f32=1, res=1, rs=1) == "TALL5" and plan(Mb, 768, 3072, Sb, f32=1, res=1, rs=1) == "TALL5"
    assert [plan(Mb, 768, K, Sb) for K in (768, 2304, 3072)] == ["TALL5"] * 3                        # the three N = 768 dgrads
    assert plan(64 * 400, 768, 3072, 0, two=1) == "TALL5"                                             # the patch kernel with hi + lo weights
    assert plan(Mb, 3072, 768, Sb, act=1, aux=1) == "256"                                             # an activation epilogue: no hand-managed form
    Ml, Sl = 32 * 421, 32 * 401                   # ViT-L, 32 pairs per step (configs 4 / 5)
    assert plan(Ml, 1024, 1024, Sl, f32=1, res=1) == "T224" and plan(Ml, 1024, 4096, Sl, f32=1, res=1) == "T224"
    assert [plan(Ml, 1024, K, Sl) for K in (1024, 3072, 4096)] == ["T224"] * 3
    assert plan(Ml, 3072, 1024, Sl) == "PERSIST" and plan(Ml, 4096, 1024, Sl) == "PERSIST"
    assert plan(Ml, 1024, 4096, Sl, f32=1, res=1, two=1) == "T224"                                    # precise_training's fc2 (ViT-L)
    # forward_test at small batches: the latency kernel while its 64 x 64 tiles fit about one residency round, then one round of 224-row tiles
    assert plan(421, 768, 768, 401) == "LAT" and plan(8 * 421, 768, 768, 8 * 401) == "LAT" and plan(8 * 421, 3072, 768, 8 * 401) == "T224"
    # half the batch (what forward_test's batch-independence property compares with): another kernel, the same bits
    assert plan(32 * 421, 768, 768, 32 * 401, f32=1, res=1) == "160"
    # the shapes tests/test_kernels_gpu.py uses to reach the remaining kernels
    assert plan(300, 12800, 64, 200) == "128" and plan(12000, 320, 64, 11000) == "256K32" and plan(7300, 768, 192, 6000) == "160"
    assert plan(20011, 768, 192, 18003, f32=1, res=1, rs=1) == "TALL4" and plan(10300, 1024, 256, 9800, act=1, aux=1) == "224"
    assert plan(7000, 2304, 128, 6500, two=1) == "PERSIST_SPLIT"
    # argument errors come back as codes
    assert lib.simvg_gemm_nt_plan(100, 64, 30, 0, 0, 0, 0, 0, 0, 0) < 0 and b"multiple of 64" in lib.simvg_last_error()


def test_grouped_gemm_rejects_bad_problem_lists():
    import ctypes as C
    from simvg_amd import _lib
    lib = _lib.load()
    assert lib.simvg_gemm_f32_grouped(None, 0, None) < 0 and b"problems" in lib.simvg_last_error()
    arr = (_lib.GemmF32Problem * 13)()
    assert lib.simvg_gemm_f32_grouped(C.byref(arr), 13, None) < 0            # more than 12 problems per launch
    one = (_lib.GemmF32Problem * 1)()                                         # M = N = K = 0
    assert lib.simvg_gemm_f32_grouped(C.byref(one), 1, None) < 0 and b"empty problem" in lib.simvg_last_error()


def test_product_refuses_cpu_tensors():
    import pytest
    import torch
    from simvg_amd import hip_ops, _lib
    with pytest.raises(_lib.SimvgHipError):
        hip_ops.gemm_nt(torch.zeros(4, 64, dtype=hip_ops.LP()), torch.zeros(4, 64, dtype=hip_ops.LP()))


def test_philox_known_answers():
    """csrc/rng.hip evaluates Philox4x32-10 on the host through the same inline function the kernel uses: the three
    known-answer vectors of the Random123 distribution (kat_vectors: philox4x32 10)."""
    import ctypes as C
    from simvg_amd import _lib
    lib = _lib.load()
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        c, k, o = (C.c_uint * 4)(*ctr), (C.c_uint * 2)(*key), (C.c_uint * 4)()
        assert lib.simvg_philox4x32(c, k, o) == 0
        assert tuple(o) == want, (ctr, [hex(x) for x in o])


def _header_arity():
    """{entry point: number of parameters} parsed from include/simvg_hip.h"""
    src = open(os.path.join(ROOT, "include", "simvg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for m in re.finditer(r"\b(simvg_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_ctypes_binding_and_the_integration_stub_have_the_headers_arity():
    """every argtypes list of simvg_amd/_lib.py has as many entries as the declaration in include/simvg_hip.h has parameters (a
    missing trailing argument is read from a garbage register, not refused), and so does the ctypes stub INTEGRATION.md shows a
    maintainer"""
    from simvg_amd import _lib
    arity = _header_arity()
    assert len(arity) >= 40
    for name, sig in _lib._SIGS.items():
        assert name in arity, name
        assert len(sig) == arity[name], (name, len(sig), arity[name])
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"lib\.simvg_ln_fwd\.argtypes = \[(.*?)\]", doc, flags=re.S)
    assert m and m.group(1).count("ctypes.") == arity["simvg_ln_fwd"], (m and m.group(1).count("ctypes."), arity["simvg_ln_fwd"])
    call = re.search(r"rc = lib\.simvg_ln_fwd\((.*?)\)\n", doc, flags=re.S).group(1)
    depth, n = 0, 1
    for ch in call:
        depth += ch in "([" 
        depth -= ch in ")]"
        n += (ch == "," and depth == 0)
    assert n == arity["simvg_ln_fwd"], n
