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
