"""GPU: the bf16 build of the same kernels (`simvg_amd/lib/libsimvg_hip_bf16.so`, `-DSIMVG_LOWP_BF16`) under the fixture tests.

BASELINE.json's config 2 is worded "forward+backward bf16"; the shipped 16-bit operand format is fp16 (bf16's 8
significand bits cannot meet the 1e-3 box bound on trained-scale weights).  So that the bf16 wording has a test of its own in the
driver's `-m gpu` run, this module runs the reference-fixture tests (whole model forward_train / forward_test against the
fixtures recorded from the executed reference, and the encoder against the oracle) in a subprocess whose `SIMVG_HIP_LIB` points at
the bf16 library -- the library is chosen at import, one per process.  The tolerances are the ones those tests state for a bf16
build (`tests/test_model_gpu.py::_box_tol`): 1e-3 L1 on the reference-initialised fixtures, 1.5e-2 on the harsh ones; gradient
direction / norm bounds as written there."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF16_LIB = os.path.join(ROOT, "simvg_amd", "lib", "libsimvg_hip_bf16.so")

# one fixture per reference geometry / head mode (each ViT-L weight set costs 10-30 s of host time to regenerate)
MODEL_K = ("base_nq1_refinit or base_nq10_grec_refinit or large_nq10_grec_refinit or tiny_nq10_grec or (base_nq1 and not fp32) "
           "or base_nq10_grec_deconly or large_nq1_w104")


def _run(args, timeout=1500):
    assert os.path.exists(BF16_LIB), f"{BF16_LIB} missing: __graft_entry__.build() compiles it (SIMVG_LOWP=bf16 python -m simvg_amd.build)"
    env = dict(os.environ, SIMVG_HIP_LIB=BF16_LIB, SIMVG_EXPECT_LOWP="bf16")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", *args],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-120:])
    print(tail)
    assert r.returncode == 0, tail
    return r.stdout


def _counts(out):
    """(passed, skipped, [skip reasons]) of a `pytest -rs` run: an inner skip must be visible in THIS test's result -- the driver
    sees only the outer test"""
    import re
    last = [l for l in out.splitlines() if re.search(r"\b(passed|failed|skipped)\b", l) and " in " in l][-1]
    n = lambda w: int((re.search(r"(\d+) " + w, last) or [0, 0])[1])
    reasons = [l for l in out.splitlines() if l.startswith("SKIPPED")]
    listed = sum(int((re.match(r"SKIPPED \[(\d+)\]", l) or [0, 1])[1]) for l in reasons)       # "SKIPPED [6] file:line: reason"
    assert listed == n("skipped"), (last, reasons)
    return n("passed"), n("skipped"), reasons


def test_bf16_build_is_what_the_subprocess_loads():
    env = dict(os.environ, SIMVG_HIP_LIB=BF16_LIB)
    r = subprocess.run([sys.executable, "-c", "from simvg_amd import _lib; print(_lib.lowp_format(), _lib.LIB_PATH)"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split()[0] == "bf16" and r.stdout.split()[1] == BF16_LIB, r.stdout


def test_bf16_build_whole_model_vs_reference_fixtures():
    out = _run(["tests/test_model_gpu.py::test_forward_train_matches_reference", "tests/test_model_gpu.py::test_forward_test_boxes",
                "-k", MODEL_K, "-s", "-rs"])
    passed, skipped, reasons = _counts(out)
    for line in out.splitlines():          # measured bf16 deviations and any Hungarian-assignment skip, into this test's own output
        if line.startswith("[gradients") or "SKIPPED" in line:
            print("[bf16 build]", line)
    # the ONLY skip this run may contain is the named one: on trained-scale weights the bf16 build's ~1e-2 box deviations flip a
    # near-tied Hungarian assignment (seen on base_nq10_grec_deconly), after which the loss terms belong to other pairs.
    # Anything else that skipped -- or more than two such cases -- fails HERE, and the count is printed
    print(f"[bf16 build] fixture cases compared: {passed} passed, {skipped} skipped ({len(reasons)} reasons listed)")
    # (as a warning too: pytest's end-of-run warnings summary is what the driver's tail of the `-m gpu` run shows)
    import warnings
    warnings.warn(UserWarning(f"bf16 build: {passed} of {passed + skipped} fixture cases compared with the reference"
                              + (f" ({skipped} skipped: Hungarian assignment flipped by the bf16 build's box deviations)" if skipped else "")))
    assert passed >= 12, (passed, skipped)
    assert skipped <= 2, (skipped, reasons)
    for r_ in reasons:
        assert "Hungarian assignment" in r_ and "bf16 build" in r_, r_


def test_bf16_build_encoder_vs_oracle():
    out = _run(["tests/test_encoder_gpu.py", "tests/test_kernels_gpu.py", "-rs"])
    passed, skipped, reasons = _counts(out)
    print(f"[bf16 build] kernel / encoder cases: {passed} passed, {skipped} skipped")
    import warnings
    warnings.warn(UserWarning(f"bf16 build: {passed} of {passed + skipped} kernel / encoder cases compared ({skipped} skipped: "
                              "geometries the one-pass attention backward does not serve / the gradient-scale test)"))
    # the only skips of these two modules: the one-pass attention backward's switch on geometries it does not serve (a
    # parametrisation artefact: attn_bwd_one_kernel exists for 27 key tiles, on either build) and the gradient-scale test (bf16 has
    # fp32's exponent range: the build carries no gradient scale)
    assert passed >= 100, (passed, skipped, reasons)
    assert all("27-tile geometry" in r_ or "no gradient scale" in r_ for r_ in reasons), reasons


def test_bf16_build_at_baseline_full_sizes():
    """BASELINE config 2's literal dtype at its literal size: tests/test_fullsize_gpu.py (ViT-B bs 64, ViT-L bs 32 x 10 queries,
    against the full-size fixtures recorded from the executed reference) on the bf16 library, with the bf16 build's stated
    bounds (`tests/test_fullsize_gpu.py::_tol`)."""
    out = _run(["tests/test_fullsize_gpu.py", "-s", "-rs"], timeout=2400)
    passed, skipped, reasons = _counts(out)
    for line in out.splitlines():
        if line.startswith("[full size"):
            print("[bf16 build]", line)
    assert passed == 4 and skipped == 0, (passed, skipped, reasons)
