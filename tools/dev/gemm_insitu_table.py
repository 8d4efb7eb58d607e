"""gemm_nt per shape: standalone (back-to-back launches of one shape, event-timed, with sclk / package power sampled while it runs)
against in situ (the same launches inside training steps: `bench.py --breakdown`, one event pair per launch), on ONE box.
    python tools/dev/gemm_insitu_table.py        -> table on stdout"""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from simvg_amd import hip_ops as ops

M, SPLIT = 26944, 25664
# key (as --breakdown prints it), N, K, fp32 out + residual?, what
SHAPES = [("gemm_nt[26944x2304x768]", 2304, 768, False, "qkv forward"),
          ("gemm_nt[26944x768x768+res+f32]", 768, 768, True, "out-proj forward (fp32 + residual)"),
          ("gemm_nt[26944x3072x768]", 3072, 768, False, "fc1 forward / dgrad fc2"),
          ("gemm_nt[26944x768x3072+res+f32]", 768, 3072, True, "fc2 forward (fp32 + residual)"),
          ("gemm_nt[26944x768x3072]", 768, 3072, False, "dgrad fc1"),
          ("gemm_nt[26944x768x768]", 768, 768, False, "dgrad out-proj"),
          ("gemm_nt[26944x768x2304]", 768, 2304, False, "dgrad qkv")]


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception:
        return None, None
    sclk = re.search(r"sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)", out)
    pw = re.search(r"Power \(W\):\s*([\d.]+)", out)
    return (int(sclk.group(1)) if sclk else None), (float(pw.group(1)) if pw else None)


def sampled(fn_loop, seconds):
    """run fn_loop() repeatedly for `seconds`, sampling rocm-smi from a thread -> (us per launch, [sclk], [power])"""
    stop, samples = [False], []

    def watch():
        time.sleep(1.0)
        while not stop[0]:
            samples.append(smi())
            time.sleep(0.5)
    th = threading.Thread(target=watch)
    th.start()
    t_end, n, ms = time.time() + seconds, 0, 0.0
    while time.time() < t_end:
        dt, k = fn_loop()
        ms += dt
        n += k
    stop[0] = True
    th.join()
    sc = [s for s, _ in samples if s]
    pw = [p for _, p in samples if p]
    return ms / n * 1e3, sc, pw


def standalone():
    dev = "cuda"
    res = {}
    for key, N, K, f32res, what in SHAPES:
        a = torch.randn(M, K, device=dev).to(ops.LP())
        w = (torch.randn(2, N, K, device=dev) * K ** -0.5).to(ops.LP())
        bias = torch.randn(2, N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32res else ops.LP())
        resid = torch.randn(M, N, device=dev) if f32res else None

        def loop(reps=200):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.gemm_nt(a, w, bias=bias, out=out, split=SPLIT, residual=resid)
            e1.record(); e1.synchronize()
            return e0.elapsed_time(e1), reps
        loop(30)
        us, sc, pw = sampled(loop, 5.0)
        res[key] = (us, sc, pw)
    return res


def insitu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "6", "--breakdown", "--no-cpu-baseline",
                        "--no-forward-test", "--no-extras"], capture_output=True, text=True, cwd=ROOT)
    res = {}
    for l in r.stderr.splitlines():
        m = re.match(r"\[breakdown\] (gemm_nt\[\S+\])\s+calls/step\s+([\d.]+)\s+ms/step\s+([\d.]+)", l)
        if m:
            res[m.group(1)] = (float(m.group(2)), float(m.group(3)) * 1e3 / float(m.group(2)))
    return res


def insitu_clock():
    """sclk / power while plain training steps run (a second bench process, sampled from here)"""
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "400", "--warmup", "6", "--no-cpu-baseline",
                          "--no-forward-test", "--no-extras"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    time.sleep(25)
    samples = []
    while p.poll() is None and len(samples) < 16:
        samples.append(smi())
        time.sleep(0.5)
    p.wait()
    return [s for s, _ in samples if s], [q for _, q in samples if q]


if __name__ == "__main__":
    ins = insitu()
    sc_i, pw_i = insitu_clock()
    sa = standalone()
    mean = lambda v: sum(v) / len(v) if v else float("nan")
    print(f"in situ (whole training step running): sclk {mean(sc_i):.0f} MHz, package {mean(pw_i):.0f} W ({len(sc_i)} samples)")
    print("| launch | per step | standalone us (TFLOP/s) | sclk MHz / W standalone | in situ us (TFLOP/s) | in situ - standalone, us per step |")
    print("|---|---|---|---|---|---|")
    tot = 0.0
    for key, N, K, f32res, what in SHAPES:
        us, sc, pw = sa[key]
        calls, us_i = ins.get(key, (0.0, float("nan")))
        fl = 2.0 * M * N * K
        d = (us_i - us) * calls
        tot += d if d == d else 0.0
        print(f"| {what} [{N} x {K}] | {calls:.0f} | {us:.1f} ({fl / us / 1e6:.0f}) | {mean(sc):.0f} / {mean(pw):.0f} | {us_i:.1f} ({fl / us_i / 1e6:.0f}) | {d:+.0f} |")
    print(f"sum over the step: {tot:+.0f} us")
