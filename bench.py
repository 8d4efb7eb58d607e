#!/usr/bin/env python
"""bench.py -- SimVG hot-path throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    N > 1: the script starts its N ranks itself (one process per GPU under torch.distributed.run on 127.0.0.1, as the
    reference's tools/dist_train.sh <cfg> N does) and refuses -- exit code 2 -- on a node with fewer than N GPUs; launched
    under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the driver's form) it joins that job,
    and refuses a WORLD_SIZE that is not N.

One "step" = one full training step of MIXDETRMB (ViT-B/32 BEiT-3, 640x640 image + 20-token expression,
num_queries=1, B = 64 per GPU, train mode: DropPath + decoder dropout) on synthetic RefCOCO-shape data already
resident in HBM: forward_train (encoder + head + on-device matcher/criterion) -> zero_grad -> backward (+ RCCL
all-reduce of the gradient arenas when N > 1) -> global-norm clip 0.15 -> Adam(amsgrad) -> bf16 weight refresh.
Nothing is skipped inside the timed region.  Prints ONE JSON line on rank 0.

The timed steps rotate through `--batches` (default 8) distinct synthetic batches (different images, expressions,
token ids and boxes), so the embedding rows touched -- and with them the Adam kernel's untouched-row skip -- and the
cache contents change from step to step as they do in training.

Extra objects in the line:
  roofline      : the dominant kernel (16-bit MFMA GEMM `gemm_nt_kernel_*`): algorithmic FLOPs of its launches divided by
                  their HIP-event-measured durations (events recorded on the launch stream during the timed steps; every
                  launch of every `--roofline-every`-th step, default 4: an event pair costs the stream ~6 us).
                  `traffic` is the PMC figure of profiles/gemm_nt_hbm_traffic.json, reported only while the kernel source
                  it was measured on (sha256 of csrc/gemm.hip) is the one this library was built from; otherwise null.
  roofline_attn : SURVEY section 8(d) metric (ii): the encoder self-attention kernels (QK^T + softmax + PV) in isolation --
                  back-to-back launches on the bench geometry (B x heads x 421 tokens x 64) between two HIP events -- and the
                  same kernels in situ (HIP-event brackets inside the timed steps).  Algorithmic FLOPs (N = 421) beside the
                  tile-padded FLOPs the MFMA pipe actually executes (N rounded up to the 64-row tile).
  ms_per_step_p50: median of the per-step times (one HIP event per step boundary on the training stream).
  host_ms_per_step: time the host needs to QUEUE a step (mean of the three fastest of the K steps: it never waits for the GPU inside a
  step, but runs at the GPU's pace once the runtime's queue is full): below ms_per_step = the GPU is the bound, the launch overhead
  of the ~460 kernels of a step is hidden behind the queue.
  cpu_baseline  : the CPU oracle (oracle/simvg_cpu.py, a restatement pinned to the reference) timed on the host
                  cores of the same box on a bounded sample (rank 0, N == 1 only): one training step at B = 8 and
                  `forward_test` at B = 1 and B = 8 (the protocol of the reference's tools/misc/inference_time.py:68-75:
                  warm-up, then the mean over repeated calls).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PAIR_FWD_BWD = {"base": 241.33e9, "large": 824.33e9}   # /32 @640, forward+backward
MFMA_BF16_PEAK_TFLOPS = 2500.0       # gfx950 dense bf16 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0                # HBM3E spec (same guide; ~6.3 TB/s measured achievable)


def model_cfg(num_queries=1, vit="base"):
    return dict(
        type="MIXDETRMB",
        vis_enc=dict(type="BEIT3", img_size=640, patch_size=32, vit_type=vit, drop_path_rate=0.1, vocab_size=64010,
                     freeze_layer=-1, vision_embed_proj_interpolate=True, pretrain=None),
        lan_enc=None, fusion=None,
        head=dict(type="TextGuidedQuerySelectKDDETRHead", num_queries=num_queries, text_max_token=20,
                  in_channels=768 if vit == "base" else 1024, embed_dim=256, decoder_freeze=False, num_classes=1,
                  aux_loss=True, num_encoder_layers=6, num_decoder_layers=3, only_decoder=True, text_embed_aug=False,
                  branch_loss_weight={"decoder": 1.0, "balanced_distill": {"token": 2.0, "distill": 1.0}},
                  distill_type="hard_weighted", prepare_target_mode="score_iou_weighted", share_predicthead=False,
                  num_token_mlp_layers=1, mlp_aux_loss=False, text_guided_query_generation=True, num_tgqg_layers=2))


def synthetic_batch(B, seed, device):
    g = torch.Generator().manual_seed(seed)
    S, T, V = 640, 20, 64010
    img = torch.randn(B, 3, S, S, generator=g)
    ids = torch.ones(B, T, dtype=torch.int64)
    pad = torch.ones(B, T, dtype=torch.int64)
    for b in range(B):
        m = int(torch.randint(2, 11, (1,), generator=g))
        ids[b, 0] = 0
        ids[b, 1:1 + m] = torch.randint(4, V, (m,), generator=g)
        ids[b, 1 + m] = 2
        pad[b, :m + 2] = 0
    xy = torch.rand(B, 2, generator=g) * 400
    wh = 32 + torch.rand(B, 2, generator=g) * 208
    gt = torch.cat([xy, xy + wh], 1).to(device)
    metas = [dict(img_shape=(S, S, 3), pad_shape=(S, S, 3), ori_shape=(S, S, 3), scale_factor=[1.0] * 4,
                  filename=f"synthetic_{b}.jpg", expression="synthetic") for b in range(B)]
    return dict(img=img.to(device), ref_expr_inds=ids.to(device), text_attention_mask=pad.to(device), img_metas=metas,
                gt_bbox=[gt[b] for b in range(B)])


def cpu_baseline(batch_size=8, max_threads=32, steps=3):
    """The oracle's training step (forward_train + backward + clip + Adam amsgrad) and its forward_test on the host cores.
    Bounded sample: the thread count is capped (eager PyTorch on hundreds of threads is slower, not faster, for the
    ~1400 small ops of this model -- measured 678 s/step with 256 threads); one B=8 training step and a few forward_test
    calls at B=1 / B=8 are timed (a few seconds each); three training steps, mean and spread reported."""
    from oracle import simvg_cpu as O, weights as W
    host_cores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(host_cores, max_threads)))
    cfg = O.make_cfg("base", 1, 640)
    sd = W.reference_init_state_dict(cfg, 1)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "empty_weight" not in k}
    sd.update(params)
    opt = torch.optim.Adam(list(params.values()), lr=5e-4, betas=(0.9, 0.98), eps=1e-9, amsgrad=True)

    def step(B, seed):
        b = W.synthetic_batch(cfg, B, seed)
        losses, _, _ = O.forward_train(sd, cfg, b["img"], b["ref_expr_inds"], b["img_metas"], b["text_attention_mask"], b["gt_bbox"])
        opt.zero_grad()
        losses["loss_total"].backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), 0.15)
        opt.step()

    step(1, 0)                    # untimed warm-up (allocator, MKL threads, Adam state)
    times = []
    for i in range(steps):        # bounded sample: `steps` training steps at B = batch_size on fresh batches (~4-5 s each)
        t0 = time.perf_counter()
        step(batch_size, 1 + i)
        times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)

    def infer(B, reps):           # tools/misc/inference_time.py:68-75: eval mode, warm-up, mean over repeated forward_test
        b = W.synthetic_batch(cfg, B, 7)
        with torch.no_grad():
            O.forward_test(sd, cfg, b["img"], b["ref_expr_inds"], b["img_metas"], b["text_attention_mask"])
            t = time.perf_counter()
            for _ in range(reps):
                O.forward_test(sd, cfg, b["img"], b["ref_expr_inds"], b["img_metas"], b["text_attention_mask"])
            return (time.perf_counter() - t) / reps

    t1, t8 = infer(1, 5), infer(8, 2)
    return dict(value=round(batch_size / dt, 4), unit="pairs/s", cores=torch.get_num_threads(), host_cores=host_cores,
                kind="port",
                sample=f"{steps} training steps (fwd+bwd+clip+Adam amsgrad) after one warm-up step, ViT-B/32 @640, B={batch_size}, fp32: "
                       f"mean {dt:.2f} s, min {min(times):.2f} s, max {max(times):.2f} s per step; value = B / mean",
                best=round(batch_size / min(times), 4), step_seconds=[round(t, 3) for t in times],
                forward_test_b1=dict(value=round(1 / t1, 3), unit="pairs/s", ms_per_call=round(t1 * 1e3, 1),
                                     sample="mean of 5 forward_test calls after 1 warm-up, B=1, fp32"),
                forward_test_b8=dict(value=round(8 / t8, 3), unit="pairs/s", ms_per_call=round(t8 * 1e3, 1),
                                     sample="mean of 2 forward_test calls after 1 warm-up, B=8, fp32"))


def attention_roofline(B, H, Nv, T, hd, device, reps=50):
    """Encoder self-attention kernels alone: `reps` back-to-back launches between two HIP events (after 5 warm-up
    launches) on random q/k/v of the bench geometry, key padding as in the synthetic batches."""
    from simvg_amd import hip_ops as ops
    N, D = Nv + T, H * hd
    g = torch.Generator(device="cpu").manual_seed(5)
    qkv = (torch.randn(B * N, 3 * D, generator=g) * 0.5).to(device).to(ops.LP())
    dout = (torch.randn(B * N, D, generator=g) * 0.1).to(device).to(ops.LP())
    pad = torch.zeros(B, T, dtype=torch.uint8)
    for b in range(B):
        pad[b, 3 + int(torch.randint(2, 11, (1,), generator=g)):] = 1
    pad = pad.to(device)
    out, lse = ops.attn_fwd(qkv, B, H, Nv, T, pad=pad)
    dqkv = torch.empty_like(qkv)

    def timed(fn):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    tq = timed(lambda: ops.attn_qk_probe(qkv, B, H, Nv, T, pad=pad)) if (N + 15) // 16 == 27 else None
    tf = timed(lambda: ops.attn_fwd(qkv, B, H, Nv, T, pad=pad, out=out))
    tb = timed(lambda: ops.attn_bwd(qkv, out, dout, lse, B, H, Nv, T, pad=pad, dqkv=dqkv))
    Np = (N + 63) // 64 * 64
    fl = lambda n, gemms: 2.0 * gemms * B * H * n * n * hd       # fwd: QK^T, PV; bwd: S, dP, dV, dQ, dK
    return dict(fwd_us=tf * 1e6, bwd_us=tb * 1e6, fwd_tf=fl(N, 2) / tf / 1e12, bwd_tf=fl(N, 5) / tb / 1e12,
                fwd_tf_padded=fl(Np, 2) / tf / 1e12, bwd_tf_padded=fl(Np, 5) / tb / 1e12, N=N, Np=Np,
                qk_us=None if tq is None else tq * 1e6, qk_tf=None if tq is None else fl(N, 1) / tq / 1e12,
                qk_tf_padded=None if tq is None else 2.0 * B * H * (27 * 16) * (27 * 16) * hd / tq / 1e12)


def traffic_stamp(kernel="gemm_nt", source="gemm.hip"):
    """profiles/gemm_nt_hbm_traffic.json -> (bytes per launch | None, provenance).  The PMC passes cannot run inside this
    process (rocprofv3 wraps it), so the committed figure is reported only while it describes the kernels that are
    running: the JSON records the sha256 of the csrc file each kernel family was measured on (gemm.hip for gemm_nt, wgrad.hip
    for wgrad_x, attention.hip for attn_fwd / attn_bwd, layernorm.hip for ln_fwd / ln_bwd, optim.hip for adam)."""
    import hashlib
    tfile = os.path.join(ROOT, "profiles", "gemm_nt_hbm_traffic.json")
    src = os.path.join(ROOT, "simvg_amd", "csrc", source)
    key = source.replace(".", "_") + "_sha256"
    try:
        j = json.load(open(tfile))
        sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
    except Exception as e:
        return None, {"stale": True, "why": repr(e)}
    prov = {"file": "profiles/gemm_nt_hbm_traffic.json", "measured": j.get("measured"), "commit": j.get("commit"), key: j.get(key)}
    if j.get(key) != sha:
        prov.update(stale=True, why=f"csrc/{source} is now {sha}: re-run tools/dev/pmc_bench.sh")
        return None, prov
    if kernel == "gemm_nt":
        return j.get("hbm_bytes_per_launch"), prov
    return (j.get("other_kernels_hbm_bytes_per_launch", {}).get(kernel, {}) or {}).get("hbm_bytes_per_launch"), prov


def bf16_line(a):
    """BASELINE.json's config says bf16; the shipped operand format is fp16 (same width, same MFMA rate, 3 more significand
    bits: bf16 cannot meet the 1e-3 box bound, DESIGN.md 'Numerics').  When the bf16 build of the same kernels exists
    (simvg_amd/lib/libsimvg_hip_bf16.so, built by __graft_entry__.build()), its throughput is measured by a short sub-run of
    this script and reported beside the fp16 line; its box parity is what tests/test_model_gpu.py holds the bf16 build to."""
    import subprocess
    lib = os.path.join(ROOT, "simvg_amd", "lib", "libsimvg_hip_bf16.so")
    if not os.path.exists(lib):
        return {"error": "simvg_amd/lib/libsimvg_hip_bf16.so not built (SIMVG_LOWP=bf16 python -m simvg_amd.build)"}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "16", "--warmup", "4", "--batch", str(a.batch), "--vit", a.vit,
           "--queries", str(a.queries), "--no-cpu-baseline", "--no-forward-test", "--no-extras"]
    env = dict(os.environ, SIMVG_HIP_LIB=lib)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SIMVG_FORCE_REDUCE"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not lines:       # e.g. a bf16 library older than csrc/ (simvg_amd/_lib.py refuses it): say so instead of an IndexError
            return {"error": "bf16 sub-run printed no line: " + (r.stderr.strip().splitlines() or ["no stderr"])[-1][:300]}
        j = json.loads(lines[-1])
        return {"dtype": j["dtype"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
                "roofline_frac_gemm_nt": j["roofline"]["frac"], "box_parity": "held to 1.5e-2 L1 (harsh fixtures measure 4e-3 / 1.1e-2): "
                "tests/test_model_gpu.py::_box_tol; the fp16 line above is the one that meets 1e-3",
                "how": "sub-run of this script with SIMVG_HIP_LIB=simvg_amd/lib/libsimvg_hip_bf16.so, 16 timed steps"}
    except Exception as e:     # never let the side measurement take the headline down
        return {"error": repr(e)}


def single_weights_line(a):
    """What `BEIT3.precise_training` (hi + lo weights in the training forward: the boxes that feed the matcher and the losses within
    1e-3 of the reference at the full batch) costs a step: a short sub-run of this script with SIMVG_PRECISE_TRAIN=0."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "16", "--warmup", "4", "--batch", str(a.batch), "--vit", a.vit,
           "--queries", str(a.queries), "--no-cpu-baseline", "--no-forward-test", "--no-extras"]
    env = dict(os.environ, SIMVG_PRECISE_TRAIN="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SIMVG_FORCE_REDUCE"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not lines:
            return {"error": "sub-run printed no line: " + (r.stderr.strip().splitlines() or ["no stderr"])[-1][:300]}
        j = json.loads(lines[-1])
        return {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "ms_per_step_p50": j["ms_per_step_p50"], "steps": j["steps"],
                "roofline_frac_gemm_nt": j["roofline"]["frac"],
                "box_parity": "token-branch boxes of the training forward at the full batch: max 1.11e-3 (ViT-B) / 1.17e-3 (ViT-L) on the harsh "
                              "fixtures (round 5's state; tools/dev/precise_train_sweep.py)",
                "how": "sub-run of this script with SIMVG_PRECISE_TRAIN=0, 16 timed steps"}
    except Exception as e:
        return {"error": repr(e)}


def reducer_overhead(a, ms_plain):
    """What the gradient exchange costs a step apart from the bytes on the links, measured on ONE GPU: sub-runs of this script under
    torch.distributed.run with one RCCL rank and SIMVG_FORCE_REDUCE=1 (every message of the N-rank schedule is issued -- 16 collectives
    over one rank, the token-id gather, the head's flat gradient buffer, the sparse text rows) against a plain sub-run of the same
    length.  Two forms: `overhead_ms` with the production op (ncclAvg: with ONE rank RCCL launches a pre-multiply kernel over every
    message, `oneRankReduce` -- 640 MB read and written on a second stream beside the backward, which no N-rank run contains), and
    `issue_only_ms` with SIMVG_REDUCE_OP=sum (a one-rank SUM is a no-op inside RCCL): what issuing the exchange costs -- the
    collectives' launches, the gather / scatter of the sparse rows, the hook's host time."""
    import socket
    import subprocess

    def sub(extra_env, launcher):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        tail = ["--gpus", "1", "--steps", "24", "--warmup", "6", "--batch", str(a.batch), "--vit", a.vit, "--queries", str(a.queries),
                "--no-cpu-baseline", "--no-forward-test", "--no-extras"]
        head = [sys.executable]
        if launcher:
            head += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", **extra_env)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SIMVG_FORCE_REDUCE", "SIMVG_REDUCE_OP"):
            if k not in extra_env:
                env.pop(k, None)
        r = subprocess.run(head + [os.path.abspath(__file__)] + tail, capture_output=True, text=True, env=env, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not lines:
            raise RuntimeError("sub-run printed no line: " + (r.stderr.strip().splitlines() or ["no stderr"])[-1][:300])
        return json.loads(lines[-1])

    try:
        j = sub(dict(SIMVG_FORCE_REDUCE="1"), True)
        # a plain sub-run of the same length beside it (same process start-up state, same box, minutes apart from the headline)
        plain = sub({}, False)["ms_per_step_p50"]
        k = sub(dict(SIMVG_FORCE_REDUCE="1", SIMVG_REDUCE_OP="sum"), True)
        red = j["reducer"]
        return {"overhead_ms": round(j["ms_per_step_p50"] - plain, 3), "issue_only_ms": round(k["ms_per_step_p50"] - plain, 3),
                "ms_per_step_p50_reduced": j["ms_per_step_p50"], "ms_per_step_p50_issue_only": k["ms_per_step_p50"],
                "ms_per_step_p50_plain": plain, "messages": red.get("messages"), "exposed_ms": (red.get("exposed") or {}).get("mean_ms"),
                "how": "three 24-step sub-runs of this script: torch.distributed.run with 1 RCCL rank and SIMVG_FORCE_REDUCE=1 (ncclAvg: "
                       "RCCL's one-rank pre-multiply kernels run beside the backward), the same with SIMVG_REDUCE_OP=sum (a one-rank SUM "
                       "is a no-op: the cost of issuing the exchange alone), and plain; medians of the per-step times"}
    except Exception as e:
        return {"error": repr(e)}


def self_launch(n, share=False):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks here -- one process per GPU under
    torch.distributed.run on 127.0.0.1 (the counterpart of the reference's launcher, tools/dist_train.sh:8-10, which takes the GPU
    count as its argument) -- and hand rank 0's one JSON line through.  Fails before anything is started when the node has
    fewer than N GPUs: a job that printed `n_gpus: 1` for `--gpus 8` would void the scaling record."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if not share and have < n:
        print(f"bench.py: --gpus {n} but torch.cuda.device_count() = {have}: refusing to run a smaller job under that name",
              file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n))))
    for k in ("RANK", "LOCAL_RANK", "SIMVG_FORCE_REDUCE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)      # stderr passes through
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    rest = [l for l in r.stdout.splitlines() if not l.startswith("{")]
    if rest:                                                                  # keep stdout to the one line
        print("\n".join(rest), file=sys.stderr)
    if r.returncode != 0 or len(lines) != 1:
        print(f"bench.py: the {n}-rank job failed (exit code {r.returncode}, {len(lines)} result lines)", file=sys.stderr)
        return r.returncode or 1
    j = json.loads(lines[0])
    if j.get("n_gpus") != n or j.get("reducer", {}).get("world") != n:
        print(f"bench.py: asked for {n} ranks, the job reports n_gpus={j.get('n_gpus')} / reducer.world={j.get('reducer', {}).get('world')}",
              file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE: 64)")
    ap.add_argument("--vit", choices=["base", "large"], default="base", help="encoder size (BASELINE metric: base)")
    ap.add_argument("--queries", type=int, default=1, help="num_queries (GRefCOCO configs: 10)")
    ap.add_argument("--batches", type=int, default=8, help="distinct synthetic batches the steps rotate through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-forward-test", action="store_true",
                    help="skip the forward_test latency section (profiling runs: keeps the kernel statistics to the training steps)")
    ap.add_argument("--breakdown", action="store_true", help="per-op HIP-event breakdown on stderr")
    ap.add_argument("--roofline-every", type=int, default=4,
                    help="bracket the gemm_nt launches with HIP events in one of every N timed steps (each event pair "
                         "costs the stream ~2 x 3 us of serialisation; N=1 times every launch of every step)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the sections measured after the timed steps (wgrad / LayerNorm / Adam rooflines, attention in "
                         "isolation, the bf16 build's line): what the bf16 sub-run and the profiling scripts pass")
    ap.add_argument("--dump-params", default=None,
                    help="after the timed steps: write {name: (sum, abs-sum, 8 sampled values)} of every parameter to this file "
                         "(torch.save; tests compare a reduced run with an unreduced one)")
    a = ap.parse_args()

    # SIMVG_BENCH_SHARE_DEVICE=1 (tests only): every rank on cuda:0 over gloo -- the control flow of a world > 1 run (barriers,
    # who leaves when, which side measurements are skipped) on a one-GPU box; RCCL refuses two ranks per device
    share = os.environ.get("SIMVG_BENCH_SHARE_DEVICE") == "1"
    if a.gpus < 1:
        sys.exit(f"bench.py: --gpus {a.gpus}: need at least one GPU")
    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:
            sys.exit(self_launch(a.gpus, share))       # `python bench.py --gpus N` = the reference's `dist_train.sh <cfg> N`
    elif int(os.environ["WORLD_SIZE"]) != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks: the line would "
                 "report a job that did not run (pass --gpus equal to --nproc-per-node)")
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist
    if not share and torch.cuda.device_count() <= local:
        sys.exit(f"bench.py: rank {rank} (LOCAL_RANK {local}) has no GPU: {torch.cuda.device_count()} visible, world {world}")
    if share:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("SIMVG_FORCE_REDUCE") == "1"
    if use_dist:
        if os.environ.get("NCCL_DEBUG") == "VERSION":    # RCCL prints its version banner on STDOUT: keep stdout to the one JSON line
            os.environ.pop("NCCL_DEBUG")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        if dist.get_world_size() != a.gpus:
            sys.exit(f"bench.py: the {dist.get_backend()} world has {dist.get_world_size()} ranks, --gpus says {a.gpus}")
    from simvg_amd.models import build_model
    from simvg_amd.dist import GradReducer
    from simvg_amd.core import build_optimizer
    from simvg_amd import hip_ops

    torch.manual_seed(1234)
    model = build_model(model_cfg(a.queries, a.vit)).to(device).train()
    if use_dist:   # identical replicas
        for p in model.parameters():
            dist.broadcast(p.data, 0)
    B = a.batch
    batches = [synthetic_batch(B, 1000 + 64 * rank + i, device) for i in range(max(1, a.batches))]
    batch = batches[0]
    counter = [0]
    model.vis_enc._ensure_engine(device)
    # the reference's optimizer construction (tools/train.py:78-96 + configs: Adam amsgrad, lr 5e-4, vis_enc lr/10)
    named = list(model.named_parameters())
    groups = [{"params": [p for n, p in named if "vis_enc" in n and p.requires_grad], "lr": 5e-5},
              {"params": [p for n, p in named if "lan_enc" in n and p.requires_grad], "lr": 5e-4},
              {"params": [p for n, p in named if "lan_enc" not in n and "vis_enc" not in n and p.requires_grad], "lr": 5e-4}]
    opt = build_optimizer(dict(type="Adam", lr=5e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0, amsgrad=True), groups,
                          model=model)
    assert type(opt).__name__ == "FlatAdam"
    reducer = GradReducer(model)
    reducer.timing = reducer.active         # two events per step around finish()'s waits: the exposed part of the exchange

    def step():
        batch = batches[counter[0] % len(batches)]
        counter[0] += 1
        losses, _ = model(batch["img"], batch["ref_expr_inds"], batch["img_metas"], return_loss=True,
                          text_attention_mask=batch["text_attention_mask"], gt_bbox=batch["gt_bbox"], rescale=False)
        opt.zero_grad()
        reducer.begin()
        losses["loss_total"].backward()
        reducer.finish()
        opt.clip_grad_norm(0.15)
        opt.step()
        return losses

    # the step runs on a non-default HIP stream (what `train_model` does as well; required by the optional hipGraph
    # replay of the head, SIMVG_HEAD_GRAPH=1, see simvg_amd/graphs.py).  The roofline events are recorded on that stream.
    from simvg_amd.graphs import training_stream
    with training_stream(device):
        # set-up, not warm-up: lazily created workspaces / constants (and, with SIMVG_HEAD_GRAPH=1, the capture of the head's
        # hipGraphs after four steps with the same input signature) happen here so that they can never land in the timed
        # steps whatever --warmup is.  No optimizer step: the weights the warm-up starts from are untouched.
        for _ in range(4):
            losses, _ = model(batch["img"], batch["ref_expr_inds"], batch["img_metas"], return_loss=True,
                              text_attention_mask=batch["text_attention_mask"], gt_bbox=batch["gt_bbox"], rescale=False)
            opt.zero_grad()
            reducer.begin()
            losses["loss_total"].backward()
            reducer.finish()
        opt.zero_grad()
        for _ in range(a.warmup):
            step()
        timer = hip_ops.KernelTimer(only=None if a.breakdown else {"gemm_nt", "attn_fwd", "attn_bwd"}) if rank == 0 else None
        every = 1 if a.breakdown else max(1, a.roofline_every)
        sampled_steps = len(range(0, a.steps, every))
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        reducer._exposed = []               # set-up / warm-up steps are not part of the exposed-exchange statistic
        t0 = time.perf_counter()
        host_t = [t0]
        for i in range(a.steps):
            marks[i].record()
            hip_ops.set_timer(timer if i % every == 0 else None)
            losses = step()
            host_t.append(time.perf_counter())
        marks[a.steps].record()
        # time the host needs to QUEUE a step: it never waits for the GPU inside one, but once it is ~1000 launches ahead the
        # runtime's queue is full and it proceeds at the GPU's pace -- the first steps after the synchronisation show its own
        host_dt = sorted(b - a_ for a_, b in zip(host_t, host_t[1:]))[:3]
        host_dt = sum(host_dt) / len(host_dt)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        hip_ops.set_timer(None)
        exposed = reducer.exposed_ms() if reducer.active else None
    if use_dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss_val = float(losses["loss_total"].detach())
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    p50 = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    if rank == 0 and a.dump_params:
        torch.cuda.synchronize()
        digest = {}
        for n, p in model.named_parameters():
            t = p.detach().double().reshape(-1)
            k = min(8, t.numel())          # integer arithmetic: a float32 linspace rounds past the end of a 49 M element tensor
            idx = torch.tensor([(t.numel() - 1) * e // max(k - 1, 1) for e in range(k)], device=t.device)
            digest[n] = (float(t.sum()), float(t.abs().sum()), t[idx].cpu(), t.numel())
        torch.save(dict(params=digest, loss=loss_val, reducer=dict(reducer.last_stats, active=reducer.active)), a.dump_params)
    if use_dist:
        # every rank leaves the job here: nothing below this line may contain a collective (the extra measurements of rank 0 are
        # single-process by construction: with world > 1 they are skipped, see `extras`)
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    # side measurements (extra bracketed steps, isolated attention, forward_test, bf16 sub-run, CPU baseline) only in a
    # single-process run: at world > 1 a step contains collectives, and the other ranks have left
    extras = not a.no_extras and not use_dist
    from simvg_amd import _lib as _simvg_lib
    _lowp = _simvg_lib.lowp_format()      # "fp16" (default build) or "bf16": the MFMA operand / storage format, fp32 accumulate
    pairs = world * B * a.steps
    value = pairs / dt
    summ = timer.summary()
    g = dict(calls=0, ms=0.0, flops=0.0, bytes=0.0)     # every gemm_nt launch (--breakdown keys them per shape)
    for k, d in summ.items():
        if k.startswith("gemm_nt"):
            for f in g:
                g[f] += d[f]
    ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
    traffic, traffic_src = traffic_stamp()
    H, hd = (12, 64) if a.vit == "base" else (16, 64)
    Nv_tok, T_tok = (640 // 32) ** 2 + 1, 20
    extra = {}
    if extras:
        # wgrad / LayerNorm / Adam launches bracketed by HIP events in four EXTRA steps after the timed region (bracketing ~150
        # more launches per step inside it would cost the headline 0.2 ms per step)
        with training_stream(device):
            t2 = hip_ops.KernelTimer(only={"gemm_tn", "wgrad_reduce", "ln_fwd", "ln_bwd", "adam"})
            hip_ops.set_timer(t2)
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            hip_ops.set_timer(None)
        extra = t2.summary()
    with training_stream(device):
        iso = attention_roofline(B, H, Nv_tok, T_tok, hd, device) if extras else None
    situ = {k: summ[k] for k in ("attn_fwd", "attn_bwd") if k in summ}
    out = {
        "metric": "image-text pairs/sec (whole node), RefCOCO 640x640 bs=64/GPU, 1/2/4/8 MI355X",
        "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 3), "ms_per_step_p50": round(p50, 3), "host_ms_per_step": round(host_dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": _lowp, "data": "synthetic",
        "config": {"workload": f"ViT-{'B' if a.vit == 'base' else 'L'}/32 SimVG (MIXDETRMB), synthetic RefCOCO 640x640 + 20-token expr, num_queries={a.queries}, "
                               f"full training step: forward+backward {_lowp} MFMA operands (fp32 accumulate, fp32 residual/master), "
                               "DropPath+dropout on, clip 0.15, Adam(amsgrad)",
                   "global_batch": world * B, "per_gpu_batch": B, "tokens_per_pair": 421, "distinct_batches": len(batches),
                   "parallelism": f"dp{world}", "loss_total": round(loss_val, 4)},
        "model_tflops_per_gpu": round(value / world * FLOP_PER_PAIR_FWD_BWD[a.vit] / 1e12, 2),
        "model_mfma_frac": round(value / world * FLOP_PER_PAIR_FWD_BWD[a.vit] / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
        "roofline": {"kernel": f"gemm_nt ({_lowp} MFMA 16x16x32, fp32 accumulate; every launch: persistent 16-wave 256x256x64 tiles for N >= 2304, "
                               "one round of 16-wave 320x256x64 tiles for N = 768 (gemm_nt_kernel_tall5_*), LDS-DMA with the swizzle on the source "
                               "address, counted waits; the hi + lo launches of precise_training at their ALGORITHMIC FLOPs)", "bound": "mfma",
                     "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "sustained_mfma_ceiling": {"value": 1930.0, "unit": "TFLOP/s", "frac_of_it": round(ach / 1930.0, 4),
                                                "source": "tools/dev/probes/mfma_power.hip (profiles/r03_sweeps.md): a register-only loop of "
                                                          "v_mfma_f32_16x16x32_f16 on random operands, 16 waves per CU, sustains 1917-1933 TFLOP/s "
                                                          "(package at 1.31-1.34 kW, sclk 1.75-2.0 GHz); 2424 on all-zero operands -- the 2.5 PFLOP/s "
                                                          "peak is a 2.4 GHz figure the part does not hold on real data"},
                     "algorithmic_bytes_per_launch": round(g["bytes"] / g["calls"]),
                     "launches": g["calls"], "avg_launch_us": round(g["ms"] / g["calls"] * 1e3, 2),
                     "timed_steps": sampled_steps,
                     "share_of_step": round(g["ms"] / (dt * 1e3 * sampled_steps / a.steps), 4)},
    }
    out["reducer"] = dict(reducer.last_stats, active=reducer.active) if reducer.active else {"active": False, "world": world}
    if reducer.active:
        sched = {}
        for k, nbytes in reducer.last_schedule:
            kk = "layer" if k.startswith("layer:") else k          # ("layer:<i>" or, for a group of layers, "layer:<top>-<bottom>")
            sched.setdefault(kk, [0, 0])
            sched[kk][0] += 1
            sched[kk][1] += nbytes
        out["reducer"]["schedule"] = {"order": [k for k, _ in reducer.last_schedule],
                                      "messages_bytes": {k: {"messages": v[0], "bytes": v[1]} for k, v in sched.items()}}
        out["reducer"]["exposed"] = dict(exposed or {}, what="time per step the training stream waits for RCCL in GradReducer.finish() "
                                         "(HIP events around the waits): the part of the exchange NOT hidden under the backward")
    if "gemm_tn" in extra:
        d = extra["gemm_tn"]
        tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
        wt, wsrc = traffic_stamp("wgrad_x", "wgrad.hip")
        # the XCD-partitioned kernel leaves its row partitions' partial sums in slabs; one batched launch per encoder layer
        # sums them into dW (ops.WgradReduceBatch): its time belongs to the weight gradient, so achieved / frac include it
        r = extra.get("wgrad_reduce", dict(calls=0, ms=0.0, bytes=0.0))
        tf_all = d["flops"] / ((d["ms"] + r["ms"]) * 1e-3) / 1e12
        out["roofline_wgrad"] = {
            "kernel": "weight + bias gradients of the encoder / head-memory Linears: wgrad_sq_kernel (256x256 tiles, 8 waves in ping-pong phases: fc1, fc2) / "
                      "wgrad_x_kernel (XCD-partitioned 12-wave tiles: qkv, out-proj), partial sums of the row partitions to slabs + "
                      "wgrad_slab_reduce_kernel (one launch per encoder layer), and the generic "
                      "gemm_tn kernels, every launch of a step", "bound": "mfma", "achieved": round(tf_all, 2),
            "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf_all / MFMA_BF16_PEAK_TFLOPS, 4),
            "first_stage_only": {"achieved": round(tf, 2), "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4)},
            "launches_per_step": d["calls"] // 4, "avg_launch_us": round(d["ms"] / d["calls"] * 1e3, 2),
            "ms_per_step": round((d["ms"] + r["ms"]) / 4, 3), "ms_per_step_first_stage": round(d["ms"] / 4, 3),
            "slab_reduce": None if not r["calls"] else {
                "launches_per_step": r["calls"] // 4, "ms_per_step": round(r["ms"] / 4, 3),
                "achieved_GBps": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1), "bytes_per_launch": round(r["bytes"] / r["calls"])},
            "algorithmic_bytes_per_launch": round(d["bytes"] / d["calls"]),
            "traffic": wt, "traffic_source": wsrc, "method": "HIP events around every launch of 4 extra steps after the timed region"}
    hb = {}
    for k, label in (("ln_fwd", "LayerNorm forward (+ GELU of the FFN)"), ("ln_bwd", "LayerNorm backward (+ GELU', residual add)"),
                     ("adam", "clip + Adam(amsgrad) over the flat arenas")):
        if k in extra:
            d = extra[k]
            gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            tr, tsrc = traffic_stamp(k, "optim.hip" if k == "adam" else "layernorm.hip")
            hb[k] = {"what": label, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                     "launches_per_step": d["calls"] // 4, "ms_per_step": round(d["ms"] / 4, 3),
                     "algorithmic_bytes_per_launch": round(d["bytes"] / d["calls"]), "traffic": tr,
                     "traffic_over_algorithmic": None if not tr else round(tr / (d["bytes"] / d["calls"]), 3), "traffic_source": tsrc}
    if hb:
        out["hbm_kernels"] = dict(hb, bound="hbm", note="algorithmic bytes / HIP-event time of every launch in 4 extra steps; "
                                  "8 TB/s spec, ~6.3 TB/s achievable (MI355X_MICROARCH.md)")
    if iso is not None:
      out["roofline_attn"] = {
          "kernel": f"encoder self-attention (QK^T + key-padding softmax + PV; {_lowp} MFMA 16x16x32), B={B} x {H} heads x {iso['N']} tokens x {hd}",
          "bound": "mfma", "achieved": round(iso["fwd_tf"], 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
          "frac": round(iso["fwd_tf"] / MFMA_BF16_PEAK_TFLOPS, 4),
          "isolated": {"fwd_us": round(iso["fwd_us"], 2), "bwd_us": round(iso["bwd_us"], 2),
                       "fwd_tflops": round(iso["fwd_tf"], 2), "bwd_tflops": round(iso["bwd_tf"], 2),
                       "fwd_tflops_tile_padded": round(iso["fwd_tf_padded"], 2), "bwd_tflops_tile_padded": round(iso["bwd_tf_padded"], 2),
                       "bwd_frac": round(iso["bwd_tf"] / MFMA_BF16_PEAK_TFLOPS, 4), "padded_tokens": iso["Np"],
                       "method": "50 back-to-back launches between two HIP events after 5 warm-up launches"},
          "in_situ": {k: {"avg_launch_us": round(d["ms"] / d["calls"] * 1e3, 2), "launches": d["calls"],
                          "tflops": round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2)} for k, d in situ.items()},
      }
      if iso.get("qk_us") is not None:
          out["roofline_attn"]["qk_only"] = {
              "what": "the QK^T contraction of the forward kernel alone (simvg_attn_qk_probe: same K staging in LDS, same 16x16x32 MFMAs, "
                      "row maxima stored; no exponentials, no row sums, no PV): what north_star's '>= 60 % of MFMA peak on the encoder QK^T "
                      "GEMM' prices, beside the fused kernel it is part of",
              "us": round(iso["qk_us"], 2), "tflops": round(iso["qk_tf"], 2), "frac": round(iso["qk_tf"] / MFMA_BF16_PEAK_TFLOPS, 4),
              "tflops_tile_padded": round(iso["qk_tf_padded"], 2),
              "bytes_bound": "it reads q and k of every head once: 2 x B x N x D x 2 B = %.1f MB -> %.1f us at 6.3 TB/s: an isolated "
                             "d = 64 contraction is HBM-bound at %.2f of MFMA peak however it is written (arithmetic intensity N / 2 FLOP per byte)"
                             % (4.0 * B * iso["N"] * H * hd / 1e6, 4.0 * B * iso["N"] * H * hd / 6.3e12 * 1e6,
                                2.0 * B * H * iso["N"] * iso["N"] * hd / (4.0 * B * iso["N"] * H * hd / 6.3e12) / 1e12 / MFMA_BF16_PEAK_TFLOPS)}
      # HBM bytes per launch by PMC beside the algorithmic bytes of a call: forward reads qkv and writes o (+ lse), backward reads
      # qkv, o, dO and writes dqkv.  The backward is ONE kernel per call for the path's geometry (csrc/attention_bwd1.hip) unless
      # SIMVG_ATTN_BWD1=0 selects the dq + dkv pair (two launches per call)
      Mrows, Dm = B * iso["N"], H * hd
      alg = {"attn_fwd": 2.0 * Mrows * (3 * Dm + Dm), "attn_bwd": 2.0 * Mrows * (3 * Dm + Dm + Dm + 3 * Dm)}
      one_pass = os.environ.get("SIMVG_ATTN_BWD1", "1") != "0" and (iso["N"] + 15) // 16 == 27
      out["roofline_attn"]["backward_kernels_per_call"] = 1 if one_pass else 2
      for k in ("attn_fwd", "attn_bwd"):
          tr, tsrc = traffic_stamp(k, "attention_bwd1.hip" if (k == "attn_bwd" and one_pass) else "attention.hip")
          per_call = None if tr is None else tr * (2 if (k == "attn_bwd" and not one_pass) else 1)
          out["roofline_attn"]["traffic_" + k] = {"hbm_bytes_per_call": per_call, "algorithmic_bytes_per_call": round(alg[k]),
                                                  "traffic_over_algorithmic": None if not per_call else round(per_call / alg[k], 3),
                                                  "traffic_source": tsrc}
      out["roofline_attn"]["traffic"] = out["roofline_attn"]["traffic_attn_fwd"]["hbm_bytes_per_call"]
    # forward_test latency / throughput on the GPU (same protocol as the CPU figures of cpu_baseline: warm-up, mean of repeated
    # calls with a synchronisation after each call -- tools/misc/inference_time.py:68-75 of the reference)
    model.eval()
    infer = {}
    with torch.no_grad(), training_stream(device):
        for nb, reps in (() if (a.no_forward_test or not extras) else ((1, 20), (8, 20), (B, 5))):
            bb = synthetic_batch(nb, 4242, device)
            kw = dict(return_loss=False, text_attention_mask=bb["text_attention_mask"], with_bbox=True, with_mask=False, rescale=False)
            for _ in range(6):        # (the hipGraph of a small batch is captured on its third call: capture stays in the warm-up)
                model(bb["img"], bb["ref_expr_inds"], bb["img_metas"], **kw)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                model(bb["img"], bb["ref_expr_inds"], bb["img_metas"], **kw)
                torch.cuda.synchronize()
            ms = (time.perf_counter() - t1) / reps * 1e3
            infer[f"b{nb}"] = {"ms_per_call": round(ms, 3), "pairs_per_s": round(nb / ms * 1e3, 1)}
        if infer and getattr(model.vis_enc, "precise_inference", False):
            # the same calls with single 16-bit weights (round 3's forward_test): what `precise_inference` costs
            model.vis_enc.precise_inference = False
            bb = synthetic_batch(B, 4242, device)
            kw = dict(return_loss=False, text_attention_mask=bb["text_attention_mask"], with_bbox=True, with_mask=False, rescale=False)
            for _ in range(3):
                model(bb["img"], bb["ref_expr_inds"], bb["img_metas"], **kw)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                model(bb["img"], bb["ref_expr_inds"], bb["img_metas"], **kw)
                torch.cuda.synchronize()
            ms = (time.perf_counter() - t1) / 5 * 1e3
            infer[f"b{B}_single_16bit_weights"] = {"ms_per_call": round(ms, 3), "pairs_per_s": round(B / ms * 1e3, 1)}
            model.vis_enc.precise_inference = True
    model.train()
    if infer:
        out["forward_test"] = dict(infer, note="MIXDETRMB.forward_test incl. post-processing, one synchronisation per call; batches <= 16 replay "
                                   "encoder + head as one hipGraph per input signature (simvg_amd/graphs.py::InferenceGraphs); the patch "
                                   "kernel and the qkv projection of the first half of the layers (ViT-L: qkv + fc2 of every layer) carry hi + lo "
                                   "16-bit weights in this forward (precise_inference, "
                                   "simvg_gemm_nt_split: every box of a full batch within 1e-3 of the reference, tests/test_fullsize_gpu.py); "
                                   "`*_single_16bit_weights` = the same call without it")
    if a.breakdown:
        tot = dt * 1e3
        for k, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["flops"] else 0.0
            gb = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            print(f"[breakdown] {k:34s} calls/step {d['calls'] / a.steps:7.1f}  ms/step {d['ms'] / a.steps:8.3f} "
                  f"({100 * d['ms'] / tot:5.1f} %)  {tf:8.1f} TFLOP/s  {gb:8.1f} GB/s(algorithmic)", file=sys.stderr)
    if extras and _lowp == "fp16" and not os.environ.get("SIMVG_HIP_LIB"):
        out["bf16_line"] = bf16_line(a)
    enc = model.vis_enc
    out["precise_training"] = {"layers": dict(enc._precise_training_depth) if enc.wbs else 0, "which": list(enc.precise_training_which),
                               "split_linears_per_step": len(enc.wbs),
                               "what": "Linears whose weight the TRAINING forward carries as a hi + lo pair of 16-bit numbers (2 x the MFMA work of "
                                       "those launches, inside the timed step and inside `roofline` at their algorithmic FLOPs): every box of "
                                       "the full batch within 1e-3 of the reference (tests/test_fullsize_gpu.py)"}
    if extras and not os.environ.get("SIMVG_HIP_LIB") and enc.wbs and os.environ.get("SIMVG_PRECISE_TRAIN") is None:
        out["precise_training"]["single_weights_line"] = single_weights_line(a)
    if extras and not os.environ.get("SIMVG_HIP_LIB"):
        out["reducer"]["overhead_one_gpu"] = reducer_overhead(a, p50)
    if extras and not a.no_cpu_baseline and a.vit == "base" and a.queries == 1:
        try:
            out["cpu_baseline"] = cpu_baseline()
        except Exception as e:   # the oracle is a checker; never let it take the GPU number down
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
