"""GPU: the data-parallel path with the REAL model and kernels, two processes on the one MI355X of the test box.
RCCL refuses two ranks on one device, so the processes talk over gloo (which all-reduces HIP tensors through the host);
everything else -- hand-sequenced backward with per-layer reduction hooks, head message, loss normalisers averaged before
the head -- is the production code path.  Check: the averaged gradients of two ranks with different half-batches equal
the gradients of one process on the concatenated batch (what the reference's DDP + reduced num_boxes compute)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    from test_tools_gpu import _tiny_model, _batch
    from test_graph_gpu import _no_dropout
    cfg, model = _tiny_model(7)
    _no_dropout(model)
    model.train()
    return cfg, model, _batch


def _grads(model):
    return {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}


def _run(model, batch):
    losses, _ = model(**batch, rescale=False)
    for p in model.parameters():
        p.grad = None
    return losses


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SIMVG_DENSE_EMBED_REDUCE="1")  # gloo cannot all_gather HIP ids
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from simvg_amd.dist import GradReducer
    from simvg_amd.graphs import training_stream
    cfg, model, _batch = _setup()
    red = GradReducer(model)
    full = _batch(cfg, B=4, seed=31)
    half = {k: (v[2 * rank: 2 * rank + 2] if torch.is_tensor(v) or isinstance(v, list) else v) for k, v in full.items()}
    with training_stream():
        losses = _run(model, half)
        red.begin()
        losses["loss_total"].backward()
        red.finish()
        torch.cuda.synchronize()
    out[rank] = (_grads(model), {k: float(v) for k, v in losses.items()}, red.last_late)
    dist.destroy_process_group()


def test_two_ranks_average_to_the_single_process_gradient_on_the_global_batch():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    g0, l0, late0 = out[0]
    g1, l1, late1 = out[1]
    assert late0 == 0 and late1 == 0
    assert g0.keys() == g1.keys()
    for n in g0:                                   # both replicas hold the same averaged gradient
        assert torch.equal(g0[n], g1[n]), n
    # single process, global batch
    from simvg_amd.graphs import training_stream
    cfg, model, _batch = _setup()
    full = _batch(cfg, B=4, seed=31)
    with training_stream():
        losses = _run(model, full)
        losses["loss_total"].backward()
        torch.cuda.synchronize()
    ref = _grads(model)
    assert abs(0.5 * (l0["loss_total"] + l1["loss_total"]) - float(losses["loss_total"])) <= 2e-3 * abs(float(losses["loss_total"]))
    errs = sorted(((float((g0[n] - r).norm()) / max(float(r.norm()), 1e-8), n) for n, r in ref.items()), reverse=True)
    print("largest relative gradient differences:", [(n, round(e, 4)) for e, n in errs[:4]])
    # not exact by construction: the distillation weight w = mean(matched weights) is a per-process (local batch) quantity
    # in the reference too (tgqs_kd_detr_head.py:340-350, no all-reduce), so the token-branch terms (1-w) / w differ
    # between two half batches and the global batch (largest on head.bbox_embed_token.*: ~2.5 %), plus bf16 operand
    # rounding; a wrong normaliser or a missed message would show as O(1)
    assert errs[0][0] <= 6e-2, errs[0]
    assert sum(e for e, _ in errs) / len(errs) <= 1e-2


def test_bench_under_torchrun_takes_the_rccl_branches_and_matches_the_unreduced_run(tmp_path):
    """`bench.py --gpus 1` launched the way the driver launches N > 1 (torch.distributed.run, backend nccl = RCCL), with
    SIMVG_FORCE_REDUCE=1 so that the one-rank job runs the whole exchange: ReduceOp.AVG inside the collective,
    all_gather_into_tensor of the token ids, the sparse text-row message, the per-layer messages issued from inside the
    backward on the training stream; SIMVG_DIST_CHECK=1 adds the same-id-list assertion.  An all-reduce over one rank is the
    identity, so three optimizer steps must leave the same parameters as the plain single-process run (up to the
    summation order of the weight-gradient atomics: 1e-6 of each tensor's abs-sum)."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    common = ["--steps", "3", "--warmup", "0", "--batch", "4", "--batches", "2", "--no-cpu-baseline", "--no-forward-test"]
    env = dict(os.environ, SIMVG_FORCE_REDUCE="1", SIMVG_DIST_CHECK="1", MASTER_ADDR="127.0.0.1")
    f_red, f_plain = str(tmp_path / "reduced.pt"), str(tmp_path / "plain.pt")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", *common, "--dump-params", f_red],
                       capture_output=True, text=True, env=env, cwd=root, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    red = line["reducer"]
    assert red["active"] and red["avg_in_collective"] and red["gather_into_tensor"] and red["sparse_rows"] == 4 * 20
    assert red["messages"] >= 5 + 2 and line["n_gpus"] == 1           # 5 groups of layers + head + embeddings (+ sparse rows)
    env2 = {k: v for k, v in os.environ.items() if k not in ("SIMVG_FORCE_REDUCE", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", *common, "--dump-params", f_plain],
                        capture_output=True, text=True, env=env2, cwd=root, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    a, b = torch.load(f_red, weights_only=False), torch.load(f_plain, weights_only=False)
    assert not b["reducer"]["active"]
    assert abs(a["loss"] - b["loss"]) <= 1e-3 * abs(b["loss"])
    # Adam normalises the update: an element whose gradient is noise-sized (summation order of the weight-gradient atomics) may
    # move by a full lr per step in either run, so samples are compared against the largest possible move (3 steps x lr 5e-4)
    # and the tensors' abs-sums relatively
    worst_sum = worst_val = 0.0
    for n, (s1, a1, v1, numel) in a["params"].items():
        s2, a2, v2, _ = b["params"][n]
        # abs-sum: 0.2 % of the tensor, plus 5 % of its elements moving by the three steps' maximum (zero-initialised biases)
        worst_sum = max(worst_sum, abs(a1 - a2) / (2e-3 * a2 + 0.05 * numel * 1.5e-3))
        worst_val = max(worst_val, float((v1 - v2).abs().max()))
    print(f"reduced vs plain after 3 steps: abs-sum difference / allowance {worst_sum:.2f}, sampled values {worst_val:.2e}")
    assert worst_sum <= 1.0 and worst_val <= 1.6e-3, (worst_sum, worst_val)


def test_bench_with_two_ranks_finishes_and_reports_the_whole_job(tmp_path):
    """`python bench.py --gpus 2` (self-launched ranks) on the one GPU of the test box (gloo, SIMVG_BENCH_SHARE_DEVICE=1): rank 0
    prints ONE line for the whole job (n_gpus 2, global batch 2 x B, value = pairs of both ranks / max time), the other rank
    leaves without printing, nothing after the timed region waits for a rank that has left (the side measurements of a
    single-process run are skipped), and the reducer reports one message per layer + head + embeddings."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, SIMVG_BENCH_SHARE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    # round 5: no launcher around it -- `python bench.py --gpus 2` starts its two ranks itself (bench.py::self_launch)
    env = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batch", "4", "--batches", "2"],
                       capture_output=True, text=True, env=env, cwd=root, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["config"]["parallelism"] == "dp2"
    assert line["scaling"] == "weak" and line["value"] > 0
    assert abs(line["value"] - 8 * 3 / (line["ms_per_step"] * 3e-3)) <= 0.02 * line["value"]
    assert line["reducer"]["active"] and line["reducer"]["messages"] >= 5 + 2
    for k in ("cpu_baseline", "bf16_line", "forward_test", "roofline_wgrad"):
        assert k not in line
    # the message schedule of a step, as the reducer recorded it on the real model: the token-id gather
    # and the packed head message first (they travel under the whole encoder backward), one message per encoder layer in
    # backward order issued from inside the backward, then what is not a layer slice, the compact text rows last
    sched = line["reducer"]["schedule"]
    order = sched["order"]
    assert sorted(order[:2]) == ["head", "ids"], order[:4]
    # (round 6: the twelve layers travel as five groups -- 4, 4, 2, 1, 1 layers, top of the encoder first -- each sent when its
    # LOWEST layer's backward returns: `GradReducer._group_of`)
    assert order[2:7] == ["layer:11-8", "layer:7-4", "layer:3-2", "layer:1", "layer:0"], order
    assert order[-1] == "text_rows" and set(order[7:-1]) == {"rest"}, order[7:]
    D, F_ = 768, 3072
    # a layer's message = the four Linears of both experts (56.7 MB); its LayerNorm parameters (three of width D, one of width F,
    # weight + bias, both experts) sit behind the layers in the arena and travel in the closing message
    per_expert = (3 * D * D + 3 * D) + (D * D + D) + (F_ * D + F_) + (D * F_ + D)
    ln_per_layer = 2 * (3 * 2 * D + 2 * F_)
    mb = sched["messages_bytes"]
    assert mb["layer"] == {"messages": 5, "bytes": 12 * 2 * per_expert * 4}, mb["layer"]
    assert mb["text_rows"]["bytes"] == 2 * 4 * 20 * D * 4 and mb["ids"]["bytes"] == 2 * 4 * 20 * 8   # world x B x T rows / ids
    # patch embedding (D x 3 x 32 x 32 + D), cls token, both position tables (401 + 2 and 1024 rows), final LayerNorm (2 x 2 D);
    # the 64 010-row text table travels as the rows above, the mask token has no gradient
    rest = (D * 3 * 32 * 32 + D) + D + (403 + 1024) * D + 4 * D + 12 * ln_per_layer
    assert rest * 4 <= mb["rest"]["bytes"] <= rest * 4 + 65536, (mb["rest"], rest * 4)      # + the mask token's slot and 256-B alignment gaps
    assert line["reducer"]["bytes"] == sum(v["bytes"] for v in mb.values())
    assert line["reducer"]["exposed"]["steps"] == 3 and line["reducer"]["exposed"]["mean_ms"] >= 0.0
