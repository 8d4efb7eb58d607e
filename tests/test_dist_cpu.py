"""CPU (gloo, world_size 2): the data-parallel exchange of simvg_amd.dist.GradReducer -- per-layer slices of the
flat gradient arena + head gradients -- averages to exactly what a single process sees on the concatenated batch.
The encoder engine itself needs a GPU, so a stand-in object exposes the same arena / hook surface."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeEnc(torch.nn.Module):
    """Same surface as BEIT3 for the reducer: _arena, L, layer_param_names(i), _grad_ready_hook."""

    def __init__(self, L=3, D=8):
        super().__init__()
        self.L = L
        self.beit3 = torch.nn.Module()
        self.beit3.emb = torch.nn.Parameter(torch.zeros(5, D))
        self.beit3.text_embed = torch.nn.Embedding(50, D)          # exchanged sparsely: only rows of this step's tokens
        self.beit3.encoder = torch.nn.Module()
        self.beit3.encoder.layers = torch.nn.ModuleList([torch.nn.Linear(D, D) for _ in range(L)])
        self.beit3.encoder.layer_norm = torch.nn.LayerNorm(D)
        from simvg_amd.arena import ParamArena
        self._arena = ParamArena(dict(self.named_parameters()), [], "cpu")
        self._grad_ready_hook = None
        self._last_ids = None

    def layer_param_names(self, i):
        return [n for n in self._arena.params if n.startswith(f"beit3.encoder.layers.{i}.")]


class _FakeModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.vis_enc = _FakeEnc()
        self.head = torch.nn.Linear(8, 2)


def _rank_ids(rank, g, uneven):
    """token ids of one rank's 2 x 4 batch.  `uneven`: ranks touch very different SETS of rows -- rank 0 one row only
    (all positions the same token), rank 1 two rows shared with rank 2, rank 3 eight distinct rows -- so duplicates inside
    a rank, overlaps between ranks and rows nobody touches all occur."""
    if not uneven:
        return torch.randint(0, 50, (2, 4), generator=g)
    return [torch.full((2, 4), 7), torch.tensor([[3, 9, 3, 9], [9, 9, 3, 3]]), torch.tensor([[9, 3, 11, 12], [13, 3, 9, 7]]),
            torch.arange(20, 28).reshape(2, 4)][rank % 4]


def _worker(rank, world, port, out, uneven=False, message=None, head_arena=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SIMVG_DIST_CHECK="1")   # + the same-id-list assertion
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from simvg_amd.dist import GradReducer
    torch.manual_seed(0)
    model = _FakeModel()
    red = GradReducer(model, message_dtype=message)
    ra = None
    if head_arena:      # what FlatAdam registers: the head's parameters own a flat gradient buffer, all-reduced in place
        from simvg_amd.core.optimizer import _RestArena
        ra = _RestArena(list(model.head.parameters()))
        model._simvg_rest_arenas = [ra]
    early_bias = head_arena == "early"       # "early": every head gradient exists when the head message leaves
    A = model.vis_enc._arena
    g = torch.Generator().manual_seed(100 + rank)
    red.begin()
    A.begin_backward()
    A.flat_grad.copy_(torch.randn(A.total, generator=g))          # this rank's local gradients
    # text table: rows of THIS rank's tokens carry gradient, every other row is exactly zero (as embed_bwd leaves it)
    ids = _rank_ids(rank, g, uneven)
    model.vis_enc._last_ids = ids
    tg = A.grad("beit3.text_embed.weight")
    keep = torch.zeros(50, dtype=torch.bool)
    keep[ids.reshape(-1)] = True
    tg[~keep] = 0.0
    model.head.weight.grad = torch.randn(2, 8, generator=g)
    late_bias = torch.randn(2, generator=g)
    if early_bias:
        model.head.bias.grad = late_bias
    local = (A.flat_grad.clone(), model.head.weight.grad.clone(), late_bias.clone())
    for i in reversed(range(model.vis_enc.L)):                     # what BEIT3._engine_backward does
        model.vis_enc._grad_ready_hook(i)
        if i == model.vis_enc.L - 1 and not early_bias:
            model.head.bias.grad = late_bias      # a head gradient that lands AFTER the early head message was packed
    model.vis_enc._grad_ready_hook(-1)
    red.finish()
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    exp_flat = sum(x[0] for x in gathered) / world
    exp_w = sum(x[1] for x in gathered) / world
    exp_b = sum(x[2] for x in gathered) / world
    # fp32 messages: the mean up to summation order; bf16 messages: every addend and the result rounded to 8 bits
    tol = dict(atol=1e-6) if message is None else dict(atol=4e-2, rtol=2e-2)
    ok = (torch.allclose(A.flat_grad, exp_flat, **tol) and torch.allclose(model.head.weight.grad, exp_w, **tol)
          and torch.allclose(model.head.bias.grad, exp_b, **tol)
          and all(torch.equal(p.grad, A.grad(n)) for n, p in A.params.items()))
    # rows of the text table no rank touched stay exactly zero (they are never exchanged)
    touched = torch.zeros(50, dtype=torch.bool)
    all_ids = [None] * world
    dist.all_gather_object(all_ids, ids)
    for t in all_ids:
        touched[t.reshape(-1)] = True
    ok = ok and bool((A.grad("beit3.text_embed.weight")[~touched] == 0).all())
    flats = [None] * world
    dist.all_gather_object(flats, A.flat_grad.clone())
    ok = ok and all(torch.equal(flats[0], f) for f in flats)       # replicas end bit-identical
    # the message schedule: ids gathered and the packed head message first, then one message per layer
    # in backward order, then everything that is not a layer slice, the compact text rows last, the late head gradient at
    # finish(); every message's bytes = what it covers (fp32, or 2 bytes per entry for bf16 messages)
    A_ = model.vis_enc._arena
    esz = 4 if message is None else 2
    kinds = [k for k, _ in red.last_schedule]
    L = model.vis_enc.L
    tail = ["text_rows"] if early_bias else ["text_rows", "late"]
    sched_ok = (kinds[:2] in (["head", "ids"], ["ids", "head"]) and kinds[2:2 + L] == [f"layer:{i}" for i in reversed(range(L))]
                and kinds[-len(tail):] == tail and set(kinds[2 + L:-len(tail)]) == {"rest"})
    by = dict()
    for k, b in red.last_schedule:
        by[k] = by.get(k, 0) + b
    layer_bytes = [esz * (A_.slice_of(model.vis_enc.layer_param_names(i))[1] - A_.slice_of(model.vis_enc.layer_param_names(i))[0]) for i in range(L)]
    sched_ok = sched_ok and all(by[f"layer:{i}"] == layer_bytes[i] for i in range(L))
    if ra is None:
        sched_ok = sched_ok and by["head"] == esz * model.head.weight.numel() and by["late"] == esz * model.head.bias.numel()
    else:
        # the flat buffer travels whole (alignment gaps included); every gradient IS its slice of it afterwards, and the
        # optimizer's own gather later in the step has nothing left to copy
        sched_ok = sched_ok and by["head"] == esz * ra.total and (early_bias or by["late"] == esz * model.head.bias.numel())
        sched_ok = sched_ok and all(p.grad is v for p, v in zip(ra.params, ra.grad_views))
        # FlatAdam's own gather (the one that keeps the has-a-gradient bookkeeping) comes after finish(): it sees the step's
        # FINAL set -- also when a gradient arrived after the reducer's early gather ("late") -- and has nothing to copy
        before, ver = ra.flat_grad.clone(), ra.flat_grad._version
        sched_ok = sched_ok and ra.gather_grads() == 2 and ra.flat_grad._version == ver and torch.equal(before, ra.flat_grad)
        sched_ok = sched_ok and ra._graded == (True, True)
    sched_ok = sched_ok and by["text_rows"] == esz * world * 8 * 8 and by["ids"] == 8 * world * 8       # int64 ids, D = 8
    # dense remainder = the arena minus the layer slices minus the (sparsely exchanged) text table
    sched_ok = sched_ok and by["rest"] == esz * (A_.total - sum(layer_bytes) // esz - A_.params["beit3.text_embed.weight"].numel())
    sched_ok = sched_ok and red.last_stats["bytes"] == sum(b for _, b in red.last_schedule)

    out[rank] = bool(ok) and red.last_sparse_rows == world * 8 and bool(sched_ok)     # the text table went as `world` x 8 token rows
    dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_grad_reducer_world4_uneven_token_sets():
    world = 4
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), out, True), nprocs=world, join=True)
    assert dict(out) == {r: True for r in range(world)}


def test_grad_reducer_head_message_is_the_optimizers_flat_gradient_buffer():
    world = 2
    for mode, message in (("early", None), ("late", None), ("early", "bf16")):
        out = mp.Manager().dict()
        mp.spawn(_worker, args=(world, _free_port(), out, False, message, mode), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}, (mode, message)


def _accumulate_after_send_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from simvg_amd.dist import GradReducer
    from simvg_amd.core.optimizer import _RestArena
    model = _FakeModel()
    model.vis_enc = None
    ra = _RestArena(list(model.head.parameters()))
    model._simvg_rest_arenas = [ra]
    red = GradReducer(model)
    red.begin()
    for p in model.head.parameters():
        p.grad = torch.ones_like(p)
    red._launch_head()
    model.head.weight.grad.add_(1.0)          # autograd accumulating into a slice that is already travelling
    try:
        red.finish()
        out[rank] = "no error"
    except RuntimeError as e:
        out[rank] = "changed after the head message had left" in str(e)
    dist.destroy_process_group()


def test_grad_reducer_refuses_a_gradient_accumulated_into_a_message_in_flight():
    out = mp.Manager().dict()
    mp.spawn(_accumulate_after_send_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert dict(out) == {0: True, 1: True}


def test_grad_reducer_bf16_message_keeps_fp32_master():
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), out, True, "bf16"), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


# ---------------------------------------------------------------------------------------------------------------------
# the training loop itself under data parallelism: world-2 gloo run of `train_model` == single-process run on the
# concatenated batches (gradient averaging by GradReducer, reduce_mean'd statistics, sampler.set_epoch)
# ---------------------------------------------------------------------------------------------------------------------
def _loop_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import logging
    from oracle import mock_loop as ML
    from simvg_amd.apis import train_model
    from simvg_amd.core import build_optimizer
    from simvg_amd.utils import get_root_logger
    cfg = ML.make_cfg("RefCOCOUNC")
    cfg.distributed, cfg.ema = True, False
    model = ML.MockVG()
    opt = build_optimizer(dict(type="Adam", lr=5e-2, betas=(0.9, 0.98), eps=1e-9, weight_decay=0, amsgrad=True),
                          [{"params": list(model.parameters()), "lr": 5e-2}], model=model)
    lines = []

    class Cap(logging.Handler):
        def emit(self, record):
            lines.append(record.getMessage())
    lg = get_root_logger()
    lg.addHandler(Cap())
    full = ML.batches(4, 8, 300)                                   # 4 global batches of 8 pairs
    shard = [{k: (v[rank * 4:(rank + 1) * 4] if not isinstance(v, list) else v[rank * 4:(rank + 1) * 4]) for k, v in b.items()}
             for b in full]
    means = train_model(0, cfg, model, None, opt, ML.Loader(shard))
    out[rank] = dict(params={k: v.detach().clone() for k, v in model.state_dict().items()}, means=means, lines=lines)
    dist.destroy_process_group()


def test_train_model_world2_equals_single_process_on_the_global_batch():
    from oracle import mock_loop as ML
    from simvg_amd.apis import train_model
    from simvg_amd.core import build_optimizer
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_loop_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    cfg = ML.make_cfg("RefCOCOUNC")
    cfg.ema = False
    model = ML.MockVG()
    opt = build_optimizer(dict(type="Adam", lr=5e-2, betas=(0.9, 0.98), eps=1e-9, weight_decay=0, amsgrad=True),
                          [{"params": list(model.parameters()), "lr": 5e-2}], model=model)
    means = train_model(0, cfg, model, None, opt, ML.Loader(ML.batches(4, 8, 300)))
    r0, r1 = out[0], out[1]
    for k, v in model.state_dict().items():
        assert torch.allclose(r0["params"][k], r1["params"][k], atol=0, rtol=0), k       # replicas stay identical
        assert torch.allclose(r0["params"][k], v, rtol=2e-5, atol=2e-6), k               # == global-batch training
    for k in means:                                                                       # reduce_mean'd statistics
        assert abs(r0["means"][k] - means[k]) <= 1e-5 * max(1.0, abs(means[k])), k
    assert any(line.startswith("train-epoch[1]-[4/4]") for line in r0["lines"]) and not r1["lines"]   # rank 0 logs only


# ---------------------------------------------------------------------------------------------------------------------
# loss normalisers: num_boxes = clamp(sum_ranks(k) / world, 1) (criterion.py:245-249) is all-reduced BEFORE the head
# ---------------------------------------------------------------------------------------------------------------------
def _nums_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from simvg_amd.models import build_head
    head = build_head(dict(type="TextGuidedQuerySelectKDDETRHead", num_queries=2, text_max_token=20, in_channels=128, embed_dim=256,
                           decoder_freeze=False, num_classes=1, aux_loss=True, num_encoder_layers=6, num_decoder_layers=3,
                           only_decoder=True, text_embed_aug=False,
                           branch_loss_weight={"decoder": 1.0, "balanced_distill": {"token": 2.0, "distill": 1.0}},
                           distill_type="hard_weighted", prepare_target_mode="score_iou_weighted", share_predicthead=False,
                           num_token_mlp_layers=1, mlp_aux_loss=False, text_guided_query_generation=True, num_tgqg_layers=2))
    metas = [dict(img_shape=(64, 64, 3), target=[dict(category_id=1)] * 3), dict(img_shape=(64, 64, 3), target=[dict(category_id=-1)])]
    if rank == 0:      # 3 targets + a no-target image: k = 3, matched pseudo targets = min(nq = 2, 3) + 0 = 2
        gt = [torch.tensor([[1.0, 2, 30, 40], [5, 5, 20, 20], [8, 9, 50, 60]]), torch.zeros(1, 4)]
    else:              # 1 + 1 targets: k = 2, pseudo = 2
        metas = [dict(img_shape=(64, 64, 3), target=[dict(category_id=1)]), dict(img_shape=(64, 64, 3), target=[dict(category_id=1)])]
        gt = [torch.tensor([[1.0, 2, 30, 40]]), torch.tensor([[3.0, 3, 9, 9]])]
    # the host side of prepare_targets (which entries are kept, the two normalisers and their all-reduce); the packing itself is a
    # device launch (tests/test_head_kernels_gpu.py::test_pack_targets_matches_the_reference_formula)
    rows, counts = head._target_rows(gt, metas, torch.device("cpu"))
    assert len(rows) == sum(counts) and all(r[0] is None and len(r[2]) == 4 for r in rows)
    nums = head.target_normalisers(counts, torch.device("cpu"))
    out[rank] = (nums.tolist(), counts)
    dist.destroy_process_group()


def test_loss_normalisers_are_averaged_over_ranks_before_the_head():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_nums_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0][0] == out[1][0] == [2.5, 2.0]            # (3 + 2) / 2 GT boxes, (2 + 2) / 2 matched pseudo targets
    assert out[0][1] == [3, 0] and out[1][1] == [1, 1]


def test_bench_gpus_n_refuses_a_node_with_fewer_gpus_and_a_mismatched_world():
    """`python bench.py --gpus N` starts its own ranks (bench.py::self_launch, the counterpart of the reference's
    tools/dist_train.sh:8-10).  On a box with fewer than N GPUs (this container: none) it must exit non-zero with a message
    instead of printing an `n_gpus: 1` line under the name of an N-GPU job; under a launcher whose WORLD_SIZE is not N it must
    refuse as well."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SIMVG_BENCH_SHARE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(have + 1 if have else 2), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "refusing" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=root, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip(), (r.returncode, r.stderr[-500:])


def test_rest_arena_without_any_gradient_zeroes_its_buffer():
    """a step on which NO parameter of a rest arena has a gradient must not re-apply the previous step's values"""
    from simvg_amd.core.optimizer import _RestArena
    lin = torch.nn.Linear(4, 3)
    ra = _RestArena(list(lin.parameters()))
    lin.weight.grad, lin.bias.grad = torch.ones(3, 4), torch.ones(3)
    assert ra.gather_grads() == 2 and float(ra.flat_grad.abs().sum()) == 15.0
    lin.weight.grad = lin.bias.grad = None
    os.environ["SIMVG_ALLOW_GRADED_SET_CHANGE"] = "1"
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert ra.gather_grads() == 0
    finally:
        del os.environ["SIMVG_ALLOW_GRADED_SET_CHANGE"]
    assert float(ra.flat_grad.abs().sum()) == 0.0
