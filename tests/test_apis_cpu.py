"""CPU tests of the outer-loop mirror: config loader semantics, optimizer / scheduler
registries, metrics, EMA, train / evaluate loops -- against golden values produced by EXECUTING the reference's own
code (`oracle/make_golden_apis.py` -> `tests/golden/apis_golden.pt`)."""
import argparse
import logging
import os
import re

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = torch.load(os.path.join(HERE, "golden", "apis_golden.pt"), weights_only=False)


# ------------------------------------------------------------------------------------------------ config
def test_config_base_merge_delete_and_attribute_access():
    from simvg_amd.config import Config
    cfg = Config.fromfile(os.path.join(HERE, "cfg_fixture", "exp", "noema#finetune#x.py"))
    assert cfg.dataset == "RefCOCOUNC" and cfg["dataset"] == "RefCOCOUNC"
    assert cfg.data.samples_per_gpu == 16 and cfg.data.workers_per_gpu == 4           # dict merged key-wise
    assert cfg.data.train.type == "RefCOCOUNC" and cfg.data.train.annsfile == "./data/annotations/x.json"
    assert cfg.data.train.pipeline[0].max_token == 20                                   # lists are replaced, not merged
    assert cfg.data.train.pipeline[1].img_scale == (640, 640)
    assert cfg.data.val.to_dict() == dict(type="Other", which_set="val")               # _delete_=True replaces
    assert cfg.ema is False and cfg.ema_factor == 0.999 and cfg.seed == 6666            # child overrides base
    assert cfg.here == "misc"                                                           # {{ fileBasenameNoExtension }}
    assert cfg.optimizer_config.betas == (0.9, 0.98)
    assert cfg.model.head.branch_loss_weight == {"decoder": 1.0}
    assert getattr(cfg.data, "val_flickr30k", None) is None and not hasattr(cfg.data, "testA")
    assert cfg.get("work_dir", None) is None
    lr_v = cfg.optimizer_config.pop("lr_vis_enc")
    assert abs(lr_v - 5e-5) < 1e-12 and "lr_vis_enc" not in cfg.optimizer_config
    cfg.distributed = False
    cfg.extra = dict(a=dict(b=1))
    assert cfg.extra.a.b == 1
    with pytest.raises(AttributeError):
        cfg.data.nope


def test_config_duplicate_base_keys_rejected_and_roundtrip(tmp_path):
    from simvg_amd.config import Config
    with pytest.raises(KeyError):
        Config.fromfile(os.path.join(HERE, "cfg_fixture", "exp", "dup.py"))
    cfg = Config.fromfile(os.path.join(HERE, "cfg_fixture", "exp", "noema#finetune#x.py"))
    out = tmp_path / "dumped.py"
    cfg.dump(str(out))
    assert Config.fromfile(str(out)).to_dict() == cfg.to_dict()


def test_cfg_options_grammar_and_merge():
    from simvg_amd.config import Config, DictAction
    p = argparse.ArgumentParser()
    p.add_argument("--cfg-options", nargs="+", action=DictAction)
    a = p.parse_args(["--cfg-options", "data.samples_per_gpu=4", "model.head.num_queries=10", "ema=true", "lr=1e-4",
                      "x=[1,2]", "y=a,b", "z=[(1,2),(3,4)]", "data.train.pipeline.0.max_token=7", "n=None", "s=abc"])
    o = a.cfg_options
    assert o["data.samples_per_gpu"] == 4 and o["ema"] is True and o["lr"] == 1e-4 and o["x"] == [1, 2]
    assert o["y"] == ["a", "b"] and o["z"] == [(1, 2), (3, 4)] and o["n"] is None and o["s"] == "abc"
    cfg = Config.fromfile(os.path.join(HERE, "cfg_fixture", "exp", "noema#finetune#x.py"))
    cfg.merge_from_dict(o)
    assert cfg.data.samples_per_gpu == 4 and cfg.model.head.num_queries == 10 and cfg.ema is True
    assert cfg.data.train.pipeline[0].max_token == 7 and cfg.data.train.pipeline[0].type == "Load"
    assert cfg.data.train.pipeline[1].type == "Resize"


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference tree only exists in the dev container")
def test_every_reference_config_loads():
    import glob
    from simvg_amd.config import Config
    files = [f for f in glob.glob("/root/reference/configs/**/*.py", recursive=True) if "/_base_/" not in f]
    assert len(files) >= 50
    for f in files:
        cfg = Config.fromfile(f)
        assert cfg.model.type == "MIXDETRMB" and cfg.optimizer_config.type == "Adam" and cfg.grad_norm_clip == 0.15
        assert cfg.scheduler_config.type == "MultiStepLRWarmUp"


# ------------------------------------------------------------------------------------------------ optimizer / scheduler
def _groups():
    ps = [torch.nn.Parameter(torch.zeros(2)) for _ in range(3)]
    return [{"params": [ps[0]], "lr": 5e-5}, {"params": [ps[1]], "lr": 5e-4}, {"params": [ps[2]], "lr": 5e-4}]


@pytest.mark.parametrize("name", sorted(GOLD["sched"]))
def test_scheduler_matches_reference_per_epoch(name):
    from simvg_amd.core import build_optimizer, build_scheduler
    g = GOLD["sched"][name]
    opt = build_optimizer(dict(type="Adam", lr=5e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0, amsgrad=True), _groups())
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["amsgrad"] and opt.defaults["betas"] == (0.9, 0.98)
    sch = build_scheduler(dict(g["cfg"]), opt)
    lrs = [[x["lr"] for x in opt.param_groups]]
    for _ in range(g["cfg"]["max_epoch"]):
        opt.step()
        sch.step()
        lrs.append([x["lr"] for x in opt.param_groups])
    assert len(lrs) == len(g["lrs"])
    for a, b in zip(lrs, g["lrs"]):
        assert a == pytest.approx(b, rel=1e-12, abs=0)


def test_optimizer_registry_types():
    from simvg_amd.core import build_optimizer, OPTIMIZERS
    for typ, cls in [("SGD", torch.optim.SGD), ("RMSProp", torch.optim.RMSprop), ("AdamW", torch.optim.AdamW)]:
        assert isinstance(build_optimizer(dict(type=typ, lr=0.1), _groups()), cls)
    with pytest.raises(KeyError):
        OPTIMIZERS.build(dict(type="Lion", lr=0.1), default_args=dict(params=_groups()))


# ------------------------------------------------------------------------------------------------ metrics
@pytest.mark.parametrize("i", range(len(GOLD["acc"])))
def test_accuracy_matches_reference(i):
    from simvg_amd.apis import accuracy
    g = GOLD["acc"][i]
    det, miou, macc = accuracy(g["pred"], [b for b in g["gt"]], None, None, device="cpu")
    assert float(det) == pytest.approx(g["det_acc"], abs=1e-5)
    assert torch.equal(miou, g["mask_iou"]) and torch.equal(macc, g["mask_acc"])


@pytest.mark.parametrize("i", range(len(GOLD["grec"])))
def test_grec_f1_nacc_matches_reference(i):
    from simvg_amd.apis import grec_evaluate_f1_nacc
    g = GOLD["grec"][i]
    f1, nacc = grec_evaluate_f1_nacc(g["preds"], g["gts"], g["targets"], device="cpu")
    assert float(f1) == pytest.approx(g["f1"], abs=1e-4) and float(nacc) == pytest.approx(g["n_acc"], abs=1e-4)
    assert tuple(float(x) for x in grec_evaluate_f1_nacc(None, [], [], device="cpu")) == (0.0, 0.0)


# ------------------------------------------------------------------------------------------------ EMA
@pytest.mark.parametrize("buffer_ema", [True, False])
def test_ema_matches_reference_trajectory(buffer_ema):
    from oracle.make_golden_apis import _Toy, ema_updates
    from simvg_amd.models.utils import ExponentialMovingAverage
    g = GOLD["ema"][f"buffer_ema_{buffer_ema}"]
    m = _Toy()
    ema = ExponentialMovingAverage(m, 0.999, buffer_ema=buffer_ema)
    traj = ema_updates(m, ema, 14)
    assert ema.step == g["step"]
    for mine, ref in zip(traj, g["traj"]):
        assert mine.keys() == ref.keys()
        for k in ref:
            assert mine[k].dtype == ref[k].dtype
            assert torch.allclose(mine[k].double(), ref[k].double(), rtol=2e-6, atol=1e-7), k
    live = {k: v.clone() for k, v in m.state_dict().items()}
    ema.apply_shadow()
    for k, v in m.state_dict().items():
        assert torch.allclose(v.double(), g["applied"][k].double(), rtol=2e-6, atol=1e-7), k
    ema.restore()
    assert all(torch.equal(live[k], v) for k, v in m.state_dict().items())
    # checkpoint hand-off: load_checkpoint assigns `model_ema.shadow = ckpt["ema_state_dict"]`
    ema.shadow = {k: v.clone() + 1 for k, v in g["applied"].items()}
    assert all(torch.equal(ema.shadow[k], g["applied"][k] + 1) for k in g["applied"])


# ------------------------------------------------------------------------------------------------ loops
class _Capture(logging.Handler):
    def __init__(self):
        super().__init__()
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


@pytest.mark.parametrize("dataset", ["RefCOCOUNC", "GRefCOCO"])
def test_train_and_evaluate_loops_match_reference(dataset):
    """Same mock model, data and optimizer as the golden run of the REFERENCE's train_model / evaluate_model: the log
    lines (times stripped), the returned metrics, the trained parameters and the EMA shadow must agree."""
    from oracle import mock_loop as ML
    from oracle.make_golden_apis import strip_times
    from simvg_amd.apis import train_model, evaluate_model
    from simvg_amd.core import build_optimizer
    from simvg_amd.models.utils import ExponentialMovingAverage
    from simvg_amd.utils import get_root_logger
    g = GOLD["loop"][dataset]
    grec = dataset == "GRefCOCO"
    cfg = ML.make_cfg(dataset)
    model = ML.MockVG(grec=grec)
    ema = ExponentialMovingAverage(model, 0.999)
    groups = [{"params": [p for n, p in model.named_parameters() if "vis_enc" in n], "lr": 5e-3},
              {"params": [p for n, p in model.named_parameters() if "vis_enc" not in n], "lr": 5e-2}]
    opt = build_optimizer(dict(type="Adam", lr=5e-2, betas=(0.9, 0.98), eps=1e-9, weight_decay=0, amsgrad=True), groups, model=model)
    assert type(opt).__name__ == "Adam"          # no arena on this model -> the plain registry class
    cap = _Capture()
    lg = get_root_logger()
    lg.addHandler(cap)
    try:
        for epoch in range(2):
            train_model(epoch, cfg, model, ema, opt, ML.Loader(ML.batches(5, 4, 100 + epoch, grec)))
        d_acc, miou = evaluate_model(1, cfg, model, ML.Loader(ML.batches(3, 4, 200, grec)))
    finally:
        lg.removeHandler(cap)
    assert [strip_times(l) for l in cap.lines] == g["lines"]
    assert d_acc == pytest.approx(g["d_acc"], abs=1e-4) and miou == pytest.approx(g["miou"], abs=1e-4)
    for k, v in model.state_dict().items():
        assert torch.allclose(v, g["params"][k], rtol=1e-5, atol=1e-6), k
    for k, v in ema.shadow.items():
        assert torch.allclose(v.double(), g["shadow"][k].double(), rtol=1e-5, atol=1e-6), k


# ------------------------------------------------------------------------------------------------ data / checkpoints
def test_synthetic_dataset_and_loader_shapes():
    from simvg_amd.datasets import build_dataset, build_dataloader, extract_data
    from oracle.mock_loop import Cfg
    with pytest.raises(FileNotFoundError):          # without synthetic=True the annotation json is opened (tests/test_datasets_cpu.py)
        build_dataset(dict(type="RefCOCOUNC", which_set="train", imgsfile="x", annsfile="/nonexistent/instances.json", pipeline=[]))
    ds = build_dataset(dict(type="RefCOCOUNC", which_set="train", annsfile="x", pipeline=[], synthetic=True, length=10, img_size=64))
    assert len(ds) == 10 and ds.word_emb is None and ds.num_token == -1
    a, b = ds[3], ds[3]
    assert torch.equal(a["img"], b["img"]) and torch.equal(a["gt_bbox"], b["gt_bbox"])
    cfg = Cfg(distributed=False, seed=1, data=Cfg(samples_per_gpu=4), rank=0, world_size=1)
    batch = next(iter(build_dataloader(cfg, ds)))
    assert batch["img"].shape == (4, 3, 64, 64) and batch["gt_bbox"].shape == (4, 4) and len(batch["img_metas"]) == 4
    assert batch["ref_expr_inds"][:, 0].eq(0).all() and batch["text_attention_mask"].dtype == torch.int64
    moved = extract_data(batch, torch.device("cpu"))
    assert moved["img"].shape == (4, 3, 64, 64) and moved["img_metas"][0]["img_shape"] == (64, 64, 3)
    gds = build_dataset(dict(type="SyntheticRefDataset", dataset="GRefCOCO", max_targets=3, length=6, img_size=32))
    metas = [gds[i]["img_metas"]["target"] for i in range(6)]
    assert all(isinstance(t, list) and "category_id" in t[0] for t in metas)


def test_reference_signature_checkpoint_roundtrip(tmp_path):
    from oracle import mock_loop as ML
    from simvg_amd.core import build_optimizer, build_scheduler
    from simvg_amd.models.utils import ExponentialMovingAverage
    from simvg_amd.utils import save_checkpoint, load_checkpoint, load_pretrained_checkpoint
    m = ML.MockVG()
    ema = ExponentialMovingAverage(m, 0.999)
    opt = build_optimizer(dict(type="Adam", lr=1e-3, amsgrad=True), [{"params": list(m.parameters()), "lr": 1e-3}])
    sch = build_scheduler(dict(type="MultiStepLRWarmUp", warmup_epochs=1, decay_steps=[3], decay_ratio=0.1, max_epoch=4), opt)
    m(**{k: v for k, v in ML.batches(1, 2, 0)[0].items()})[0]["loss_total"].backward()
    opt.step(); sch.step(); ema.update_params()
    info = {"epoch": 2, "d_acc": 50.0, "miou": 0.0, "best_d_acc": 40.0, "best_miou": 0.0, "amp": False}
    save_checkpoint(str(tmp_path), 3, m, ema, opt, sch, info)
    assert sorted(os.listdir(tmp_path)) == ["det_best.pth", "epoch_3.pth", "latest.pth"]
    ck = torch.load(tmp_path / "latest.pth", weights_only=False)
    assert {"state_dict", "ema_state_dict", "optimizer", "scheduler", "lr", "epoch", "d_acc", "best_d_acc"} <= ck.keys()
    m2 = ML.MockVG()
    with torch.no_grad():
        for p in m2.parameters():
            p.zero_()
    ema2 = ExponentialMovingAverage(m2, 0.999)
    opt2 = build_optimizer(dict(type="Adam", lr=1e-3, amsgrad=True), [{"params": list(m2.parameters()), "lr": 1e-3}])
    sch2 = build_scheduler(dict(type="MultiStepLRWarmUp", warmup_epochs=1, decay_steps=[3], decay_ratio=0.1, max_epoch=4), opt2)
    start, best_d, best_m, flag = load_checkpoint(m2, ema2, resume_from=str(tmp_path / "latest.pth"), optimizer=opt2, scheduler=sch2)
    assert (start, best_d, best_m, flag) == (2, 40.0, 0.0, True)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert all(torch.equal(ema.shadow[k], ema2.shadow[k]) for k in ema.shadow)
    assert sch2.last_epoch == sch.last_epoch
    # load_from: epoch counter stays -1; `module.`-prefixed states (written by a DDP-wrapped reference run) are stripped
    ck["state_dict"] = {"module." + k: v for k, v in ck["state_dict"].items()}
    ck["ema_state_dict"] = {"module." + k: v for k, v in ck["ema_state_dict"].items()}
    torch.save(ck, tmp_path / "ddp.pth")
    m3 = ML.MockVG(); ema3 = ExponentialMovingAverage(m3, 0.999)
    start, _, _, flag = load_checkpoint(m3, ema3, load_from=str(tmp_path / "ddp.pth"))
    assert start == -1 and flag and all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m3.state_dict().values()))
    # a file without a shadow: the 4th value says so (the caller restarts its EMA from the loaded weights), weights still load
    ck2 = {k: v for k, v in ck.items() if k != "ema_state_dict"}
    torch.save(ck2, tmp_path / "noema.pth")
    m4 = ML.MockVG(); ema4 = ExponentialMovingAverage(m4, 0.999)
    before = {k: v.clone() for k, v in ema4.shadow.items()}
    assert load_checkpoint(m4, ema4, load_from=str(tmp_path / "noema.pth"))[3] is False
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m4.state_dict().values()))
    assert all(torch.equal(before[k], ema4.shadow[k]) for k in before)          # untouched: NOT the loaded weights
    with pytest.raises(AssertionError):
        load_pretrained_checkpoint(m3, ema3, str(tmp_path / "ddp.pth"))
    assert load_pretrained_checkpoint(ML.MockVG(), None, str(tmp_path / "ddp.pth")) == (-1, 40.0, 0.0)


def test_analytic_forward_macs_equal_the_baseline_table():
    """tools/misc/inference_time.py::forward_macs counts the GEMM-like work of one forward_test; 2 x MACs must be the
    algorithmic FLOPs per pair of BASELINE.md section 2 (80.44 GFLOP for ViT-B/32 @640, 274.78 for ViT-L/32)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tools", "misc"))
    import bench
    import inference_time
    from simvg_amd.models import build_model
    for vit, gflop in (("base", 80.44), ("large", 274.78)):
        model = build_model(bench.model_cfg(vit=vit))
        assert abs(2 * inference_time.forward_macs(model, 20) / 1e9 - gflop) < 0.01, vit
        assert abs(3 * 2 * inference_time.forward_macs(model, 20) - bench.FLOP_PER_PAIR_FWD_BWD[vit]) < 0.02e9, vit
