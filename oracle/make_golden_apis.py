"""Golden vectors for the outer-loop pieces of SURVEY.md section 8 rows f-1 / f-4, produced by EXECUTING the
reference's own code (dev container only; /root/reference never travels):

  * `MultiStepLRWarmUp` / cosine schedulers (simvg/core/scheduler.py): per-epoch learning rates of a 3-group Adam;
  * `accuracy` (Det@0.5) and `grec_evaluate_f1_nacc` (simvg/apis/test.py) on seeded synthetic boxes;
  * `ExponentialMovingAverage` (simvg/models/utils.py): shadow trajectory of a small module over 14 updates,
    apply_shadow / restore.

    python -m oracle.make_golden_apis        -> tests/golden/apis_golden.pt

TEST INFRASTRUCTURE ONLY.  Third-party leaves under the reference functions (mmcv Registry, mmdet bbox_overlaps,
torchvision box_area) are the restatements in oracle/leaf.py -- parity unpinned at that boundary."""
import importlib.util
import logging
import os
import re
import sys
import warnings

import torch

from . import ref_loader

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "apis_golden.pt")

SCHED_CASES = [
    ("multistep_w3_d25", dict(type="MultiStepLRWarmUp", warmup_epochs=3, decay_steps=[25], decay_ratio=0.1, max_epoch=30)),
    ("multistep_w0_d2_4", dict(type="MultiStepLRWarmUp", warmup_epochs=0, decay_steps=[2, 4], decay_ratio=0.5, max_epoch=6)),
    ("multistep_linear", dict(type="MultiStepLRWarmUp", warmup_epochs=2, decay_steps=None, decay_ratio=None, max_epoch=10)),
    ("cosine", dict(type="CosineAnnealingLR", T_max=8, max_epoch=8, eta_min=1e-6)),
    ("cosine_restarts", dict(type="CosineAnnealingLRWarmRestarts", T_0=3, T_mult=2, max_epoch=9, eta_min=0.0)),
]


def sched_inputs():
    ps = [torch.nn.Parameter(torch.zeros(2)) for _ in range(3)]
    return [{"params": [ps[0]], "lr": 5e-5}, {"params": [ps[1]], "lr": 5e-4}, {"params": [ps[2]], "lr": 5e-4}]


def boxes_case(seed, B, degenerate=False):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(B, 2, generator=g) * 400
    wh = 20 + torch.rand(B, 2, generator=g) * 200
    gt = torch.cat([xy, xy + wh], 1)
    jitter = (torch.rand(B, 4, generator=g) - 0.5) * wh.repeat(1, 2) * 0.9
    pred = gt + jitter
    pred = torch.cat([torch.min(pred[:, :2], pred[:, 2:]), torch.max(pred[:, :2], pred[:, 2:])], 1)
    if degenerate:
        pred[0] = gt[0]                       # exact hit
        pred[1] = torch.tensor([5.0, 5.0, 5.0, 5.0])   # zero-area prediction
        gt[2] = torch.tensor([7.0, 7.0, 7.0, 7.0]); pred[2] = gt[2]   # zero-area both: union clamps to eps
    return gt, pred


def grec_case(seed, B, nq=10, max_t=3):
    g = torch.Generator().manual_seed(seed)
    preds, gts, targets = [], [], []
    for b in range(B):
        k = int(torch.randint(0, max_t + 1, (1,), generator=g))
        n = max(k, 1)
        xy = torch.rand(n, 2, generator=g) * 300
        wh = 30 + torch.rand(n, 2, generator=g) * 150
        gt = torch.cat([xy, xy + wh], 1) if k > 0 else torch.zeros(1, 4)
        scores = torch.rand(nq, generator=g)
        boxes = torch.rand(nq, 4, generator=g) * 300
        boxes = torch.cat([torch.min(boxes[:, :2], boxes[:, 2:]), torch.max(boxes[:, :2], boxes[:, 2:]) + 1.0], 1)
        for j in range(min(k, nq)):            # plant good detections with high scores for some targets
            if float(torch.rand(1, generator=g)) < 0.7:
                boxes[j] = gt[j] + (torch.rand(4, generator=g) - 0.5) * 8
                scores[j] = 0.75 + 0.25 * float(torch.rand(1, generator=g))
        if k == 0 and b % 2 == 0:
            scores = scores * 0.6              # correctly predicts "no target"
        preds.append({"scores": scores, "boxes": boxes})
        gts.append(gt)
        targets.append([dict(category_id=-1 if k == 0 else 1) for _ in range(n)])
    return preds, gts, targets


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.a = torch.nn.Linear(5, 4)
        self.b = torch.nn.LayerNorm(4)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g))
        self.register_buffer("w", torch.tensor([1.0, 0.1]))
        self.register_buffer("count", torch.tensor(0, dtype=torch.int64))


def ema_updates(model, ema, steps, seed=11):
    g = torch.Generator().manual_seed(seed)
    traj = []
    for s in range(steps):
        with torch.no_grad():
            for p in model.parameters():
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
            model.w.mul_(1.01)
            model.count.add_(3)
        ema.update_params()
        if s in (0, 1, 8, steps - 1):
            traj.append({k: v.clone() for k, v in ema.shadow.items()})
    return traj


class _Capture(logging.Handler):
    def __init__(self):
        super().__init__()
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


def strip_times(line):
    line = re.sub(r"time: ?[0-9.]+, ", "time:T, ", line)
    return re.sub(r"data_time: ?[0-9.]+, ", "data_time:T, ", line)


def loop_golden(R):
    """Drive the REFERENCE's train_model / evaluate_model (simvg/apis/train.py, test.py) with the mock model."""
    from . import mock_loop as ML
    spec = importlib.util.spec_from_file_location("simvg.apis.train", os.path.join(ref_loader.REF_ROOT, "simvg/apis/train.py"))
    sys.modules["simvg.apis"].test = R["test"]
    sys.modules["simvg.apis.test"] = R["test"]
    tr = importlib.util.module_from_spec(spec)
    sys.modules["simvg.apis.train"] = tr
    spec.loader.exec_module(tr)
    cap = _Capture()
    lg = logging.getLogger("SimVG-ref")
    lg.setLevel(logging.INFO)
    lg.addHandler(cap)
    res = {}
    for dataset in ("RefCOCOUNC", "GRefCOCO"):
        grec = dataset == "GRefCOCO"
        cfg = ML.make_cfg(dataset)
        model = ML.MockVG(grec=grec)
        ema = R["utils"].ExponentialMovingAverage(model, 0.999)
        groups = [{"params": [p for n, p in model.named_parameters() if "vis_enc" in n], "lr": 5e-3},
                  {"params": [p for n, p in model.named_parameters() if "vis_enc" not in n], "lr": 5e-2}]
        opt = R["optimizer"].build_optimizer(dict(type="Adam", lr=5e-2, betas=(0.9, 0.98), eps=1e-9, weight_decay=0, amsgrad=True), groups)
        cap.lines.clear()
        for epoch in range(2):
            tr.train_model(epoch, cfg, model, ema, opt, ML.Loader(ML.batches(5, 4, 100 + epoch, grec, wrap=True)))
        d_acc, miou = R["test"].evaluate_model(1, cfg, model, ML.Loader(ML.batches(3, 4, 200, grec, wrap=True)))
        res[dataset] = dict(lines=[strip_times(l) for l in cap.lines], d_acc=float(d_acc), miou=float(miou),
                            params={k: v.detach().clone() for k, v in model.state_dict().items()},
                            shadow={k: v.clone() for k, v in ema.shadow.items()})
    lg.removeHandler(cap)
    return res


def main():
    warnings.filterwarnings("ignore")
    R = ref_loader.load_apis()
    out = {"sched": {}, "acc": [], "grec": [], "ema": {}}
    for name, cfg in SCHED_CASES:
        opt = torch.optim.Adam(sched_inputs(), lr=5e-4)
        sch = R["scheduler"].build_scheduler(dict(cfg), opt)
        lrs = [[g["lr"] for g in opt.param_groups]]
        for _ in range(cfg["max_epoch"]):
            opt.step()
            sch.step()
            lrs.append([g["lr"] for g in opt.param_groups])
        out["sched"][name] = dict(cfg=cfg, lrs=lrs)
    for seed, B, deg in [(1, 16, False), (2, 64, False), (3, 8, True)]:
        gt, pred = boxes_case(seed, B, deg)
        det, miou, macc = R["test"].accuracy(pred, [g for g in gt], None, None, device="cpu")
        out["acc"].append(dict(seed=seed, B=B, degenerate=deg, gt=gt, pred=pred, det_acc=float(det), mask_iou=miou,
                               mask_acc=macc))
    for seed, B in [(5, 12), (6, 40), (7, 3)]:
        preds, gts, targets = grec_case(seed, B)
        f1, nacc = R["test"].grec_evaluate_f1_nacc(preds, gts, targets, device="cpu")
        out["grec"].append(dict(seed=seed, B=B, preds=preds, gts=gts, targets=targets, f1=float(f1), n_acc=float(nacc)))
    for buffer_ema in (True, False):
        m = _Toy()
        ema = R["utils"].ExponentialMovingAverage(m, 0.999, buffer_ema=buffer_ema)
        traj = ema_updates(m, ema, 14)
        live = {k: v.clone() for k, v in m.state_dict().items()}
        ema.apply_shadow()
        applied = {k: v.clone() for k, v in m.state_dict().items()}
        ema.restore()
        restored = {k: v.clone() for k, v in m.state_dict().items()}
        assert all(torch.equal(live[k], restored[k]) for k in live)
        out["ema"][f"buffer_ema_{buffer_ema}"] = dict(traj=traj, applied=applied, step=ema.step)
    out["loop"] = loop_golden(R)
    torch.save(out, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    for k, v in out["sched"].items():
        print(k, [round(x[1], 7) for x in v["lrs"][:6]], "...")
    print("acc", [(a["B"], round(a["det_acc"], 3)) for a in out["acc"]])
    print("grec", [(a["B"], round(a["f1"], 3), round(a["n_acc"], 3)) for a in out["grec"]])


if __name__ == "__main__":
    main()
