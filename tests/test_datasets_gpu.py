"""GPU: the reference's annotation-file datasets end to end on the HIP path -- json record -> JPEG decode -> uint8 frame in
HBM -> device transforms (resize / normalise / pad kernels) -> DataLoader with aspect-ratio batches -> `tools/train.py`
and `tools/test.py` on the miniature dataset of tests/golden/loading_golden.pt (annotation records, JPEG bytes and the
sentencepiece model travel in the fixture)."""
import glob
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
S = 96
NORM = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375])


@pytest.fixture(scope="module")
def mini(tmp_path_factory):
    fx = torch.load(os.path.join(HERE, "golden", "loading_golden.pt"), weights_only=False)
    root = str(tmp_path_factory.mktemp("refdata"))
    for rel, data in fx["images"].items():
        os.makedirs(os.path.dirname(os.path.join(root, rel)), exist_ok=True)
        with open(os.path.join(root, rel), "wb") as f:
            f.write(data)
    for name, s in fx["sets"].items():
        os.makedirs(os.path.join(root, "anns", name), exist_ok=True)
        with open(os.path.join(root, "anns", name, "instances.json"), "w") as f:
            json.dump(s["anns"], f)
    with open(os.path.join(root, "beit3.spm"), "wb") as f:
        f.write(fx["spm"])
    return fx, root


def _pipeline(root, dataset, train, keys=("img", "ref_expr_inds", "gt_bbox", "text_attention_mask"), meta=None):
    steps = [dict(type="LoadImageAnnotationsFromFile", max_token=20, with_bbox=True, dataset=dataset, use_token_type="beit3",
                  spm_path=os.path.join(root, "beit3.spm"))]
    if train:
        steps.append(dict(type="LargeScaleJitter", out_max_size=S, jitter_min=0.3, jitter_max=1.4))
    steps += [dict(type="Resize", img_scale=(S, S), keep_ratio=False), dict(type="Normalize", **NORM), dict(type="Pad", size_divisor=32),
              dict(type="DefaultFormatBundle")]
    collect = dict(type="CollectData", keys=list(keys))
    if meta:
        collect["meta_keys"] = meta
    return steps + [collect]


def test_validation_item_equals_the_cpu_restatement_of_the_reference_pipeline(mini):
    from oracle import pipeline_cpu as P
    from simvg_amd.datasets import build_dataset
    fx, root = mini
    ds = build_dataset(dict(type="RefCOCOUNC", which_set="val", img_source=["coco"], imgsfile=os.path.join(root, "coco"),
                            annsfile=os.path.join(root, "anns", "RefCOCOUNC", "instances.json"),
                            pipeline=_pipeline(root, "RefCOCOUNC", False)))
    np.random.seed(3)
    item = ds[0]
    case = next(c for c in fx["cases"] if c["set"] == "RefCOCOUNC" and c["which_set"] == "val" and c["token_type"] == "beit3"
                and c["max_token"] == 20)
    frame = fx["decoded"][case["out"]["filename"]].numpy()
    h, w = frame.shape[:2]
    ref, wscale, hscale = P.imresize(frame, (S, S), return_scale=True, interpolation="bilinear", backend="cv2")
    ref = P.imnormalize(ref, np.array(NORM["mean"], dtype=np.float32), np.array(NORM["std"], dtype=np.float32), True)
    assert item["img"].is_cuda and item["img"].dtype == torch.float32 and tuple(item["img"].shape) == (3, S, S)
    assert torch.equal(item["img"].cpu(), torch.from_numpy(np.ascontiguousarray(ref.transpose(2, 0, 1))))       # bit-exact
    assert item["ref_expr_inds"].tolist() == np.asarray(case["out"]["ref_expr_inds"]).tolist()
    assert item["text_attention_mask"].tolist() == np.asarray(case["out"]["text_attention_mask"]).tolist()
    box = case["out"]["gt_bbox"].numpy() * np.array([wscale, hscale, wscale, hscale], dtype=np.float32)      # mmcv keeps the factor in fp32
    assert np.allclose(item["gt_bbox"].numpy(), box, rtol=0, atol=1e-9)
    m = item["img_metas"]
    assert m["ori_shape"] == (h, w, 3) and m["img_shape"] == (S, S, 3) and m["pad_shape"] == (S, S, 3)
    assert m["expression"] == case["out"]["expression"] and m["filename"].endswith(case["out"]["filename"])


def test_grefcoco_items_carry_box_lists_and_targets_through_the_loader(mini):
    from simvg_amd.datasets import build_dataset, build_dataloader, extract_data
    from simvg_amd.config import Config
    fx, root = mini
    meta = ["filename", "expression", "ori_shape", "img_shape", "pad_shape", "scale_factor", "target"]
    ds = build_dataset(dict(type="GRefCOCO", which_set="train", img_source=["coco"], imgsfile=os.path.join(root, "coco"),
                            annsfile=os.path.join(root, "anns", "GRefCOCO", "instances.json"),
                            pipeline=_pipeline(root, "GRefCOCO", False, meta=meta)))
    cfg = Config(dict(distributed=False, seed=1, data=dict(samples_per_gpu=2, workers_per_gpu=2)))      # decode in 2 worker processes
    np.random.seed(0)
    batches = [extract_data(b, torch.device("cuda")) for b in build_dataloader(cfg, ds)]
    assert len(batches) == 1 and all(b["img"].shape == (2, 3, S, S) and b["img"].is_cuda for b in batches)
    for b in batches:
        for gt, m in zip(b["gt_bbox"], b["img_metas"]):
            assert gt.dim() == 2 and gt.shape[1] == 4 and gt.shape[0] == len(m["target"])
            if m["target"][0]["category_id"] == -1:
                assert float(gt.abs().max()) == 0.0                       # the no-target record keeps its zero box


def test_train_and_test_tools_on_the_annotation_file_dataset(mini, tmp_path):
    import train as train_tool
    import test as test_tool
    fx, root = mini
    base = open(os.path.join(HERE, "cfg_fixture", "tiny_train.py")).read()
    base = base.replace('_base_ = ["./_base_/misc.py"]', f'_base_ = [{os.path.join(HERE, "cfg_fixture", "_base_", "misc.py")!r}]')
    head, tail = base.split("data = dict(", 1)
    tail = tail.split("model = dict(", 1)[1]
    common = (f'img_source=["coco"], imgsfile={os.path.join(root, "coco")!r}, '
              f'annsfile={os.path.join(root, "anns", "RefCOCOUNC", "instances.json")!r}')
    data = ("data = dict(samples_per_gpu=2, workers_per_gpu=2,\n"
            f"    train=dict(type='RefCOCOUNC', which_set='train', {common}, pipeline={_pipeline(root, 'RefCOCOUNC', True)!r}),\n"
            + "".join(f"    {s}=dict(type='RefCOCOUNC', which_set='{s}', {common}, pipeline={_pipeline(root, 'RefCOCOUNC', False)!r}),\n"
                      for s in ("val", "testA", "testB")) + ")\n")
    cfg_path = str(tmp_path / "mini_refcoco.py")
    with open(cfg_path, "w") as f:
        f.write(head + data + "model = dict(" + tail)
    work = str(tmp_path / "run")
    train_tool.main([cfg_path, "--work-dir", work, "--cfg-options", "log_interval=1"])
    run = glob.glob(os.path.join(work, "*"))[0]
    log = open(glob.glob(os.path.join(run, "*_train_log.txt"))[0]).read()
    assert "train-epoch[1]-[2/2]" in log and "train-epoch[2]-[2/2]" in log and "saved epoch 2 checkpoint" in log
    assert "nan" not in log.lower()
    res = test_tool.main([cfg_path, "--load-from", os.path.join(run, "latest.pth")])
    assert set(res) == {"val", "val_ema", "testA", "testA_ema", "testB", "testB_ema"}


def test_batched_device_stage_equals_the_per_frame_pipeline(mini, monkeypatch):
    """the two-stage loader's device stage against the same transforms run frame by frame on the device -- bit-exact, train
    pipeline (LargeScaleJitter incl. its crop / escape branches) and eval pipeline.  Two forms: the batch program compiled
    in the worker (launch descriptors with offsets, `run_batch_program`) and the fallback that replays the recorded
    operations of the DeferredFrames (`materialize_batch`)."""
    import random
    from simvg_amd.datasets import build_dataset, pipelines
    from simvg_amd.datasets.pipelines import materialize_batch, run_batch_program
    from simvg_amd.datasets.refsets import pack_host_batch
    fx, root = mini
    dev = torch.device("cuda")
    for train in (True, False):
        ds = build_dataset(dict(type="RefCOCOUNC", which_set="train", img_source=["coco"], imgsfile=os.path.join(root, "coco"),
                                annsfile=os.path.join(root, "anns", "RefCOCOUNC", "instances.json"),
                                pipeline=_pipeline(root, "RefCOCOUNC", train)))
        assert ds.host_steps() == len(ds.pipeline.transforms)
        order = [0, 1, 2, 1, 0, 2, 2, 0] * 5                    # 40 frames: more than one 32-job launch
        random.seed(4); np.random.seed(4)
        direct = [ds[i] for i in order]
        # (1) compiled program
        random.seed(4); np.random.seed(4)
        host = pack_host_batch([ds.host_item(i) for i in order])
        assert host["program"] is not None and host["items"] is None
        out = run_batch_program(host["program"], host["frames"].to(dev), dev)
        assert tuple(out.shape) == (len(order), 3, S, S)
        for k, a in enumerate(direct):
            assert torch.equal(a["img"], out[k])
            assert np.array_equal(a["gt_bbox"].numpy(), host["fields"]["gt_bbox"][k])
            assert np.array_equal(a["ref_expr_inds"].numpy(), host["fields"]["ref_expr_inds"][k])
            assert a["img_metas"]["img_shape"] == host["metas"][k]["img_shape"]
        # (2) fallback: recorded operations replayed
        monkeypatch.setattr(pipelines, "compile_batch_program", lambda frames: None)
        random.seed(4); np.random.seed(4)
        host = pack_host_batch([ds.host_item(i) for i in order])
        assert host["program"] is None
        imgs, stacked = materialize_batch([it["img"] for it in host["items"]], host["frames"].to(dev), dev)
        assert stacked is not None
        for a, b in zip(direct, imgs):
            assert torch.equal(a["img"], b)
        monkeypatch.undo()
