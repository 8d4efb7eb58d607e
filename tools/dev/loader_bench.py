"""Throughput of the annotation-file input pipeline: N synthetic 640x480 JPEGs + one annotation json in a temp dir ->
two-stage loader (decode / tokenise in worker processes, LargeScaleJitter + Resize + Normalize + Pad on the GPU)."""
import io, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from PIL import Image
import sentencepiece as spm
from simvg_amd.config import Config
from simvg_amd.datasets import build_dataset, build_dataloader, extract_data

N, B = int(os.environ.get("N", 512)), int(os.environ.get("B", 64))
root = tempfile.mkdtemp()
os.makedirs(os.path.join(root, "coco"))
rng = np.random.RandomState(0)
yy, xx = np.mgrid[0:480, 0:640]
words = "the man in red shirt left dog on grass woman holding umbrella near bus second giraffe from right".split()
records = []
for i in range(N):
    img = np.stack([(xx * (1 + i % 3) + yy * 2 + 50 * c + i) % 256 for c in range(3)], -1) + rng.randint(-12, 13, size=(480, 640, 3))
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(root, "coco", "COCO_train2014_%012d.jpg" % i), quality=90)
    records.append(dict(image_id=i, height=480, width=640, expressions=[" ".join(rng.choice(words, size=rng.randint(3, 9)))],
                        bbox=[float(rng.randint(0, 300)), float(rng.randint(0, 200)), float(rng.randint(60, 300)), float(rng.randint(60, 250))]))
json.dump(dict(train=records, val=records[:64]), open(os.path.join(root, "instances.json"), "w"))
open(os.path.join(root, "c.txt"), "w").write("\n".join([" ".join(words)] * 50))
spm.SentencePieceTrainer.train(input=os.path.join(root, "c.txt"), model_prefix=os.path.join(root, "t"), vocab_size=60, hard_vocab_limit=False,
                               bos_id=-1, eos_id=-1, unk_id=0, pad_id=-1, minloglevel=2)
pipe = [dict(type="LoadImageAnnotationsFromFile", max_token=20, with_bbox=True, dataset="RefCOCOUNC", use_token_type="beit3", spm_path=os.path.join(root, "t.model")),
        dict(type="LargeScaleJitter", out_max_size=640, jitter_min=0.3, jitter_max=1.4),
        dict(type="Resize", img_scale=(640, 640), keep_ratio=False),
        dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375]), dict(type="Pad", size_divisor=32),
        dict(type="DefaultFormatBundle"), dict(type="CollectData", keys=["img", "ref_expr_inds", "gt_bbox", "text_attention_mask"])]
ds = build_dataset(dict(type="RefCOCOUNC", which_set="train", img_source=["coco"], imgsfile=os.path.join(root, "coco"),
                        annsfile=os.path.join(root, "instances.json"), pipeline=pipe))
print("host cores", os.cpu_count(), " frames", N)
for workers in [int(w) for w in os.environ.get("WORKERS", "0,8,16,32").split(",")]:
    cfg = Config(dict(distributed=False, seed=1, data=dict(samples_per_gpu=B, workers_per_gpu=workers)))
    loader = build_dataloader(cfg, ds)
    for epoch in range(2):                      # epoch 0 starts the workers and warms the page cache
        loader.sampler.set_epoch(epoch)
        torch.cuda.synchronize()
        t0, n = time.perf_counter(), 0
        for batch in loader:
            b = extract_data(batch, torch.device("cuda"))
            n += b["img"].shape[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"workers {workers:3d}: {n / dt:8.1f} images/s  ({dt / (n / B) * 1e3:6.1f} ms per batch of {B})", flush=True)
    del loader
if os.environ.get("PROFILE"):
    import cProfile, pstats
    cfg = Config(dict(distributed=False, seed=1, data=dict(samples_per_gpu=B, workers_per_gpu=16)))
    loader = build_dataloader(cfg, ds)
    for batch in loader:
        pass
    pr = cProfile.Profile()
    pr.enable()
    for batch in loader:
        b = extract_data(batch, torch.device("cuda"))
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
if os.environ.get("TRAIN"):
    if os.environ.get("SWITCH"):
        sys.setswitchinterval(float(os.environ["SWITCH"]))
    import bench
    from simvg_amd.models import build_model
    from simvg_amd.core import build_optimizer
    from simvg_amd.graphs import training_stream
    dev = torch.device("cuda", 0)
    model = build_model(bench.model_cfg()).to(dev).train()
    model.vis_enc._ensure_engine(dev)
    named = list(model.named_parameters())
    groups = [{"params": [p for n, p in named if "vis_enc" in n], "lr": 5e-5}, {"params": [], "lr": 5e-4},
              {"params": [p for n, p in named if "vis_enc" not in n], "lr": 5e-4}]
    opt = build_optimizer(dict(type="Adam", lr=5e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0, amsgrad=True), groups, model=model)
    for workers, background in ((16, True), (16, False)):
        cfg = Config(dict(distributed=False, seed=1, data=dict(samples_per_gpu=64, workers_per_gpu=workers)))
        loader = build_dataloader(cfg, ds)
        loader.background = background
        with training_stream(dev):
            for epoch in range(2):
                loader.sampler.set_epoch(epoch)
                torch.cuda.synchronize()
                t0, n = time.perf_counter(), 0
                prof = None
                if os.environ.get("PROFILE_TRAIN") and epoch == 1 and background:
                    import cProfile
                    prof = cProfile.Profile()
                    prof.enable()
                t_second = None
                for batch in loader:
                    if n == 64:                                    # steady state: from the second step on
                        torch.cuda.synchronize()
                        t_second = time.perf_counter()
                    b = extract_data(batch, dev)
                    losses, _ = model(b["img"], b["ref_expr_inds"], b["img_metas"], return_loss=True,
                                      text_attention_mask=b["text_attention_mask"], gt_bbox=b["gt_bbox"], rescale=False)
                    opt.zero_grad()
                    losses["loss_total"].backward()
                    opt.clip_grad_norm(0.15)
                    opt.step()
                    n += b["img"].shape[0]
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if prof is not None:
                    import pstats
                    prof.disable()
                    pstats.Stats(prof).sort_stats("tottime").print_stats(14)
        if hasattr(loader, "wait_s"):
            print(f"  consumer waited {loader.first_wait_s * 1e3:.1f} ms for the first batch of the epoch, then "
                  f"{loader.wait_s / max(n / 64 - 1, 1) * 1e3:.2f} ms per step for the loader thread")
        hg = getattr(model, "_head_graphs", None)
        if hg is not None:
            print("head graphs captured:", len(hg.graphs), "disabled:", hg.disabled, "default stream seen:", hg.default_stream_seen)
        steady = (n - 64) / (t0 + dt - t_second) if t_second is not None and n > 64 else float("nan")
        print(f"training from JPEG files, workers {workers}, background device stage {background}: {n / dt:7.1f} pairs/s over the "
              f"epoch incl. its pipeline fill ({dt / (n / 64) * 1e3:5.1f} ms per step), {steady:7.1f} pairs/s from the second step on",
              flush=True)
