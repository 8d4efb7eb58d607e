#!/usr/bin/env python
"""Drop-in for the reference's `tools/train.py` (:26-216): same CLI, same config semantics, same epoch loop
(train_model -> evaluate (+ EMA double pass) -> save_checkpoint -> scheduler.step), on one MI355X per process.

    python tools/train.py configs/x.py [--work-dir D] [--resume-from F | --load-from F | --finetune-from F]
                          [--launcher none|pytorch] [--cfg-options k=v ...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py cfg.py --launcher pytorch

Differences from the reference, all forced by the hardware mapping and documented in INTEGRATION.md:
 * no (MM)DistributedDataParallel wrapper -- gradients are all-reduced from the flat arenas by `GradReducer` inside
   `train_model`, state_dict keys therefore never carry a `module.` prefix;
 * `type="Adam"` resolves to the fused flat-arena Adam (same arithmetic); `optimizer_config.flat=False` opts out;
 * `use_fp16` (apex O1) is not supported: the encoder's compute dtype is bf16-on-MFMA with fp32 master weights, or
   exact fp32 with `--cfg-options model.vis_enc.precision=fp32` for evaluation;
 * file-backed datasets are section 8 f-3 (not built): `--cfg-options data.synthetic=True` runs any reference
   config on synthetic RefCOCO-shaped pairs."""
import argparse
import os
import os.path as osp
import sys
import time

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))

import torch                                                                         # noqa: E402
import torch.distributed as dist                                                     # noqa: E402

from simvg_amd.config import Config, DictAction                                      # noqa: E402
from simvg_amd.core import build_optimizer, build_scheduler                          # noqa: E402
from simvg_amd.datasets import build_dataset, build_dataloader                       # noqa: E402
from simvg_amd.models import build_model                                             # noqa: E402
from simvg_amd.models.utils import ExponentialMovingAverage                          # noqa: E402
from simvg_amd.apis import set_random_seed, train_model, evaluate_model              # noqa: E402
from simvg_amd.utils import (get_root_logger, load_checkpoint, save_checkpoint,      # noqa: E402
                             load_pretrained_checkpoint, is_main, init_dist, get_dist_info)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="SimVG-train (MI355X)")
    parser.add_argument("config", help="training configuration file path.")
    parser.add_argument("--work-dir", help="directory of config file, training logs, and checkpoints.")
    parser.add_argument("--resume-from", help="resume training from the saved .pth checkpoint, only used in training.")
    parser.add_argument("--load-from", help="load weights (and EMA shadow) from the saved .pth checkpoint.")
    parser.add_argument("--finetune-from", help="finetune from the saved .pth checkpoint (non-strict, epoch reset).")
    parser.add_argument("--launcher", choices=["none", "pytorch"], default="none")
    parser.add_argument("--cfg-options", nargs="+", action=DictAction,
                        help="override settings of the config: key=value pairs, lists as key=a,b or key=\"[a,b]\", "
                             "nested tuples as key=\"[(a,b),(c,d)]\" (no white space).")
    return parser.parse_args(argv)


def split_param_groups(model, optimizer_config):
    """tools/train.py:78-94 of the reference: groups by parameter NAME; lr_vis_enc / lr_lan_enc are popped."""
    return [
        {"params": [p for n, p in model.named_parameters() if "vis_enc" in n and p.requires_grad],
         "lr": optimizer_config.pop("lr_vis_enc")},
        {"params": [p for n, p in model.named_parameters() if "lan_enc" in n and p.requires_grad],
         "lr": optimizer_config.pop("lr_lan_enc")},
        {"params": [p for n, p in model.named_parameters() if "lan_enc" not in n and "vis_enc" not in n and p.requires_grad],
         "lr": optimizer_config.lr},
    ]


def apply_synthetic(cfg):
    if cfg.data.get("synthetic", False):
        for k, v in cfg.data.items():
            if isinstance(v, dict) and "pipeline" in v:
                v["synthetic"] = True
                v.setdefault("type", cfg.dataset)
                v.setdefault("which_set", k)
                v.setdefault("img_size", cfg.get("img_size", 640))
                v.setdefault("max_token", cfg.get("max_token", 20))


def main_worker(cfg):
    cfg.distributed = False
    if cfg.launcher == "pytorch":
        cfg.distributed = True
        init_dist()
    cfg.rank, cfg.world_size = get_dist_info()
    logger = None
    if is_main():
        logger = get_root_logger(log_file=osp.join(cfg.work_dir, str(cfg.timestamp) + "_train_log.txt"))
        logger.info(cfg.pretty_text)
        cfg.dump(osp.join(cfg.work_dir, f"{cfg.timestamp}_" + osp.basename(cfg.config).replace("#", "_")))
    if cfg.use_fp16:
        raise NotImplementedError("use_fp16 (apex O1) is not part of this build: the encoder computes in bf16 on MFMA "
                                  "with fp32 master weights; every reference config sets use_fp16=False")
    apply_synthetic(cfg)
    datasets_cfgs = [cfg.data.train]
    if cfg.dataset == "Mixed":
        for item in ["val_refcoco_unc", "val_refcocoplus_unc", "val_refcocog_umd", "val_referitgame_berkeley", "val_flickr30k"]:
            if getattr(cfg.data, item, None):
                datasets_cfgs += [getattr(cfg.data, item)]
    else:
        datasets_cfgs += [cfg.data.val]
    datasets = list(map(build_dataset, datasets_cfgs))
    dataloaders = list(map(lambda dataset: build_dataloader(cfg, dataset), datasets))

    device = torch.device("cuda", torch.cuda.current_device())
    model = build_model(cfg.model, word_emb=datasets[0].word_emb, num_token=datasets[0].num_token)
    model = model.to(device)
    model.vis_enc._ensure_engine(device)          # lay the encoder out in its flat arenas before anyone takes views
    if cfg.distributed:                            # identical replicas (DDP broadcasts rank 0's parameters at wrap time)
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, 0)
    train_params = split_param_groups(model, cfg.optimizer_config)
    optimizer = build_optimizer(cfg.optimizer_config, train_params, model=model)
    scheduler = build_scheduler(cfg.scheduler_config, optimizer)

    model_ema = ExponentialMovingAverage(model, cfg.ema_factor) if cfg.ema else None
    start_epoch, best_d_acc, best_miou = -1, 0.0, 0.0
    if cfg.resume_from:
        start_epoch, _, _, _ = load_checkpoint(model, model_ema, cfg.resume_from, amp=cfg.use_fp16, optimizer=optimizer,
                                               scheduler=scheduler)
    elif cfg.finetune_from:
        load_pretrained_checkpoint(model, model_ema, cfg.finetune_from, amp=cfg.use_fp16)
    elif cfg.load_from:
        start_epoch, best_d_acc, best_miou, flag = load_checkpoint(model, model_ema, load_from=cfg.load_from)
        if not flag:
            model_ema = ExponentialMovingAverage(model, cfg.ema_factor) if cfg.ema else None

    begin_time = time.time()
    for epoch in range(start_epoch + 1, cfg.scheduler_config.max_epoch):
        start_time = time.time()
        train_model(epoch, cfg, model, model_ema, optimizer, dataloaders[0])
        t = int(time.time() - start_time)
        if is_main():
            logger.info("this_epoch_train_time={}m-{}s".format(t // 60, t % 60))
        if epoch % cfg.evaluate_interval == 0 and epoch >= cfg.start_evaluate_epoch:
            d_acc, miou = 0, 0
            for _loader in dataloaders[1:]:
                if is_main():
                    logger.info("Evaluating dataset: {}".format(_loader.dataset.which_set))
                set_d_acc, set_miou = evaluate_model(epoch, cfg, model, _loader)
                if cfg.ema:
                    if is_main():
                        logger.info("Evaluating dataset using ema: {}".format(_loader.dataset.which_set))
                    model_ema.apply_shadow()
                    set_d_acc, set_miou = evaluate_model(epoch, cfg, model, _loader)
                    model_ema.restore()
                d_acc += set_d_acc
                miou += set_miou
            d_acc /= len(dataloaders[1:])
            miou /= len(dataloaders[1:])
            if is_main():
                t = int(time.time() - start_time)
                logger.info("this_epoch_total_time={}m-{}s".format(t // 60, t % 60))
                t = int(time.time() - begin_time)
                logger.info("total_time={}m-{}s".format(t // 60, t % 60))
                saved_info = {"epoch": epoch, "d_acc": d_acc, "miou": miou, "best_d_acc": best_d_acc,
                              "best_miou": best_miou, "amp": cfg.use_fp16}
                save_checkpoint(cfg.work_dir, cfg.save_interval, model, model_ema, optimizer, scheduler, saved_info)
            best_d_acc = max(d_acc, best_d_acc)
            best_miou = max(miou, best_miou)
        scheduler.step()
        if cfg.distributed:
            dist.barrier()
    if cfg.distributed:
        dist.destroy_process_group()


def main(argv=None):
    args = parse_args(argv)
    cfg = Config.fromfile(args.config)
    cfg.timestamp = time.strftime("%Y%m%d_%H%M%S", time.localtime())
    if args.cfg_options is not None:
        cfg.merge_from_dict(args.cfg_options)
    if args.work_dir is not None:
        cfg.work_dir = args.work_dir
    elif cfg.get("work_dir", None) is None:
        cfg.work_dir = "./work_dir/" + args.config.split("configs/")[-1].split(".py")[0]
    cfg.work_dir = osp.join(cfg.work_dir, f"{cfg.timestamp}")
    if args.resume_from is not None:
        cfg.resume_from = args.resume_from
    if args.finetune_from is not None:
        cfg.finetune_from = args.finetune_from
    if args.load_from is not None:
        cfg.load_from = args.load_from
    cfg.launcher = args.launcher
    cfg.config = args.config
    os.makedirs(osp.abspath(cfg.work_dir), exist_ok=True)
    if cfg.seed is not None:
        set_random_seed(cfg.seed, deterministic=cfg.deterministic)
    main_worker(cfg)


if __name__ == "__main__":
    main()
