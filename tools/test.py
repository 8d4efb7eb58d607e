#!/usr/bin/env python
"""Drop-in for the reference's `tools/test.py` (:19-134): evaluate a checkpoint on every split of the config's
dataset (val [, testA, testB | test]; the `Mixed` pre-training set evaluates its three val_* splits), with the
EMA double pass (`apply_shadow` -> evaluate -> `restore`) when `cfg.ema` is set.

    python tools/test.py configs/x.py --load-from work_dir/.../det_best.pth [--cfg-options k=v ...]"""
import argparse
import os
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))

import torch                                                                         # noqa: E402
import torch.distributed as dist                                                     # noqa: E402

from simvg_amd.config import Config, DictAction                                      # noqa: E402
from simvg_amd.datasets import build_dataset, build_dataloader                       # noqa: E402
from simvg_amd.models import build_model                                             # noqa: E402
from simvg_amd.models.utils import ExponentialMovingAverage                          # noqa: E402
from simvg_amd.apis import evaluate_model, set_random_seed                           # noqa: E402
from simvg_amd.utils import (get_root_logger, load_checkpoint, init_dist, is_main,   # noqa: E402
                             load_pretrained_checkpoint, get_dist_info)
from train import apply_synthetic                                                    # noqa: E402  (tools/train.py)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="SimVG-test (MI355X)")
    parser.add_argument("config", help="test configuration file path.")
    parser.add_argument("--load-from", help="load from the saved .pth checkpoint, only used in validation.")
    parser.add_argument("--finetune-from", help="load from the pretrained checkpoint, only used in validation.")
    parser.add_argument("--launcher", choices=["none", "pytorch"], default="none")
    parser.add_argument("--cfg-options", nargs="+", action=DictAction, help="override settings of the config (key=value).")
    return parser.parse_args(argv)


def main_worker(cfg):
    cfg.distributed = False
    if cfg.launcher == "pytorch":
        cfg.distributed = True
        init_dist()
    cfg.rank, cfg.world_size = get_dist_info()
    if is_main():
        log_file = osp.join(osp.dirname(cfg.load_from), "test_log.txt") if cfg.load_from else None
        logger = get_root_logger(log_file=log_file)
        logger.info(cfg.pretty_text)
    apply_synthetic(cfg)
    if cfg.dataset == "Mixed":
        prefix = ["val_refcoco_unc", "val_refcocoplus_unc", "val_refcocog_umd"]
        datasets_cfgs = [cfg.data.train, cfg.data.val_refcoco_unc, cfg.data.val_refcocoplus_unc, cfg.data.val_refcocog_umd]
    else:
        prefix = ["val"]
        datasets_cfgs = [cfg.data.train, cfg.data.val]
        if hasattr(cfg.data, "testA") and hasattr(cfg.data, "testB"):
            datasets_cfgs += [cfg.data.testA, cfg.data.testB]
            prefix.extend(["testA", "testB"])
        elif hasattr(cfg.data, "test"):
            datasets_cfgs.append(cfg.data.test)
            prefix.extend(["test"])
    datasets = list(map(build_dataset, datasets_cfgs))
    dataloaders = list(map(lambda dataset: build_dataloader(cfg, dataset), datasets[1:]))

    device = torch.device("cuda", torch.cuda.current_device())
    model = build_model(cfg.model, word_emb=datasets[0].word_emb, num_token=datasets[0].num_token).to(device)
    model.vis_enc._ensure_engine(device)
    if cfg.use_fp16:
        raise NotImplementedError("use_fp16 (apex O1) is not part of this build")
    model_ema = ExponentialMovingAverage(model, cfg.ema_factor) if cfg.ema else None
    if cfg.load_from:
        load_checkpoint(model, model_ema, load_from=cfg.load_from)
    elif cfg.finetune_from:
        load_pretrained_checkpoint(model, None, cfg.finetune_from, amp=cfg.use_fp16)

    results = {}
    for eval_loader, _prefix in zip(dataloaders, prefix):
        if is_main():
            get_root_logger().info(f"SimVG - evaluating set {_prefix}")
        results[_prefix] = evaluate_model(-1, cfg, model, eval_loader)
        if cfg.ema:
            if is_main():
                get_root_logger().info(f"SimVG - evaluating set {_prefix} using ema")
            model_ema.apply_shadow()
            results[_prefix + "_ema"] = evaluate_model(-1, cfg, model, eval_loader)
            model_ema.restore()
    if cfg.distributed:
        dist.destroy_process_group()
    return results


def main(argv=None):
    args = parse_args(argv)
    cfg = Config.fromfile(args.config)
    if args.cfg_options is not None:
        cfg.merge_from_dict(args.cfg_options)
    cfg.load_from = args.load_from
    cfg.finetune_from = args.finetune_from
    cfg.launcher = args.launcher
    if cfg.seed is not None:
        set_random_seed(cfg.seed, deterministic=cfg.deterministic)
    return main_worker(cfg)


if __name__ == "__main__":
    main()
