#!/usr/bin/env python
"""Evaluation entry point with the command line of the reference's `tools/test.py` (:19-134): every split of the config's
dataset (val [, testA, testB | test]; the `Mixed` pre-training set reports its three val_* splits), each with the live
weights and -- when `cfg.ema` is set -- once more with the EMA shadow applied.

    python tools/test.py configs/x.py --load-from work_dir/.../det_best.pth [--cfg-options k=v ...]

-> {split: (d_acc, miou), split + "_ema": (...)}  (for GRefCOCO the pair is (F1, no-target accuracy))."""
import argparse
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))

from simvg_amd.apis import set_random_seed                                           # noqa: E402
from simvg_amd.config import Config, DictAction                                      # noqa: E402
from simvg_amd.runtime import Session                                                # noqa: E402
from simvg_amd.utils import get_root_logger, is_main, load_checkpoint, load_pretrained_checkpoint   # noqa: E402


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="SimVG-test (MI355X)")
    parser.add_argument("config", help="test configuration file path.")
    parser.add_argument("--load-from", help="load from the saved .pth checkpoint, only used in validation.")
    parser.add_argument("--finetune-from", help="load from the pretrained checkpoint, only used in validation.")
    parser.add_argument("--launcher", choices=["none", "pytorch"], default="none")
    parser.add_argument("--cfg-options", nargs="+", action=DictAction, help="override settings of the config (key=value).")
    return parser.parse_args(argv)


def report(cfg):
    run = Session(cfg)
    run.open_log(osp.join(osp.dirname(cfg.load_from), "test_log.txt") if cfg.load_from else None)
    train_set = run.dataset("train")          # only its vocabulary hand-over is used (word_emb / num_token)
    splits = run.test_splits()
    loaders = [run.loader(run.dataset(s)) for s in splits]
    run.build(train_set)
    if cfg.load_from:
        has_ema = load_checkpoint(run.model, run.ema, load_from=cfg.load_from)[3]
        if run.ema is not None and not has_ema:
            # no shadow in the file (the reference fails with a NameError here): evaluate the loaded weights in both passes
            # rather than a shadow of weights that were never loaded, and say so
            run.fresh_ema()
            if is_main():
                get_root_logger().info(f"{cfg.load_from} carries no ema_state_dict: the `_ema` results below are those of the plain weights")
    elif cfg.finetune_from:
        load_pretrained_checkpoint(run.model, None, cfg.finetune_from, amp=cfg.use_fp16)
    results = {}
    for split, loader in zip(splits, loaders):
        res = run.evaluate(-1, loader, f"SimVG - evaluating set {split}", f"SimVG - evaluating set {split} using ema")
        results.update({split + suffix: pair for suffix, pair in res.items()})
    run.close()
    return results


def main(argv=None):
    args = parse_args(argv)
    cfg = Config.fromfile(args.config)
    if args.cfg_options is not None:
        cfg.merge_from_dict(args.cfg_options)
    cfg.load_from, cfg.finetune_from, cfg.launcher = args.load_from, args.finetune_from, args.launcher
    if cfg.seed is not None:
        set_random_seed(cfg.seed, deterministic=cfg.deterministic)
    return report(cfg)


if __name__ == "__main__":
    main()
