#!/usr/bin/env python
"""Training entry point with the command line of the reference's `tools/train.py` (:26-216):

    python tools/train.py configs/x.py [--work-dir D] [--resume-from F | --load-from F | --finetune-from F]
                          [--launcher none|pytorch] [--cfg-options k=v ...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py cfg.py --launcher pytorch

Per epoch: train_model -> (every cfg.evaluate_interval epochs from cfg.start_evaluate_epoch on) evaluate every validation
split, twice with cfg.ema (live and shadow weights; the shadow result counts) -> save_checkpoint -> scheduler.step().
Parameter groups are formed by NAME like the reference's (`vis_enc` at lr_vis_enc, `lan_enc` at lr_lan_enc, the rest at lr).

Deliberate differences, all consequences of the MI355X mapping:
 * no (MM)DistributedDataParallel wrapper: `GradReducer` all-reduces the flat gradient arenas inside `train_model`, so
   state_dict keys never carry a `module.` prefix;
 * `type="Adam"` resolves to the fused flat-arena Adam (same arithmetic); `optimizer_config.flat=False` opts out;
 * `use_fp16` (apex O1) is refused; `--cfg-options model.vis_enc.precision=fp32` selects the exact-fp32 parity mode;
 * `--cfg-options data.synthetic=True` runs any reference config on synthetic RefCOCO-shaped pairs when its `data/` tree
   is absent."""
import argparse
import os
import os.path as osp
import sys
import time

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))

import torch.distributed as dist                                                     # noqa: E402

from simvg_amd.apis import set_random_seed, train_model                              # noqa: E402
from simvg_amd.config import Config, DictAction                                      # noqa: E402
from simvg_amd.core import build_optimizer, build_scheduler                          # noqa: E402
from simvg_amd.runtime import Session, elapsed                                       # noqa: E402
from simvg_amd.utils import is_main, load_checkpoint, load_pretrained_checkpoint, save_checkpoint   # noqa: E402


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


def param_groups_by_name(model, optimizer_config):
    """three groups in the reference's order (tools/train.py:78-94); lr_vis_enc / lr_lan_enc leave the optimizer config"""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    rates = {"vis_enc": optimizer_config.pop("lr_vis_enc"), "lan_enc": optimizer_config.pop("lr_lan_enc")}
    groups = [{"params": [p for n, p in named if tag in n], "lr": lr} for tag, lr in rates.items()]
    groups.append({"params": [p for n, p in named if not any(tag in n for tag in rates)], "lr": optimizer_config.lr})
    return groups


class TrainingRun(Session):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.open_log(osp.join(cfg.work_dir, str(cfg.timestamp) + "_train_log.txt"))
        if is_main():
            cfg.dump(osp.join(cfg.work_dir, f"{cfg.timestamp}_" + osp.basename(cfg.config).replace("#", "_")))
        train_set = self.dataset("train")
        self.train_loader = self.loader(train_set)
        self.val_loaders = [self.loader(self.dataset(s)) for s in self.validation_splits()]
        self.build(train_set)
        self.optimizer = build_optimizer(cfg.optimizer_config, param_groups_by_name(self.model, cfg.optimizer_config),
                                         model=self.model)
        self.scheduler = build_scheduler(cfg.scheduler_config, self.optimizer)
        self.last_epoch, self.best = -1, [0.0, 0.0]            # best (d_acc, miou) seen so far
        self._restore()

    def _restore(self):
        cfg = self.cfg
        if cfg.resume_from:          # everything: weights, EMA, optimizer, scheduler, epoch counter
            self.last_epoch = load_checkpoint(self.model, self.ema, cfg.resume_from, amp=cfg.use_fp16,
                                              optimizer=self.optimizer, scheduler=self.scheduler)[0]
        elif cfg.finetune_from:      # weights only, non-strict
            load_pretrained_checkpoint(self.model, self.ema, cfg.finetune_from, amp=cfg.use_fp16)
        elif cfg.load_from:          # weights + EMA + the best metrics; a checkpoint without a shadow restarts the EMA
            self.last_epoch, self.best[0], self.best[1], has_ema = load_checkpoint(self.model, self.ema, load_from=cfg.load_from)
            if not has_ema:          # the file carried no shadow: the EMA starts from the weights just loaded
                self.fresh_ema()

    def validate(self, epoch):
        """mean over the validation splits of (d_acc, miou); with an EMA the shadow weights' numbers are the ones kept"""
        sums = [0.0, 0.0]
        for loader in self.val_loaders:
            name = loader.dataset.which_set
            res = self.evaluate(epoch, loader, "Evaluating dataset: {}".format(name), "Evaluating dataset using ema: {}".format(name))
            kept = res.get("_ema", res[""])
            sums = [s + v for s, v in zip(sums, kept)]
        return [s / len(self.val_loaders) for s in sums]

    def run(self):
        cfg = self.cfg
        began = time.time()
        for epoch in range(self.last_epoch + 1, cfg.scheduler_config.max_epoch):
            epoch_began = time.time()
            train_model(epoch, cfg, self.model, self.ema, self.optimizer, self.train_loader)
            self.say("this_epoch_train_time=" + elapsed(epoch_began))
            if epoch % cfg.evaluate_interval == 0 and epoch >= cfg.start_evaluate_epoch:
                d_acc, miou = self.validate(epoch)
                if is_main():
                    self.say("this_epoch_total_time=" + elapsed(epoch_began))
                    self.say("total_time=" + elapsed(began))
                    save_checkpoint(cfg.work_dir, cfg.save_interval, self.model, self.ema, self.optimizer, self.scheduler,
                                    {"epoch": epoch, "d_acc": d_acc, "miou": miou, "best_d_acc": self.best[0],
                                     "best_miou": self.best[1], "amp": cfg.use_fp16})
                self.best = [max(d_acc, self.best[0]), max(miou, self.best[1])]
            self.scheduler.step()
            if cfg.distributed:
                dist.barrier()
        self.close()


def configure(args):
    cfg = Config.fromfile(args.config)
    cfg.timestamp = time.strftime("%Y%m%d_%H%M%S", time.localtime())
    if args.cfg_options is not None:
        cfg.merge_from_dict(args.cfg_options)
    if args.work_dir is not None:
        cfg.work_dir = args.work_dir
    elif cfg.get("work_dir", None) is None:
        cfg.work_dir = "./work_dir/" + args.config.split("configs/")[-1].split(".py")[0]
    cfg.work_dir = osp.join(cfg.work_dir, f"{cfg.timestamp}")
    for key in ("resume_from", "finetune_from", "load_from"):
        if getattr(args, key) is not None:
            setattr(cfg, key, getattr(args, key))
    cfg.launcher, cfg.config = args.launcher, args.config
    return cfg


def main(argv=None):
    cfg = configure(parse_args(argv))
    os.makedirs(osp.abspath(cfg.work_dir), exist_ok=True)
    if cfg.seed is not None:
        set_random_seed(cfg.seed, deterministic=cfg.deterministic)
    TrainingRun(cfg).run()


if __name__ == "__main__":
    main()
