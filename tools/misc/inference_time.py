#!/usr/bin/env python
"""Single-sample inference latency with the command line and protocol of the reference's `tools/misc/inference_time.py`
(:21-79): the validation split at samples_per_gpu = 1, the first batch, 10 warm-up calls of `model(**inputs,
return_loss=False, with_bbox=True)`, then `--test_samples_number` timed calls -> "inference_time = X ms/iter", followed
by the model's MACs and parameter count.

    python tools/misc/inference_time.py --config configs/x.py [--checkpoint work_dir/.../latest.pth]
                                        [--test_samples_number 2000] [--cfg-options data.synthetic=True ...]

Differences: the clock stops after a device synchronisation (the reference reads the host clock with work still queued);
MACs come from this build's analytic GEMM count of the forward pass (the reference asks `thop`, which is not a dependency
here) -- the same convention as BASELINE.md section 2: every GEMM-like op, elementwise / softmax / LayerNorm excluded."""
import argparse
import os.path as osp
import sys
import time

import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))

from simvg_amd.config import Config, DictAction                                      # noqa: E402
from simvg_amd.datasets import extract_data                                          # noqa: E402
from simvg_amd.runtime import Session                                                # noqa: E402
from simvg_amd.utils import load_checkpoint                                          # noqa: E402


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Inference Time (MI355X)")
    parser.add_argument("--config", required=True, type=str, help="configuration file of the run")
    parser.add_argument("--checkpoint", default=None, type=str, help="saved .pth checkpoint (random weights without it)")
    parser.add_argument("--test_samples_number", default=2000, type=int, help="number of timed forward_test calls")
    parser.add_argument("--cfg-options", nargs="+", action=DictAction, help="override settings of the config (key=value).")
    return parser.parse_args(argv)


def forward_macs(model, n_text):
    """multiply-accumulates of one forward_test on one sample: patch embedding, the encoder's Linears and attention
    products on (1 + patches + text) tokens, the head's projections, decoder layers and MLPs"""
    enc, head = model.vis_enc, model.head
    D, L, Fd = enc.D, enc.L, enc.F
    P = enc.np
    N = P + 1 + n_text
    macs = P * D * 3 * enc.patch_size ** 2                                   # patch embedding (im2col GEMM)
    macs += L * N * (3 * D * D + D * D + 2 * D * Fd)                         # qkv, out, fc1, fc2
    macs += L * 2 * N * N * D                                                # Q K^T and P V over all heads
    E, nq, C = head.embed_dim, head.num_queries, head.in_channels
    macs += (P + 1) * C * E + n_text * C * E + C * E                          # input_proj, input_text_proj, input_cls_proj

    def layer(Lk, kv_rows, ffn):
        return nq * (3 * E * E + E * E) + 2 * nq * nq * E + nq * E * E + kv_rows * 2 * E * E + 2 * nq * Lk * E + nq * E * E \
            + 2 * nq * E * ffn
    macs += head.num_tgqg_layers * layer(n_text, n_text, head.tgqg_ffn) + head.num_decoder_layers * layer(P, P + 1, head.dec_ffn)
    macs += nq * (E * E + 2 * (2 * E * E + E * 4 + E * 2))                   # token MLP, the two box MLPs and class heads
    return macs


def main(argv=None):
    args = parse_args(argv)
    cfg = Config.fromfile(args.config)
    if args.cfg_options is not None:
        cfg.merge_from_dict(args.cfg_options)
    cfg.launcher = "none"
    cfg.data.samples_per_gpu = 1
    run = Session(cfg)
    val_set = run.dataset("val")
    loader = run.loader(val_set)
    run.build(val_set)
    if args.checkpoint:
        load_checkpoint(run.model, run.ema, load_from=args.checkpoint)
    model = run.model.eval()
    inputs = next(iter(loader))
    inputs.pop("gt_bbox", None)
    inputs = extract_data(inputs, run.device)
    inputs.update(return_loss=False, with_bbox=True, rescale=False)
    with torch.no_grad():
        for _ in range(10):                                                    # warm-up (tools/misc/inference_time.py:68-69)
            model(**inputs)
        torch.cuda.synchronize()
        since = time.time()
        for _ in range(args.test_samples_number):
            model(**inputs)
        torch.cuda.synchronize()
    ms = (time.time() - since) / args.test_samples_number * 1000
    print("inference_time = {}ms/iter".format(ms))
    macs = forward_macs(model, int(inputs["ref_expr_inds"].shape[1]))
    params = sum(p.numel() for p in model.parameters())
    print("total_macs:{:.3f}G, total_params:{:.3f}M".format(macs / 1e9, params / 1e6))
    return ms, macs, params


if __name__ == "__main__":
    main()
