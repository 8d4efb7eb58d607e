"""GPU: how many layers of hi + lo qkv weights the TRAINING forward needs for every box of a FULL batch to stay within 1e-3 of the
reference (fixtures base_nq1_full / large_nq10_grec_full), and what a training step pays for it.
    python tools/dev/precise_train_sweep.py [base|large] [which=wqkv]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from test_fullsize_gpu import _model, _batch, _l1_stats, FULL_FIXTURE
    args = [a for a in sys.argv[1:] if "=" not in a] or ["base", "large"]
    whichs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("which=")] or ["wqkv"]
    for vit, B, nq, grec in [("base", 64, 1, False), ("large", 32, 10, True)]:
        if vit not in args:
            continue
        fx = torch.load(os.path.join(ROOT, "tests", "golden", FULL_FIXTURE[(vit, B, nq)] + ".pt"), weights_only=False)
        ref = {"outputs_coord_decoder_branch": fx["dec_boxes"].float(), "outputs_coord_token_branch": fx["tok_boxes"].float()}
        L = 12 if vit == "base" else 24
        combos = [("wqkv", 0), ("patch", L), ("patch,wqkv", L // 2), ("patch,wqkv", L), ("patch,wqkv,w1", L // 2), ("patch,wqkv,w1", L)]
        if os.environ.get("ZERO_LO"):
            combos = [("patch,wqkv", L // 4), ("patch,wqkv", L // 2), ("patch,wqkv", L)]
        if whichs != ["wqkv"]:
            combos = [(w, l) for w in whichs for l in ((L // 2, L) if ":" not in w else (L,))]
        for which, layers in combos:
            if True:
                os.environ["SIMVG_PRECISE_TRAIN"] = str(layers)
                os.environ["SIMVG_PRECISE_TRAIN_WHICH"] = which
                model, cfg = _model(vit, nq)
                model.eval()          # dropout / DropPath off, as the fixture's reference run
                b = _batch(cfg, B, grec)

                def step():
                    model.zero_grad(set_to_none=True)
                    losses, _ = model(b["img"], b["ref_expr_inds"], b["img_metas"], return_loss=True,
                                      text_attention_mask=b["text_attention_mask"], gt_bbox=b["gt_bbox"], rescale=False)
                    losses["loss_total"].backward()
                    return losses
                step()
                zero = os.environ.get("ZERO_LO", "")          # emulate "no lo half" for the q / k / v rows of the qkv weights (eval mode:
                if zero:                                      # the [lo | hi] rows are rewritten only when the master weights move)
                    enc = model.vis_enc
                    D = enc.D
                    for tag, w in enc.wbs.items():
                        if tag.startswith("wqkv"):
                            for part in zero.split(","):
                                r0 = {"q": 0, "k": D, "v": 2 * D}[part]
                                w[:, r0:r0 + D, :D] = 0
                    step()
                out = {k: model._last_output[k].detach().float().cpu() for k in ref}
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 5 * 1e3
                d = _l1_stats(out["outputs_coord_decoder_branch"], ref["outputs_coord_decoder_branch"])
                t = _l1_stats(out["outputs_coord_token_branch"], ref["outputs_coord_token_branch"])
                print(f"[{vit}] train fwd: {which:24s} first {layers:2d} layers  fwd+bwd {ms:7.2f} ms  decoder max {d[0]:.2e} mean {d[2]:.2e}  "
                      f"token max {t[0]:.2e} p99 {t[1]:.2e} mean {t[2]:.2e}", flush=True)
                del model
                torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
