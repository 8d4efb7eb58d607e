"""Race / uninitialised-read hunt: the same forward_test call (and the same training step) repeated many times with allocator churn
in between must give the same bits every time.
    python tools/dev/determinism_stress.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import bench
    from simvg_amd.models import build_model
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = build_model(bench.model_cfg(1, "base")).to(dev)
    for B in (2, 8, 64):
        bb = bench.synthetic_batch(B, 11, dev)
        kw = dict(return_loss=False, text_attention_mask=bb["text_attention_mask"], with_bbox=True, with_mask=False, rescale=False)
        model.eval()
        ref, bad = None, 0
        junk = []
        with torch.no_grad():
            for i in range(reps):
                model(bb["img"], bb["ref_expr_inds"], bb["img_metas"], **kw)
                out = model._last_output
                cur = torch.cat([out[k].detach().float().reshape(-1) for k in ("outputs_coord_decoder_branch", "outputs_coord_token_branch",
                                                                                 "outputs_class_decoder_branch", "outputs_class_token_branch")]).clone()
                if ref is None:
                    ref = cur
                elif not torch.equal(ref, cur):
                    bad += 1
                    print(f"  forward_test B={B} rep {i}: differs, max abs {float((ref - cur).abs().max()):.3e}")
                junk.append(torch.full((1 + (i * 7919) % 4096, 257), float("nan"), device=dev))     # allocator churn with poisoned memory
                if len(junk) > 5:
                    junk.pop(0)
        print(f"forward_test B={B}: {bad} of {reps - 1} repetitions differ")
    model.eval()
    bb = bench.synthetic_batch(16, 5, dev)
    ref, bad = None, 0
    for i in range(max(8, reps // 4)):
        model.zero_grad(set_to_none=True)
        losses, _ = model(bb["img"], bb["ref_expr_inds"], bb["img_metas"], return_loss=True, text_attention_mask=bb["text_attention_mask"],
                          gt_bbox=bb["gt_bbox"], rescale=False)
        losses["loss_total"].backward()
        cur = torch.cat([p.grad.detach().float().reshape(-1) for n, p in model.named_parameters() if p.grad is not None and not n.startswith("vis_enc.")] +
                        [model.vis_enc._arena.flat_grad.detach().reshape(-1)[::97]]).clone()
        if ref is None:
            ref = cur
        elif not torch.equal(ref, cur):
            bad += 1
            d = (ref - cur).abs()
            print(f"  training step rep {i}: gradients differ at {int((d > 0).sum())} entries, max abs {float(d.max()):.3e}")
        junk = [torch.full((1 + (i * 104729) % 8192, 129), float("nan"), device=dev)]
    print(f"training step B=16 (eval-mode dropout): {bad} repetitions differ")


if __name__ == "__main__":
    main()
