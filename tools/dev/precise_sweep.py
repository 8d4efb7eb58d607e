"""GPU: which encoder Linears / how many layers need hi + lo 16-bit weights in forward_test for every box of a FULL batch to stay
within 1e-3 of the reference (fixtures base_nq1_full / large_nq10_grec_full), and what each choice costs.
    python tools/dev/precise_sweep.py [base|large]"""
import itertools
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from test_fullsize_gpu import _model, _batch, _boxes, _l1_stats, FULL_FIXTURE
    which = sys.argv[1:] or ["base", "large"]
    for vit, B, nq, grec in [("base", 64, 1, False), ("large", 32, 10, True)]:
        if vit not in which:
            continue
        fx = torch.load(os.path.join(ROOT, "tests", "golden", FULL_FIXTURE[(vit, B, nq)] + ".pt"), weights_only=False)
        ref = {"outputs_coord_decoder_branch": fx["dec_boxes"].float(), "outputs_coord_token_branch": fx["tok_boxes"].float()}
        model, cfg = _model(vit, nq)
        model.eval()
        enc = model.vis_enc
        b = _batch(cfg, B, grec)
        L = enc.L
        tags = ("wqkv", "wout", "w1", "w2")
        combos = [()] + [c for r in (1, 2, 3, 4) for c in itertools.combinations(tags, r)]
        for layers in (L, L // 2, L // 4):
            for c in combos:
                if not c and layers != L:
                    continue
                enc.precise_layers, enc.precise_which = (layers if c else 0), (c or tags)
                enc.precise_inference = enc.precise_layers > 0
                enc.wb2 = None if not c else enc.wb2
                enc.mark_weights_dirty()
                with torch.no_grad():
                    out = _boxes(model, b)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        _boxes(model, b)
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) / 3 * 1e3
                d = _l1_stats(out["outputs_coord_decoder_branch"].cpu(), ref["outputs_coord_decoder_branch"])
                t = _l1_stats(out["outputs_coord_token_branch"].cpu(), ref["outputs_coord_token_branch"])
                print(f"[{vit}] layers {layers:2d} which {'+'.join(c) or 'none':18s} {ms:7.2f} ms  decoder max {d[0]:.2e} mean {d[2]:.2e}  "
                      f"token max {t[0]:.2e} p99 {t[1]:.2e} mean {t[2]:.2e}", flush=True)


if __name__ == "__main__":
    main()
