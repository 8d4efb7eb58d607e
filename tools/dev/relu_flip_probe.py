"""Is a sampled-entry deviation of a decoder FFN bias gradient a ReLU gate that differs between the 16-bit engine and the exact
engine?  Runs one fixture in both precision modes, records the saved FFN activations (h1d = relu(.) * m) of every decoder layer
and the FFN bias gradients, and prints gates that differ + the entries of db1 that differ most.
    python tools/dev/relu_flip_probe.py large_nq1"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "large_nq1"
    import test_model_gpu as T
    from simvg_amd import hip_ops as ops
    fx = torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)
    model, batch, cfg = T._build(fx)
    model.eval()
    db = T._dev_batch(batch)
    rec = {}
    orig = ops.dec_ffn_fwd

    def spy(*a, **k):
        out = orig(*a, **k)
        rec.setdefault(mode[0], []).append((out["h1d"].clone(), a[0].clone()))
        return out
    ops.dec_ffn_fwd = spy
    mode = ["lowp"]
    grads = {}
    for m in ("lowp", "fp32"):
        mode[0] = m
        model.vis_enc.set_precision(m)
        model.zero_grad(set_to_none=True)
        losses, _ = model(db["img"], db["ref_expr_inds"], db["img_metas"], return_loss=True, text_attention_mask=db["text_attention_mask"],
                          gt_bbox=batch["gt_bbox"], rescale=False)
        losses["loss_total"].backward()
        torch.cuda.synchronize()
        grads[m] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if "ffns.0.layers.0.0.bias" in k}
    ga = fx["grads_all"]
    for li, ((h16, t16), (h32, t32)) in enumerate(zip(rec["lowp"], rec["fp32"])):
        flips = ((h16 > 0) != (h32 > 0)).nonzero()
        print(f"ffn call {li}: rows {h16.shape[0]} hidden {h16.shape[1]}: gates that differ {flips.shape[0]}; t2 max abs diff {float((t16 - t32).abs().max()):.2e}")
        for r, f in flips.tolist()[:10]:
            print(f"    row {r} unit {f}: pre-activation-ish h 16-bit {float(h16[r, f]):.3e} fp32 {float(h32[r, f]):.3e}")
    for k in grads["lowp"]:
        d = (grads["lowp"][k] - grads["fp32"][k]).abs()
        i = ga["keys"].index(k)
        idx = ga["idx"][i].long()
        print(f"{k}: max |16-bit - fp32| {float(d.max()):.3e} at unit {int(d.argmax())}, largest entry {float(ga['amax'][i]):.3e}; "
              f"fixture samples units {idx.tolist()}: worst sampled diff {float(d.cpu()[idx].max()):.3e}")


if __name__ == "__main__":
    main()
