"""Shared by the GPU tests: the model's gradients against the fixture's `grads_all` record (EVERY parameter the reference's
backward reaches: gradient norm, largest entry, 16 evenly spaced entries -- oracle/make_golden.py::_all_grads) and against the
two whole-module norms `grad_norm_vis_enc` / `grad_norm_head`."""
import torch

# A handful of gradients are ZERO in exact arithmetic and rounding noise in fp32 (the reference records norms of 1e-8 ... 2e-6
# beside module norms of 1e2): the self-attention in-projections of the FIRST layer of both query decoders (their value
# input is the all-zero target, so every key carries the same value row and the attention weights cannot change the output:
# transformer.py:134-176 with `tgt = zeros`).  Below NOISE_FLOOR x (module norm) a gradient is compared as "also noise".
NOISE_FLOOR = 1e-6


def module_norm(params, prefix):
    return float(torch.sqrt(sum((p.grad.detach().double() ** 2).sum() for k, p in params.items()
                                if k.startswith(prefix) and p.grad is not None)))


def check_all_grads(fx, params, ntol, stol, tag=""):
    """-> (violations, worst norm error, worst sampled-entry error).  Norms relative; sampled entries relative to the
    tensor's LARGEST entry (16 entries of a 2 M-entry matrix carry no usable direction cosine)."""
    bad = []
    for pre, key in (("vis_enc.", "grad_norm_vis_enc"), ("head.", "grad_norm_head")):
        got, ref = module_norm(params, pre), fx[key]
        e = abs(got - ref) / max(ref, 1e-12)
        print(f"[gradients {tag}] {key}: {got:.6g} vs reference {ref:.6g} (rel {e:.2e})")
        if e > ntol:
            bad.append((key, got, ref))
    ga = fx.get("grads_all")
    assert ga is not None, "fixture without grads_all: regenerate with oracle/make_golden.py"
    floor = NOISE_FLOOR * max(fx["grad_norm_vis_enc"], fx["grad_norm_head"])
    worst_n = worst_s = 0.0
    n_noise = 0
    for i, k in enumerate(ga["keys"]):
        g = params[k].grad
        if g is None:
            bad.append((k, "no gradient"))
            continue
        g = g.detach().float()
        ref_n, amax = float(ga["norm"][i]), float(ga["amax"][i])
        got_n = float(g.double().norm())
        if ref_n <= floor:                       # zero in exact arithmetic: ours has to be noise as well
            n_noise += 1
            if got_n > 100 * floor:
                bad.append((k, "reference gradient is rounding noise", ref_n, got_n))
            continue
        en = abs(got_n - ref_n) / ref_n
        got = g.reshape(-1).cpu()[ga["idx"][i].long()]
        es = float((got - ga["vals"][i]).abs().max()) / max(amax, 1e-20)
        worst_n, worst_s = max(worst_n, en), max(worst_s, es)
        if en > ntol or es > stol:
            bad.append((k, round(en, 5), round(es, 5)))
    print(f"[gradients {tag}] all {len(ga['keys'])} parameters ({n_noise} of them rounding noise in the reference): worst norm "
          f"error {worst_n:.3e}, worst sampled-entry error {worst_s:.3e} of the tensor's largest entry")
    return bad, worst_n, worst_s
