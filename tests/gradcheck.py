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


def check_all_grads(fx, params, ntol, stol, tag="", excuse=None):
    """-> (violations, worst norm error, worst sampled-entry error).  Norms relative; sampled entries relative to the
    tensor's LARGEST entry (16 entries of a 2 M-entry matrix carry no usable direction cosine).
    excuse(key, flat indices of the entries beyond stol) -> bool mask: entries the caller can show to sit on a kink of the
    piecewise-linear network (a ReLU whose pre-activation is below the forward's rounding noise: `relu_gate_flips`)."""
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
        dev = (got - ga["vals"][i]).abs() / max(amax, 1e-20)
        over = dev > stol
        if excuse is not None and bool(over.any()):
            idx = ga["idx"][i].long()
            ok = excuse(k, idx[over])
            if bool(ok.any()):
                print(f"[gradients {tag}] {k}: {int(ok.sum())} sampled entr(y/ies) on a ReLU kink (gate differs from the exact engine's), "
                      f"deviation {float(dev[over][ok].max()):.3e} of the largest entry: excused")
                dev = dev.clone()
                dev[over.nonzero().reshape(-1)[ok]] = 0.0
        es = float(dev.max())
        worst_n, worst_s = max(worst_n, en), max(worst_s, es)
        if en > ntol or es > stol:
            bad.append((k, round(en, 5), round(es, 5)))
    print(f"[gradients {tag}] all {len(ga['keys'])} parameters ({n_noise} of them rounding noise in the reference): worst norm "
          f"error {worst_n:.3e}, worst sampled-entry error {worst_s:.3e} of the tensor's largest entry")
    return bad, worst_n, worst_s


def relu_gate_flips(model, run_training_forward):
    """Hidden units of the head's FFNs whose ReLU gate differs between the 16-bit engine and the exact-fp32 engine on this input:
    {ffn call index (TGQG layers, then decoder layers): set of hidden units}.  A pre-activation smaller than the rounding noise the
    encoder leaves on the FFN's input (1e-3) may land on either side of zero; the gradients of that unit's bias and weight row then
    differ by the unit's whole contribution -- a property of the piecewise-linear function, not of the kernels (the exact engine
    reproduces the reference's gates and gradients to 5e-5, test_exact_fp32_training_step_matches_reference_gradients)."""
    from simvg_amd import hip_ops as ops
    rec, mode = {}, ["lowp"]
    orig = ops.dec_ffn_fwd

    def spy(*a, **k):
        out = orig(*a, **k)
        rec.setdefault(mode[0], []).append(out["h1d"] > 0)
        return out
    ops.dec_ffn_fwd = spy
    try:
        prev = model.vis_enc.precision
        for m in ("lowp", "fp32"):
            mode[0] = m
            model.vis_enc.set_precision(m)
            run_training_forward()          # with gradients enabled: the forward the backward under test belongs to (single 16-bit weights)
        model.vis_enc.set_precision(prev)
    finally:
        ops.dec_ffn_fwd = orig
    return {i: set(((a != b).any(0)).nonzero().reshape(-1).tolist()) for i, (a, b) in enumerate(zip(rec["lowp"], rec["fp32"]))}


def ffn_unit_excuse(model, flips):
    """excuse callback of `check_all_grads`: a sampled entry of an FFN's first Linear (bias entry f / weight row f) or second Linear
    (weight column f) whose hidden unit f flipped its gate"""
    import re
    head = model.head

    def excuse(key, idx):
        m = re.match(r"head\.(text_guided_query_generation_transformer|transformer\.decoder)\.layers\.(\d+)\.ffns\.0\.layers\.(0\.0|1)\.(weight|bias)", key)
        if not m:
            return torch.zeros(len(idx), dtype=torch.bool)
        call = int(m.group(2)) + (0 if m.group(1).startswith("text") else head.num_tgqg_layers)
        units = flips.get(call, set())
        E = head.embed_dim
        Fd = head.tgqg_ffn if m.group(1).startswith("text") else head.dec_ffn
        if m.group(3) == "0.0":
            unit = idx if m.group(4) == "bias" else idx // E
        elif m.group(4) == "weight":
            unit = idx % Fd
        else:
            return torch.zeros(len(idx), dtype=torch.bool)
        return torch.tensor([int(u) in units for u in unit.tolist()], dtype=torch.bool)
    return excuse
