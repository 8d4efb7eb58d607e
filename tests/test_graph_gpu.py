"""hipGraph replay of the head (simvg_amd/graphs.py) must be the same computation as the eager head: losses, predictions
and every parameter gradient, on replay after replay with changing inputs and weights."""
import copy
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _no_dropout(model):
    model.head.attn_dropout = 0.0
    model.head.ffn_dropout = 0.0
    model.vis_enc.drop_path_probs = [0.0] * model.vis_enc.L


def test_graphed_head_equals_eager_over_several_steps():
    """ONE model, every step run twice on identical weights and inputs -- eager head, then (once the signature has
    repeated) the replayed hipGraphs -- so that the comparison is not polluted by two trajectories drifting apart
    through bf16 rounding flips; the weights move between steps so that a stale captured copy would show."""
    from test_tools_gpu import _tiny_model, _batch
    from simvg_amd.graphs import train_stream
    cfg, model = _tiny_model(3)
    _no_dropout(model)
    model.train()
    used_graph = []
    train_stream().wait_stream(torch.cuda.current_stream())     # model.to(device) was queued on the caller's stream
    torch.cuda.set_stream(train_stream())          # the stream rule of simvg_amd/graphs.py
    try:
        for step in range(7):
            batch = _batch(cfg, B=4, seed=20 + step)
            res = []
            for graph in (False, True):
                model.head_graph = graph
                model.zero_grad(set_to_none=True)
                losses, preds = model(**batch, rescale=False)
                losses["loss_total"].backward()
                res.append(({k: v.detach().clone() for k, v in losses.items()}, [p["pred_bboxes"].clone() for p in preds],
                            {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
                del losses, preds
            used_graph.append(len(model._head_graphs.graphs) > 0)
            (l0, p0, g0), (l1, p1, g1) = res
            for k in l0:
                assert torch.allclose(l0[k].float(), l1[k].float(), rtol=1e-5, atol=1e-6), (step, k, float(l0[k]), float(l1[k]))
            for a, b in zip(p0, p1):
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-4), step
            assert g0.keys() == g1.keys()
            for n in g0:
                ref = g0[n].float()
                tol = 1e-3 * float(ref.abs().max()) + 1e-7     # fp32 atomics order in the wgrad / LN reductions
                assert float((ref - g1[n].float()).abs().max()) <= tol, (step, n)
            with torch.no_grad():
                for n, p in model.named_parameters():
                    if n in g0:
                        p.add_(g0[n], alpha=-1e-3)
            model.vis_enc.mark_weights_dirty()
    finally:
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream())
    assert not used_graph[0] and used_graph[-1] and not model._head_graphs.disabled


def test_graphed_head_draws_fresh_dropout_masks_and_falls_back_on_new_shapes():
    from test_tools_gpu import _tiny_model, _batch
    cfg, model = _tiny_model(4)
    model.vis_enc.drop_path_probs = [0.0] * model.vis_enc.L
    model.head_graph = True
    model.train()
    batch = _batch(cfg, B=4, seed=1)
    vals = []
    from simvg_amd.graphs import train_stream
    train_stream().wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(train_stream())
    for _ in range(7):
        model.zero_grad(set_to_none=True)
        losses, _ = model(**batch, rescale=False)
        losses["loss_total"].backward()
        vals.append(float(losses["loss_total"]))
    assert len(model._head_graphs.graphs) == 1
    assert len(set(round(v, 6) for v in vals[4:])) > 1      # same input, replayed graph: dropout masks differ per replay
    small = _batch(cfg, B=2, seed=2)                        # another signature: eager for its first steps, no error
    losses, _ = model(**small, rescale=False)
    assert torch.isfinite(losses["loss_total"]) and len(model._head_graphs.graphs) == 1
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream())


def test_default_stream_training_never_captures():
    from test_tools_gpu import _tiny_model, _batch
    cfg, model = _tiny_model(5)
    model.head_graph = True
    model.train()
    batch = _batch(cfg, B=4, seed=1)
    for _ in range(6):
        model.zero_grad(set_to_none=True)
        losses, _ = model(**batch, rescale=False)
        losses["loss_total"].backward()
    assert model._head_graphs.default_stream_seen and not model._head_graphs.graphs


def test_graphed_forward_test_equals_eager_and_follows_the_weights():
    """forward_test as one hipGraph per input signature (simvg_amd/graphs.py::InferenceGraphs): bit-identical boxes to the
    eager path on fresh inputs, still after the master weights moved (load_state_dict -> the 16-bit copies are refreshed
    before the replay), a second signature gets its own graph, and a deep copy of the model starts without graphs."""
    from test_tools_gpu import _tiny_model, _batch
    cfg, model = _tiny_model(5)
    model.eval()

    def infer(m, batch, graph):
        m.infer_graph = graph
        kw = {k: v for k, v in batch.items() if k != "gt_bbox"}
        with torch.no_grad():
            preds = m(**kw, return_loss=False, rescale=False)
        return [p["pred_bboxes"].clone() for p in preds], {k: v.clone() for k, v in m._last_output["decoder_branch_output"].items()}

    for step in range(5):
        batch = _batch(cfg, B=2, seed=40 + step)
        (b_e, o_e), (b_g, o_g) = infer(model, batch, False), infer(model, batch, True)
        for a, b in zip(b_e, b_g):
            assert torch.equal(a, b), step
        for k in o_e:
            assert torch.equal(o_e[k], o_g[k]), (step, k)
    ig = model._infer_graphs
    assert len(ig.graphs) == 1 and ig.replays >= 2 and not ig.disabled
    # the weights move: every floating-point tensor of the state is perturbed and loaded back
    sd = {k: (v + 0.01 * torch.randn_like(v) if v.is_floating_point() else v) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    batch = _batch(cfg, B=2, seed=99)
    before = ig.replays
    (b_g, o_g), (b_e, o_e) = infer(model, batch, True), infer(model, batch, False)
    assert ig.replays == before + 1
    for k in o_e:
        assert torch.equal(o_e[k], o_g[k]), k
    # another batch size: its own signature, captured after two eager calls
    for step in range(4):
        batch3 = _batch(cfg, B=3, seed=60 + step)
        (b_e, o_e), (b_g, o_g) = infer(model, batch3, False), infer(model, batch3, True)
        for k in o_e:
            assert torch.equal(o_e[k], o_g[k]), (step, k)
    assert len(ig.graphs) == 2
    twin = copy.deepcopy(model)
    assert twin._infer_graphs is None
    (b_t, o_t) = infer(twin, batch3, True)
    for k in o_e:
        assert torch.equal(o_e[k], o_t[k]), k


def test_inference_graphs_of_rebuilt_weight_buffers_are_dropped():
    """Captured graphs are keyed on the serial numbers of the 16-bit weight buffers their launches point into.  When those buffers
    are rebuilt (here: the head's preparation object is discarded, as FlatAdam re-pointing p.data does) the old graphs leave the
    table -- they neither replay against freed memory nor occupy the slots -- and the signature is captured again."""
    from test_tools_gpu import _tiny_model, _batch
    cfg, model = _tiny_model(6)
    model.eval()
    model.infer_graph = True

    def infer(batch):
        kw = {k: v for k, v in batch.items() if k != "gt_bbox"}
        with torch.no_grad():
            return [p["pred_bboxes"].clone() for p in model(**kw, return_loss=False, rescale=False)]

    batch = _batch(cfg, B=2, seed=7)
    for _ in range(4):
        ref = infer(batch)
    ig = model._infer_graphs
    assert len(ig.graphs) == 1
    (old_sig,) = ig.graphs
    gen0 = model.head._prep.generation
    model.head._prep = None                                   # the next call rebuilds the head's 16-bit buffers
    for _ in range(4):
        got = infer(batch)
    assert model.head._prep.generation > gen0
    assert old_sig not in ig.graphs and len(ig.graphs) == 1   # the stale graph is gone, the signature was captured again
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


def test_graphed_head_decoder_only_equals_eager():
    """branch_loss_weight={"decoder": 1.0} (21 of the 53 reference configs): no token branch -- the graphed step returns no token
    tensors, `HeadGraphs.run` hands back None placeholders like the eager head, losses / gradients equal the eager step's and the
    capture does not fall back to eager."""
    import warnings
    from test_tools_gpu import CFG
    from simvg_amd.config import Config
    from simvg_amd.graphs import train_stream
    from simvg_amd.models import build_model
    from test_tools_gpu import _batch
    torch.manual_seed(6)
    cfg = Config.fromfile(CFG)
    cfg.model.head.branch_loss_weight = {"decoder": 1.0}
    model = build_model(cfg.model).to("cuda")
    model.vis_enc._ensure_engine(torch.device("cuda"))
    _no_dropout(model)
    model.train()
    train_stream().wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(train_stream())
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error")              # "head hipGraph capture failed" must not appear
            for step in range(6):
                batch = _batch(cfg, B=4, seed=40 + step)
                res = []
                for graph in (False, True):
                    model.head_graph = graph
                    model.zero_grad(set_to_none=True)
                    losses, preds = model(**batch, rescale=False)
                    losses["loss_total"].backward()
                    assert list(losses) == ["loss_dgt", "loss_total"]
                    assert preds[1]["pred_bboxes"] is None and model._last_output["token_branch_output"]["pred_logits"] is None
                    res.append(({k: v.detach().clone() for k, v in losses.items()}, preds[0]["pred_bboxes"].clone(),
                                {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
                (l0, p0, g0), (l1, p1, g1) = res
                for k in l0:
                    assert torch.allclose(l0[k].float(), l1[k].float(), rtol=1e-5, atol=1e-6), (step, k)
                assert torch.allclose(p0, p1, rtol=1e-5, atol=1e-4)
                assert g0.keys() == g1.keys()
                for n in g0:
                    tol = 1e-3 * float(g0[n].float().abs().max()) + 1e-7
                    assert float((g0[n].float() - g1[n].float()).abs().max()) <= tol, (step, n)
    finally:
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream())
    assert len(model._head_graphs.graphs) == 1 and not model._head_graphs.disabled
