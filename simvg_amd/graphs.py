"""OPTIONAL hipGraph replay of the launch-bound part of the training step: the decoder head + matcher + criterion
(`MIXDETRMB(head_graph=True)` or SIMVG_HEAD_GRAPH=1; off by default).

History: when the head was ~650 small launches forward and ~700 backward through per-op autograd nodes, Python / autograd
dispatch cost ~13 ms per step and the MI355X idled while the CPU fed it (a flat 21.5 ms per step at B=2); replaying the
head as two hipGraphs (`torch.cuda.make_graphed_callables`, one capture per input signature) removed that.  Since the
decoder layers became single autograd nodes with grouped GEMM launches (~350 launches for the whole head, 5 ms of host time
per direction) the eager head keeps the host ahead of the GPU at every batch size, and the graphs lose: `hipGraphLaunch` of
the captured BACKWARD blocks the host until the stream has drained (10 ms inside `CUDAGraph.replay()` per step; the forward
graph's launch costs 0.07 ms), which throws the host's lead away every step (B=64: 1736 vs 1782 pairs/s; table in
profiles/r01_sweeps.md).  The machinery stays for hosts that are slower relative to the GPU.

What makes the capture legal:
  * everything data-dependent that needs the host -- target packing, the loss normalisers and their RCCL all-reduce --
    happens BEFORE the graphed region (`head.prepare_targets`); the region itself contains no collective, no host copy,
    no synchronisation (the matcher and the criterion run on the device);
  * dropout / attention-dropout masks come from PyTorch's graph-safe Philox generator (fresh masks on every replay);
  * the 16-bit weight refresh of the head's memory projections is part of the captured forward, so replays always see
    the current master weights.
A signature = (shapes, dtypes, per-image `img_shape`s, train/eval mode).  Anything else (first steps, a ragged last
batch, GRefCOCO-style variable metas) runs eagerly; a failed capture disables the feature with a warning.

Stream rule: the training step must run on a NON-default HIP stream (`with simvg_amd.graphs.training_stream():`
-- bench.py, `train_model` and the tests do).  The autograd engine waits, at the end of every backward,
on the stream each parameter's gradient accumulator was created on; accumulators created by an eager step on the legacy
default stream make the captured backward touch that stream, which is illegal during a global capture (this HIP runtime
dies in hipStreamEndCapture instead of reporting it).  A model that has ever run a training forward on the default
stream therefore never captures -- it silently stays eager.
"""
import time
import warnings

import torch


_train_streams = {}


def train_stream(device=None):
    """The (per-device, process-wide) non-default stream training steps should run on -- see the stream rule above."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    s = _train_streams.get(device.index)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _train_streams[device.index] = s
    return s


class training_stream:
    """`with training_stream(device):` -- make the training stream current, ORDERED against the caller's stream on both
    sides: the side stream first waits for whatever the caller already queued (model.to(device), checkpoint loads, input
    copies), and on exit the caller's stream waits for the steps (so a following evaluation or save sees them)."""

    def __init__(self, device=None):
        self.side = train_stream(device)
        self.outer = torch.cuda.current_stream(self.side.device)
        self._ctx = torch.cuda.stream(self.side)

    def __enter__(self):
        self.side.wait_stream(self.outer)
        self._ctx.__enter__()
        return self.side

    def __exit__(self, *exc):
        self._ctx.__exit__(*exc)
        self.outer.wait_stream(self.side)
        return False


class _HeadStep(torch.nn.Module):
    """enc_out, text mask, packed targets -> (loss_total, the configured logged loss terms, 4 prediction tensors[, post-processed boxes
    and classes of both branches]); parameters = the head's.  Only `loss_total` is differentiable: the other outputs are
    detached, so the replayed backward takes one incoming gradient instead of materialising zeros for eight."""

    def __init__(self, head, B, Nv, T, img_metas, predict_fn=None):
        super().__init__()
        self.head = head
        self.geo = (B, Nv, T)
        self.img_metas = img_metas
        self.predict_fn = predict_fn

    def forward(self, enc_out, enc_lp, text_mask, tboxes, tlabels, tcount, nums):
        B, Nv, T = self.geo
        enc_out.lp = enc_lp           # the 16-bit copy of the same rows (operand of the memory projections)
        out = self.head.forward_fused(enc_out, B, Nv, T, self.img_metas, text_mask)
        losses, _ = self.head.loss_from_targets(out, tboxes, tlabels, tcount, nums)
        t, d = out["token_branch_output"], out["decoder_branch_output"]
        logged = tuple(losses[k].detach() for k in self.head.loss_keys if k != "loss_total")
        # a decoder-only head (branch_loss_weight={"decoder": w}) has NO token branch: its output dict carries None, and a graphed
        # callable can only return tensors -- the token entries are emitted only when the branch exists (HeadGraphs.run mirrors it)
        has_tok = t["pred_logits"] is not None
        res = (losses["loss_total"],) + logged
        if has_tok:
            res = res + (t["pred_logits"].detach(), t["pred_boxes"].detach())
        res = res + (d["pred_logits"].detach(), d["pred_boxes"].detach())
        if self.predict_fn is not None:
            with torch.no_grad():      # get_predictions of both branches rides in the forward graph (no eager launches)
                dec, tok = self.predict_fn(out, self.img_metas)
            res = res + (dec["pred_bboxes"], dec["predict_classes"])
            if has_tok:
                res = res + (tok["pred_bboxes"], tok["predict_classes"])
        return res


class HeadGraphs:
    """Per-signature cache of graphed head steps; `run()` returns (losses dict, output dict for `_predict`, predictions or
    None) or None when the step must run eagerly."""

    def __init__(self, head, warm_steps=3):
        self.head = head
        self.warm_steps = warm_steps
        self.seen = {}
        self.graphs = {}
        self.disabled = False
        self.default_stream_seen = False

    @staticmethod
    def _signature(enc_out, text_mask, tboxes, img_metas, training):
        return (tuple(enc_out.shape), enc_out.dtype, tuple(text_mask.shape), text_mask.dtype, tuple(tboxes.shape),
                tuple(tuple(m["img_shape"][:2]) for m in img_metas), bool(training))

    def run(self, enc_out, B, Nv, T, img_metas, text_mask, targets, predict_fn=None, sig_extra=()):
        """`predict_fn(output, img_metas) -> [decoder prediction dict, token prediction dict]` (tensor-only, sync-free
        post-processing) is captured with the forward when given; the third return value is then its result."""
        if self.disabled or not torch.is_grad_enabled() or not enc_out.requires_grad:
            return None
        if torch.cuda.current_stream(enc_out.device) == torch.cuda.default_stream(enc_out.device):
            self.default_stream_seen = True      # gradient accumulators now live on the legacy stream: never capture
        if self.default_stream_seen:
            return None
        tboxes, tlabels, tcount, nums = targets
        sig = self._signature(enc_out, text_mask, tboxes, img_metas, self.head.training) + (predict_fn is not None,) + tuple(sig_extra)
        g = self.graphs.get(sig)
        if g is None:
            n = self.seen.get(sig, 0)
            self.seen[sig] = n + 1
            if n < self.warm_steps:          # lazily created workspaces / constants must exist before the capture
                return None
            try:
                if torch.distributed.is_available() and torch.distributed.is_initialized():
                    # drain every RCCL operation and give the process group's watchdog thread (100 ms poll) time to retire
                    # them: an event query from that thread while this thread captures in global mode would abort the capture
                    torch.cuda.synchronize()
                    time.sleep(0.5)
                mod = _HeadStep(self.head, B, Nv, T, [dict(m) for m in img_metas], predict_fn)
                mod.train(self.head.training)
                sample = (enc_out.detach().clone().requires_grad_(True), enc_out.lp.clone(), text_mask.clone(), tboxes.clone(),
                          tlabels.clone(), tcount.clone(), nums.clone())
                g = torch.cuda.make_graphed_callables(mod, sample, num_warmup_iters=2, allow_unused_input=True)
            except Exception as e:      # noqa: BLE001 -- any capture problem: stay eager, say so once
                warnings.warn(f"simvg_amd: head hipGraph capture failed ({type(e).__name__}: {e}); running eagerly")
                self.disabled = True
                torch.cuda.synchronize()
                return None
            self.graphs[sig] = g
        outs = g(enc_out, enc_out.lp, text_mask, tboxes, tlabels, tcount, nums)
        keys = [k for k in self.head.loss_keys if k != "loss_total"]      # the head's configured branches (reference order)
        n = 1 + len(keys)
        named = dict(zip(["loss_total"] + keys, outs[:n]))
        losses = {k: named[k] for k in self.head.loss_keys}
        outs = list(outs[n:])
        has_tok = "balanced_distill" in self.head.branch_loss_weight
        tl, tb = (outs.pop(0), outs.pop(0)) if has_tok else (None, None)
        output = dict(token_branch_output={"pred_logits": tl, "pred_boxes": tb},
                      decoder_branch_output={"pred_logits": outs.pop(0), "pred_boxes": outs.pop(0)})
        preds = None
        if outs:
            preds = [dict(pred_bboxes=outs.pop(0), pred_masks=None, predict_classes=outs.pop(0))]
            preds.append(dict(pred_bboxes=outs.pop(0), pred_masks=None, predict_classes=outs.pop(0)) if has_tok
                         else dict(pred_bboxes=None, pred_masks=None, predict_classes=None))
        return losses, output, preds


class InferenceGraphs:
    """hipGraph replay of `MIXDETRMB._run` (encoder + head forward, eval mode) -- one graph per input signature.

    `forward_test` at small batch is ~220 launches of a few microseconds each: the GPU finishes them faster than Python
    issues them (B = 1: 2.9 ms per call eager against ~2 ms of kernel time after the small-problem GEMM / attention
    changes).  A signature = (input shapes and dtypes, per-image `img_shape`s, the identity of the two modules' 16-bit weight
    tables); it is captured after it has been seen `warm_calls` times, at most `max_graphs` signatures are kept, anything
    else runs eagerly.  The 16-bit weight copies are refreshed EAGERLY before every replay (a no-op unless the master
    weights moved: load_state_dict, EMA apply_shadow / restore, a training step), so the captured region holds no
    weight-dependent decision.  The returned tensors are the graph's static outputs: consume them (the post-processing
    does) before the next call with the same signature."""

    def __init__(self, model, warm_calls=2, max_graphs=8, max_batch=16):
        self.model, self.warm_calls, self.max_graphs, self.max_batch = model, warm_calls, max_graphs, max_batch
        self.seen, self.graphs, self.disabled = {}, {}, False
        self.replays = 0
        self._gen = None                   # generation of the weight buffers the kept graphs were captured against

    def __deepcopy__(self, memo):      # a copied model captures its own graphs (the attribute is re-created lazily)
        return None

    def __reduce__(self):              # ... and so does an unpickled one
        return (type(None), ())

    def _signature(self, img, ids, mask, img_metas):
        enc, head = self.model.vis_enc, self.model.head
        return (tuple(img.shape), img.dtype, tuple(ids.shape), None if mask is None else (tuple(mask.shape), mask.dtype),
                tuple(tuple(m["img_shape"][:2]) for m in img_metas),
                tuple(img_metas[0].get("batch_input_shape", ())), getattr(enc, "precision", None),
                # which kernels the captured forward launches: hi + lo weights or single ones, and for how many layers / Linears
                (bool(getattr(enc, "precise_inference", False)), getattr(enc, "precise_layers", 0),
                 getattr(enc, "precise_which", ()), getattr(enc, "wb2", None) is not None)) + self._generation()

    def _generation(self):
        """(serial number of the encoder's, of the head's 16-bit weight buffers): a captured graph's launches point into exactly
        these buffers; when either set is rebuilt (FlatAdam re-pointing p.data, a device move) older graphs are dropped"""
        enc, head = self.model.vis_enc, self.model.head
        return (getattr(getattr(enc, "_prep", None), "generation", 0), getattr(getattr(head, "_prep", None), "generation", 0))

    def _refresh(self, device):
        enc, head = self.model.vis_enc, self.model.head
        if getattr(enc, "_prep", None) is not None:
            enc._refresh_weights()
        if getattr(head, "_prep", None) is not None:
            head._refresh_weights(device)

    def run(self, img, ids, img_metas, mask):
        """-> the head's output dict or None (caller runs eagerly).  The tensors of the dict are the graph's STATIC outputs: the next
        replay of the same signature overwrites them (`MIXDETRMB._last_output` aliases them; the predictions `forward_test` returns
        are computed from them into fresh tensors and stay valid)."""
        model = self.model
        if self.disabled or model.training or torch.is_grad_enabled() or not img.is_cuda or img.shape[0] > self.max_batch:
            return None
        if torch.cuda.is_current_stream_capturing():
            return None
        sig = self._signature(img, ids, mask, img_metas)
        gen = sig[-2:]
        if self._gen != gen:               # graphs of older weight buffers: their pointers are stale, and they would hold the slots
            self.graphs = {k: v for k, v in self.graphs.items() if k[-2:] == gen}
            self._gen = gen
        entry = self.graphs.get(sig)
        if entry is None:
            n = self.seen.get(sig, 0)
            self.seen[sig] = n + 1
            if n < self.warm_calls or len(self.graphs) >= self.max_graphs:
                if len(self.seen) > 256:
                    self.seen.clear()
                return None
            entry = self._capture(sig, img, ids, img_metas, mask)
            if entry is None:
                return None
        g, s_img, s_ids, s_mask, out = entry
        self._refresh(img.device)
        s_img.copy_(img, non_blocking=True)
        s_ids.copy_(ids, non_blocking=True)
        if s_mask is not None:
            s_mask.copy_(mask, non_blocking=True)
        g.replay()
        self.replays += 1
        return out

    def _capture(self, sig, img, ids, img_metas, mask):
        model = self.model
        try:
            s_img, s_ids = img.clone(), ids.clone()
            s_mask = None if mask is None else mask.clone()
            metas = [dict(m) for m in img_metas]
            cur = torch.cuda.current_stream(img.device)
            side = torch.cuda.Stream(device=img.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                model._run(s_img, s_ids, metas, s_mask)          # workspaces / constants of this signature exist before capture
            cur.wait_stream(side)
            torch.cuda.synchronize(img.device)
            self._refresh(img.device)
            g = torch.cuda.CUDAGraph()
            # thread_local: a loader thread may be issuing its own device work while this thread captures
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                out = model._run(s_img, s_ids, metas, s_mask)
            torch.cuda.synchronize(img.device)
        except Exception as e:      # noqa: BLE001  (a failed capture must never take inference down)
            warnings.warn(f"inference graph capture failed ({type(e).__name__}: {e}); forward_test stays eager")
            self.disabled = True
            return None
        entry = (g, s_img, s_ids, s_mask, out)
        self.graphs[sig] = entry
        return entry
