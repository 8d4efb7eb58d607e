"""MIXDETRMB -- SimVG's model class on MI355X.

Mirror of `simvg/models/det_seg/mix_detr_mb.py:13-190` (registered in MODELS; `forward_train` returns
`(loss_dict, [pred_decoder, pred_token])`, `forward_test` returns `[pred_decoder, pred_token]`, each a dict with
`pred_bboxes` / `pred_masks` / `predict_classes`).  The encoder -> head hand-off stays in the modality-major
bf16 layout (no `[B,C,h,w]` transpose copy, `mix_detr_mb.py:52` of the reference), and the post-processing of
`get_predictions` (detectron2 Boxes.scale / clip / nonempty + argmax) is one HIP kernel (`simvg_postprocess`).
"""
import os
from collections.abc import Sequence

import torch

from .. import builder
from .base import OneStageModel


class KeptInstances(Sequence):
    """`pred_bboxes` of the GRefCOCO path: per image {"boxes", "scores", "labels"} of the kept queries, as a read-only
    sequence that is materialised on first access.  The counts per image need ONE device-to-host copy; deferring it keeps
    forward_train free of host synchronisation, so the host can enqueue the backward while the GPU is still busy with the
    forward -- by the time the training loop reads the predictions for its metrics the copy no longer stalls anything."""

    def __init__(self, xyxy, scores, labels, keep):
        # a stable sort brings the kept queries to the front of every row in their original order (what the reference's
        # boolean indexing per image and field returns); the slices taken later are views
        order = torch.argsort((~keep).to(torch.uint8), dim=1, stable=True)
        self._xyxy = xyxy.gather(1, order[..., None].expand(-1, -1, 4))
        self._scores, self._labels = scores.gather(1, order), labels.gather(1, order)
        self._kept = keep.sum(1)
        self._items = None

    def _materialise(self):
        if self._items is None:
            counts = self._kept.tolist()
            self._items = [{"boxes": self._xyxy[b, :c], "scores": self._scores[b, :c], "labels": self._labels[b, :c]}
                           for b, c in enumerate(counts)]
        return self._items

    def __len__(self):
        return int(self._xyxy.shape[0])

    def __getitem__(self, i):
        return self._materialise()[i]

    def host_arrays(self):
        """per image (scores float64 [k_b], boxes float32 [k_b, 4]) as numpy -- the kept counts differ between images, so
        the padded rows and the counts cross to the host in ONE copy each and are cut there (the GRefCOCO metric's input)"""
        counts = self._kept.tolist()
        scores, boxes = self._scores.detach().double().cpu().numpy(), self._xyxy.detach().float().cpu().numpy()
        return [(scores[b, :c], boxes[b, :c]) for b, c in enumerate(counts)]


class KeptLabels:
    """`predict_classes` for num_queries > 1: the classes of ALL kept queries of the batch, concatenated
    (mix_detr_mb.py:152,157) -- a tensor whose length is data dependent, i.e. a host synchronisation.  Nothing on the
    training path reads it, so it is computed on first use; until then this object stands in for the tensor
    (attribute access, indexing, len() and torch functions resolve to the materialised tensor)."""

    def __init__(self, labels, keep):
        self._labels, self._keep, self._t = labels, keep, None

    def tensor(self):
        if self._t is None:
            self._t = self._labels[self._keep]
        return self._t

    def __getattr__(self, name):
        if name.startswith("_"):       # private / dunder look-ups (copy, pickle) must not trigger the materialisation
            raise AttributeError(name)
        return getattr(self.tensor(), name)

    def __len__(self):
        return len(self.tensor())

    def __getitem__(self, i):
        return self.tensor()[i]

    def __iter__(self):
        return iter(self.tensor())

    def __repr__(self):
        return "KeptLabels(" + repr(self.tensor()) + ")"

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def unwrap(x):
            if isinstance(x, KeptLabels):
                return x.tensor()
            if isinstance(x, (list, tuple)):
                return type(x)(unwrap(y) for y in x)
            return x
        return func(*unwrap(args), **{k: unwrap(v) for k, v in (kwargs or {}).items()})


@builder.MODELS.register_module()
class MIXDETRMB(OneStageModel):
    def __init__(self, word_emb, num_token, vis_enc, lan_enc, head, fusion, head_graph=False, infer_graph=True):
        super().__init__(word_emb, num_token, vis_enc, lan_enc, head, fusion)
        self.patch_size = vis_enc["patch_size"]
        # optional: replay the head (+ matcher + criterion) forward / backward as two hipGraphs once a training input
        # signature has repeated (simvg_amd/graphs.py).  Off by default: since the head's launch count was cut to ~45 per
        # decoder layer the eager head keeps the host ahead of the GPU at every batch size, while replaying the backward
        # graph blocks the host until the stream has drained (measured: profiles/r01_sweeps.md).  `head_graph=True` in the
        # model cfg or SIMVG_HEAD_GRAPH=1 turns it on, SIMVG_HEAD_GRAPH=0 forces it off.
        env = os.environ.get("SIMVG_HEAD_GRAPH")
        self.head_graph = (bool(head_graph) or env == "1") and env != "0"
        self._head_graphs = None
        # forward_test at batch <= 16 replays encoder + head as one hipGraph per input signature once the signature has
        # repeated (simvg_amd/graphs.py::InferenceGraphs); `infer_graph=False` in the model cfg or SIMVG_INFER_GRAPH=0: eager
        self.infer_graph = bool(infer_graph) and os.environ.get("SIMVG_INFER_GRAPH") != "0"
        self._infer_graphs = None
        self._pp_const = {}

    def extract_visual_language(self, img, ref_expr_inds, text_attention_mask=None):
        return self.vis_enc(img, ref_expr_inds, text_attention_mask)

    def _run(self, img, ref_expr_inds, img_metas, text_attention_mask):
        B, T = ref_expr_inds.shape
        enc_out = self.vis_enc.encode(img, ref_expr_inds, text_attention_mask)
        Nv = self.vis_enc.np + 1
        return self.head.forward_fused(enc_out, B, Nv, T, img_metas, text_attention_mask)

    def forward_train(self, img, ref_expr_inds, img_metas, text_attention_mask=None, gt_bbox=None,
                      gt_mask_vertices=None, rescale=False):
        B, T = ref_expr_inds.shape
        Nv = self.vis_enc.np + 1
        enc_out = self.vis_enc.encode(img, ref_expr_inds, text_attention_mask)
        targets = self.head.prepare_targets(gt_bbox, img_metas, enc_out.device)
        graphed = None
        if self.head_graph and self.training and text_attention_mask is not None and getattr(enc_out, "lp", None) is not None:
            if self._head_graphs is None:
                from ...graphs import HeadGraphs
                self._head_graphs = HeadGraphs(self.head)
            # the single-box post-processing is sync-free tensor code: it is captured with the head's forward
            grec = img_metas[0].get("target", None) is not None
            fn = None
            if not grec and self.head.num_queries == 1 and not rescale:
                fn = (lambda out, metas: self._predict(out, metas, False))
            graphed = self._head_graphs.run(enc_out, B, Nv, T, img_metas, text_attention_mask, targets, predict_fn=fn,
                                              sig_extra=(bool(rescale),))
        predictions = None
        if graphed is not None:
            losses_dict, output, predictions = graphed
            self._last_output, self._last_detail = output, None
        else:
            output = self.head.forward_fused(enc_out, B, Nv, T, img_metas, text_attention_mask)
            losses_dict, detail = self.head.loss_from_targets(output, *targets)
            self._last_output, self._last_detail = output, detail     # debugging / parity tests
        if predictions is None:
            with torch.no_grad():
                predictions = self._predict(output, img_metas, rescale)
        return losses_dict, predictions

    @torch.no_grad()
    def forward_test(self, img, ref_expr_inds, img_metas, text_attention_mask=None, with_bbox=False, with_mask=False,
                     rescale=False):
        output = None
        if self.infer_graph and img.is_cuda:
            if self._infer_graphs is None:
                from ...graphs import InferenceGraphs
                self._infer_graphs = InferenceGraphs(self)
            output = self._infer_graphs.run(img, ref_expr_inds, img_metas, text_attention_mask)
        if output is None:
            output = self._run(img, ref_expr_inds, img_metas, text_attention_mask)
        self._last_output = output
        return self._predict(output, img_metas, rescale)

    def _predict(self, output, img_metas, rescale):
        fn = self.get_predictions if img_metas[0].get("target", None) is None else self.get_predictions_grec
        tok = fn(output["token_branch_output"], img_metas, rescale=rescale)
        dec = fn(output["decoder_branch_output"], img_metas, rescale=rescale)
        return [dec, tok]   # index 0 = decoder branch, 1 = token branch (mix_detr_mb.py:69,123)

    def _shape_consts(self, img_metas, device, rescale):
        """[w, h, w, h] per image (and the scale factors when rescaling) as device tensors, cached per batch geometry: no
        host-to-device copy in the steady state (and none inside a hipGraph capture)."""
        key = (str(device), tuple(tuple(m["img_shape"][:2]) for m in img_metas),
               tuple(tuple(float(x) for x in m["scale_factor"]) for m in img_metas) if rescale else None)
        c = self._pp_const.get(key)
        if c is None:
            lim = torch.tensor([[m["img_shape"][1], m["img_shape"][0]] * 2 for m in img_metas], dtype=torch.float32).to(device)
            sf = torch.tensor([list(m["scale_factor"]) for m in img_metas], dtype=torch.float32).to(device)[:, None, :] \
                if rescale else None
            if len(self._pp_const) >= 64:
                self._pp_const.clear()
            c = self._pp_const[key] = (lim, sf)
        return c

    def _post(self, output, img_metas, rescale):
        """one launch (`simvg_postprocess`): scores / labels / clipped (and rescaled) xyxy / keep per query, best kept box and
        its label per image"""
        from ... import hip_ops as ops
        lim, sf = self._shape_consts(img_metas, output["pred_boxes"].device, rescale)
        return ops.postprocess(output["pred_logits"], output["pred_boxes"], lim, sf)

    def get_predictions(self, output, img_metas, rescale=False):
        if output["pred_logits"] is None:
            return dict(pred_bboxes=None, pred_masks=None, predict_classes=None)
        scores, labels, xyxy, keep, box, best_label = self._post(output, img_metas, rescale)
        if scores.shape[1] == 1:
            cls = best_label
        else:   # the reference concatenates the classes of ALL kept queries (mix_detr_mb.py:152,157)
            cls = KeptLabels(labels, keep)
        return dict(pred_bboxes=box, pred_masks=None, predict_classes=cls)

    def get_predictions_grec(self, output, img_metas, rescale=False):
        if output["pred_logits"] is None:
            return dict(pred_bboxes=None, pred_masks=None, predict_classes=None)
        scores, labels, xyxy, keep, _, _ = self._post(output, img_metas, rescale)
        return dict(pred_bboxes=KeptInstances(xyxy, scores, labels, keep), pred_masks=None)
