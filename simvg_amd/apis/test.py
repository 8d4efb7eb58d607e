"""Evaluation loop and metrics -- mirror of the reference's `simvg/apis/test.py` (`accuracy` :70-88,
`grec_evaluate_f1_nacc` :91-163, `evaluate_model` :166-293).

Third-party leaves restated here (absent from the image, pinned through the imported reference functions in
`oracle/make_golden_apis.py` -> `tests/golden/apis_golden.json`):
  * mmdet `bbox_overlaps(b1, b2, is_aligned=True)` (mmdet 2.x `iou2d_calculator.py`): IoU of row i with row i,
    union clamped to eps = 1e-6;
  * torchvision `box_area`: (x2 - x1) * (y2 - y1).
Mask metrics (pycocotools RLE IoU) belong to the segmentation heads, which no reference config of this path uses;
`accuracy` returns the reference's placeholders for them and raises if masks are actually passed.

Per-batch metric values stay on the device; the running means are read back only when a log line is due (the
reference calls `.item()` on every batch, which would stall the launch queue of the MI355X once per step)."""
import time
from collections import defaultdict

import torch

from ..utils import get_root_logger, reduce_mean, is_main


def box_area(boxes):
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def bbox_overlaps_aligned(bboxes1, bboxes2, eps=1e-6):
    """mmdet bbox_overlaps(..., mode='iou', is_aligned=True)."""
    area1 = (bboxes1[..., 2] - bboxes1[..., 0]) * (bboxes1[..., 3] - bboxes1[..., 1])
    area2 = (bboxes2[..., 2] - bboxes2[..., 0]) * (bboxes2[..., 3] - bboxes2[..., 1])
    lt = torch.max(bboxes1[..., :2], bboxes2[..., :2])
    rb = torch.min(bboxes1[..., 2:], bboxes2[..., 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = torch.max(area1 + area2 - overlap, overlap.new_tensor([eps]))
    return overlap / union


def box_iou(boxes1, boxes2):
    area1 = box_area(boxes1)
    area2 = box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    return inter / union, union


def generalized_box_iou(boxes1, boxes2):
    assert (boxes1[:, 2:] >= boxes1[:, :2]).all()
    assert (boxes2[:, 2:] >= boxes2[:, :2]).all()
    iou, union = box_iou(boxes1, boxes2)
    lt = torch.min(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.max(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / area


def accuracy(pred_bboxes, gt_bbox, pred_masks, gt_mask, is_crowd=None, device="cuda:0"):
    """-> (Det@0.5 accuracy in %, mask IoU %, mask accuracy at 5 thresholds %)."""
    if pred_masks is not None:
        raise NotImplementedError("mask metrics need the segmentation heads, which are outside this hot path")
    det_acc = torch.tensor([0.0], device=device)
    if pred_bboxes is not None:
        gt = torch.stack(list(gt_bbox)).to(device)
        iou = bbox_overlaps_aligned(gt, pred_bboxes.to(device))
        det_acc = (iou >= 0.5).float().mean()
    mask_iou = torch.tensor([0.0], device=device)
    mask_acc_at_thrs = torch.full((5,), -1.0, device=device)
    return det_acc * 100.0, mask_iou * 100.0, mask_acc_at_thrs * 100.0


def grec_evaluate_f1_nacc(predictions, gt_bboxes, targets, thresh_score=0.7, thresh_iou=0.5, thresh_F1=1.0, device="cuda:0"):
    """Generalised-REC metrics (F1 @ score 0.7 / GIoU 0.5, no-target accuracy), reference :91-163.  Inherently a
    per-image greedy matching on a handful of boxes: done on the host after ONE device->host copy per batch."""
    if predictions is None:
        return torch.tensor(0.0, device=device).float(), torch.tensor(0.0, device=device).float()
    correct_image, num_image = 0, 0
    nt = {"TP": 0.0, "TN": 0.0, "FP": 0.0, "FN": 0.0}
    scores_all = torch.stack([p["scores"].reshape(-1) for p in predictions]).detach().cpu()
    boxes_all = torch.stack([p["boxes"].reshape(-1, 4) for p in predictions]).detach().cpu()
    for i, (gt_bbox, target) in enumerate(zip(gt_bboxes, targets)):
        order = sorted(zip(scores_all[i].tolist(), boxes_all[i].tolist()), reverse=True)
        sorted_scores = torch.tensor([s for s, _ in order], dtype=torch.float64)
        sorted_boxes = torch.tensor([b for _, b in order], dtype=torch.float32).view(-1, 4)
        no_target_flag = any(one_target["category_id"] == -1 for _, one_target in zip(gt_bbox, target))
        gt_bbox_all = torch.stack([torch.as_tensor(b) for b, _ in zip(gt_bbox, target)], dim=0).detach().cpu().float()
        filtered_boxes = sorted_boxes[sorted_scores >= thresh_score]
        giou = generalized_box_iou(filtered_boxes, gt_bbox_all.view(-1, 4))
        num_prediction, num_gt = filtered_boxes.shape[0], gt_bbox_all.shape[0]
        if no_target_flag:
            if num_prediction >= 1:
                nt["FN"] += 1
                F_1 = 0.0
            else:
                nt["TP"] += 1
                F_1 = 1.0
        else:
            if num_prediction >= 1:
                nt["TN"] += 1
            else:
                nt["FP"] += 1
            TP = 0
            for _ in range(min(num_prediction, num_gt)):
                top_value, top_index = torch.topk(giou.flatten(0, 1), 1)
                if top_value < thresh_iou:
                    break
                TP += 1
                giou[top_index[0] // num_gt, :] = 0.0
                giou[:, top_index[0] % num_gt] = 0.0
            FP = num_prediction - TP
            FN = num_gt - TP
            F_1 = 2 * TP / (2 * TP + FP + FN)
        if F_1 >= thresh_F1:
            correct_image += 1
        num_image += 1
    f1 = torch.tensor(correct_image / num_image, device=device)
    n_acc = torch.tensor(nt["TP"] / (nt["TP"] + nt["FN"]) if nt["TP"] != 0 else 0.0, device=device)
    return f1.float() * 100, n_acc.float() * 100


def _unwrap(v):
    """mmcv DataContainer -> its single-GPU payload; tensors / lists pass through."""
    return v.data[0] if hasattr(v, "data") and not isinstance(v, torch.Tensor) else v


def _split_gt(inputs, key="gt_bbox"):
    v = inputs[key]
    if isinstance(v, torch.Tensor):
        return [v[i] for i in range(v.shape[0])]
    return list(_unwrap(v))


class RunningMeans:
    """per-name running mean of per-batch device scalars, read back on demand (one sync per read, not per batch)"""

    def __init__(self):
        self.sum, self.n = {}, defaultdict(int)

    def add(self, name, value):
        value = value.detach().reshape(()).float()
        self.sum[name] = value if name not in self.sum else self.sum[name] + value
        self.n[name] += 1

    def names(self):
        return list(self.sum)

    def means(self):
        if not self.sum:
            return {}
        names = list(self.sum)
        vals = torch.stack([self.sum[k] for k in names]).tolist()
        return {k: v / self.n[k] for k, v in zip(names, vals)}


MAP_DICT = {0: "decoder", 1: "token"}


def evaluate_model(epoch, cfg, model, loader):
    from ..datasets import extract_data
    model.eval()
    device = next(model.parameters()).device
    batches = len(loader)
    end = time.time()
    with_bbox, with_mask = False, False
    grec = cfg["dataset"] == "GRefCOCO"
    det, f1, nacc = RunningMeans(), RunningMeans(), RunningMeans()
    n_pred = 1
    with torch.no_grad():
        for batch, inputs in enumerate(loader):
            inputs = dict(inputs)
            gt_bbox = None
            if "gt_bbox" in inputs:
                with_bbox = True
                gt_bbox = _split_gt(inputs)
                inputs.pop("gt_bbox")
            if "gt_mask_rle" in inputs:
                raise NotImplementedError("mask evaluation is outside this hot path")
            inputs.pop("is_crowd", None)
            img_metas = _unwrap(inputs["img_metas"])
            inputs = extract_data(inputs, device)
            predictions = model(**inputs, return_loss=False, rescale=False, with_bbox=with_bbox, with_mask=with_mask)
            predictions_list = predictions if isinstance(predictions, list) else [predictions]
            n_pred = len(predictions_list)
            for ind, predictions in enumerate(predictions_list):
                predict_type = MAP_DICT[ind]
                pred_bboxes = predictions.pop("pred_bboxes")
                pred_masks = predictions.pop("pred_masks")
                if not grec:
                    batch_det_acc, _, _ = accuracy(pred_bboxes, [g.to(device) for g in gt_bbox], pred_masks, None, device=device)
                    if cfg.distributed:
                        batch_det_acc = reduce_mean(batch_det_acc)
                    det.add(predict_type, batch_det_acc)
                else:
                    targets = [meta["target"] for meta in img_metas]
                    batch_f1_score, batch_n_acc = grec_evaluate_f1_nacc(pred_bboxes, gt_bbox, targets, device=device)
                    if cfg.distributed:
                        batch_f1_score, batch_n_acc = reduce_mean(batch_f1_score), reduce_mean(batch_n_acc)
                    f1.add(predict_type, batch_f1_score)
                    nacc.add(predict_type, batch_n_acc)
            if is_main() and ((batch + 1) % cfg.log_interval == 0 or batch + 1 == batches):
                logger = get_root_logger()
                if not grec:
                    m = det.means()
                    acc_str = "".join("{}Det@.5: {:.2f}, ".format(MAP_DICT[i], m[MAP_DICT[i]]) for i in range(n_pred))
                    logger.info(f"val - epoch [{epoch+1}]-[{batch+1}/{batches}] " + f"time: {(time.time()- end):.2f}, " + acc_str)
                else:
                    mf, mn = f1.means(), nacc.means()
                    f1_str = "".join("{}_f1_score: {:.2f}, ".format(MAP_DICT[i], mf[MAP_DICT[i]]) for i in range(n_pred))
                    na_str = "".join("{}_n_acc: {:.2f}, ".format(MAP_DICT[i], mn[MAP_DICT[i]]) for i in range(n_pred))
                    logger.info(f"Validate - epoch [{epoch+1}]-[{batch+1}/{batches}] " + f"time: {(time.time()- end):.2f}, "
                                + f1_str + na_str)
            end = time.time()
    if not grec:
        m = det.means()
        return sum(m.values()) / len(m), 0
    mf, mn = f1.means(), nacc.means()
    return sum(mf.values()) / len(mf), sum(mn.values()) / len(mn)
