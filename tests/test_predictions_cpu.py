"""CPU: the lazily materialised prediction containers of MIXDETRMB equal the reference's eager formulation
(mix_detr_mb.py:127-190: boolean indexing per image / over the batch)."""
import torch


def _case(seed=0, B=5, nq=7):
    g = torch.Generator().manual_seed(seed)
    xyxy = torch.rand(B, nq, 4, generator=g) * 100
    scores = torch.rand(B, nq, generator=g)
    labels = torch.randint(0, 3, (B, nq), generator=g)
    keep = torch.rand(B, nq, generator=g) > 0.4
    keep[1] = False          # an image without a surviving query
    keep[2] = True           # ... and one that keeps them all
    return xyxy, scores, labels, keep


def test_kept_instances_equal_boolean_indexing_per_image():
    from simvg_amd.models.det_seg.mix_detr_mb import KeptInstances
    xyxy, scores, labels, keep = _case()
    inst = KeptInstances(xyxy, scores, labels, keep)
    assert len(inst) == xyxy.shape[0] and inst._items is None            # nothing materialised (no host sync) yet
    for b, item in enumerate(inst):
        k = keep[b]
        assert torch.equal(item["boxes"], xyxy[b][k]) and torch.equal(item["scores"], scores[b][k])
        assert torch.equal(item["labels"], labels[b][k])
    assert inst[1]["boxes"].shape == (0, 4) and inst[2]["scores"].shape == (xyxy.shape[1],)
    assert [len(d["scores"]) for d in inst] == keep.sum(1).tolist()


def test_kept_labels_behave_like_the_concatenated_tensor():
    from simvg_amd.models.det_seg.mix_detr_mb import KeptLabels
    _, _, labels, keep = _case(3)
    ref = labels[keep]
    lazy = KeptLabels(labels, keep)
    assert lazy._t is None
    assert len(lazy) == len(ref) and lazy.shape == ref.shape and lazy.dtype == ref.dtype
    assert torch.equal(lazy.tensor(), ref) and torch.equal(lazy.cpu(), ref) and lazy.tolist() == ref.tolist()
    assert torch.equal(torch.cat([lazy, torch.tensor([7])]), torch.cat([ref, torch.tensor([7])]))
    assert int(lazy[0]) == int(ref[0]) and [int(x) for x in lazy] == ref.tolist()


def test_grec_metric_with_unequal_kept_counts():
    """images keep different numbers of queries (a box that is empty after clipping is dropped): the GRefCOCO metric takes
    the lazily materialised container as it is (one padded device-to-host copy, cut per image on the host) and agrees
    with the same predictions passed as a plain list of per-image dicts of different lengths"""
    from simvg_amd.apis import grec_evaluate_f1_nacc
    from simvg_amd.models.det_seg.mix_detr_mb import KeptInstances
    g = torch.Generator().manual_seed(7)
    B, nq = 5, 6
    xy = torch.rand(B, nq, 2, generator=g) * 300
    xyxy = torch.cat([xy, xy + 20 + torch.rand(B, nq, 2, generator=g) * 200], -1)
    scores = torch.rand(B, nq, generator=g)
    scores[:, 0] = 0.95
    labels = torch.zeros(B, nq, dtype=torch.int64)
    keep = torch.ones(B, nq, dtype=torch.bool)
    keep[0, 1:] = False          # one kept query
    keep[1, ::2] = False         # three
    keep[3] = False              # none at all
    inst = KeptInstances(xyxy, scores, labels, keep)
    assert sorted({len(p["scores"]) for p in inst}) == [0, 1, 3, 6]
    gts = [xyxy[b, :2].clone() for b in range(B)]
    gts[2] = torch.zeros(1, 4)
    targets = [[{"category_id": 1}, {"category_id": 1}] for _ in range(B)]
    targets[2] = [{"category_id": -1}]
    f1_a, na_a = grec_evaluate_f1_nacc(inst, gts, targets, device="cpu")
    plain = [{"boxes": xyxy[b][keep[b]], "scores": scores[b][keep[b]], "labels": labels[b][keep[b]]} for b in range(B)]
    f1_b, na_b = grec_evaluate_f1_nacc(plain, gts, targets, device="cpu")
    assert float(f1_a) == float(f1_b) and float(na_a) == float(na_b)
    assert 0.0 <= float(f1_a) <= 100.0
