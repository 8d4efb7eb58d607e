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
