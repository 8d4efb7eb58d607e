"""CPU: the file-bound head of the data pipeline (simvg_amd/datasets/{loading,refsets,tokenizer}.py) against
tests/golden/loading_golden.pt, recorded by EXECUTING the reference's `LoadImageAnnotationsFromFile` and vocabulary
builder on a miniature dataset (oracle/make_golden_loading.py).  The fixture carries the dataset itself -- annotation
records, JPEG bytes, the sentencepiece model trained for it -- so it is rebuilt in a temporary directory here.
Bit-exact: file names, decoded frames (BGR), chosen expression (same numpy draw), cleaned text, token ids / padding masks
(XLM-R alignment, truncation to max_token - 2, unknown pieces), word ids, clipped xyxy boxes, GRefCOCO targets."""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def mini(tmp_path_factory):
    fx = torch.load(os.path.join(HERE, "golden", "loading_golden.pt"), weights_only=False)
    root = str(tmp_path_factory.mktemp("refdata"))
    for rel, data in fx["images"].items():
        os.makedirs(os.path.dirname(os.path.join(root, rel)), exist_ok=True)
        with open(os.path.join(root, rel), "wb") as f:
            f.write(data)
    for name, s in fx["sets"].items():
        os.makedirs(os.path.join(root, "anns", name), exist_ok=True)
        with open(os.path.join(root, "anns", name, "instances.json"), "w") as f:
            json.dump(s["anns"], f)
    with open(os.path.join(root, "beit3.spm"), "wb") as f:
        f.write(fx["spm"])
    return fx, root


def _imgsfile(root, s):
    v = s["imgsfile"]
    return {k: os.path.join(root, d) for k, d in v.items()} if isinstance(v, dict) else os.path.join(root, v)


def test_loader_matches_the_reference_class_on_every_record(mini):
    from simvg_amd.datasets.loading import LoadImageAnnotationsFromFile
    fx, root = mini
    loaders = {}
    for c in fx["cases"]:
        s = fx["sets"][c["set"]]
        key = (c["set"], c["token_type"], c["max_token"])
        if key not in loaders:
            loaders[key] = LoadImageAnnotationsFromFile(dataset=s["dataset"], max_token=c["max_token"], with_bbox=True,
                                                        use_token_type=c["token_type"], spm_path=os.path.join(root, "beit3.spm"),
                                                        device="cpu")
        records = s["anns"][c["which_set"]]
        if c["which_set"] == "train" and records[0].get("data_source") is not None:
            records = [r for r in records if r["data_source"] in s["img_source"]]
        np.random.seed(c["seed"])
        res = loaders[key](dict(ann=json.loads(json.dumps(records[c["index"]])), which_set=c["which_set"],
                                token2idx=fx["vocab"][c["set"]], imgsfile=_imgsfile(root, s)))
        exp, tag = c["out"], (c["set"], c["which_set"], c["index"], c["token_type"], c["max_token"])
        assert loaders[key].random_ind == c["random_ind"], tag
        assert os.path.relpath(res["filename"], root) == exp["filename"], tag
        assert torch.equal(res["img"], fx["decoded"][exp["filename"]]) and res["img"].dtype == torch.uint8, tag
        assert tuple(res["img_shape"]) == tuple(exp["img_shape"]) == tuple(res["ori_shape"]), tag
        assert res["expression"] == exp["expression"] and res["max_token"] == exp["max_token"], tag
        assert np.array_equal(np.asarray(res["ref_expr_inds"]), np.asarray(exp["ref_expr_inds"])), tag
        if c["token_type"] == "beit3":
            assert np.array_equal(np.asarray(res["text_attention_mask"]), np.asarray(exp["text_attention_mask"])), tag
        else:
            assert "text_attention_mask" not in res
        if isinstance(exp["gt_bbox"], list):
            assert len(res["gt_bbox"]) == len(exp["gt_bbox"]), tag
            for a, b in zip(res["gt_bbox"], exp["gt_bbox"]):
                assert np.array_equal(np.asarray(a), np.asarray(b)), tag
            assert res["target"] == exp["target"], tag
        else:
            assert np.array_equal(np.asarray(res["gt_bbox"]), np.asarray(exp["gt_bbox"])), tag
        assert res["with_bbox"] is True and res["with_mask"] is False


def test_vocabulary_and_dataset_classes(mini):
    from simvg_amd.datasets import DATASETS, build_dataset
    from simvg_amd.datasets.refsets import build_vocabulary
    fx, root = mini
    for name, s in fx["sets"].items():
        annsfile = os.path.join(root, "anns", name, "instances.json")
        token2idx, idx2token, word_emb = build_vocabulary(annsfile, s["anns"], None)
        assert token2idx == fx["vocab"][name] and list(token2idx) == list(fx["vocab"][name]), name      # same ids, same order
        assert all(idx2token[i] == t for t, i in token2idx.items()) and word_emb.size == 0
        assert not os.path.exists(os.path.join(root, "anns", name, "token_to_ix.pkl"))                  # no files written unasked
    s = fx["sets"]["Mixed"]
    pipe = [dict(type="LoadImageAnnotationsFromFile", dataset="Mixed", max_token=20, with_bbox=True, use_token_type="beit3",
                 spm_path=os.path.join(root, "beit3.spm"), device="cpu")]
    ds = build_dataset(dict(type="Mixed", which_set="train", img_source=s["img_source"], imgsfile=_imgsfile(root, s),
                            annsfile=os.path.join(root, "anns", "Mixed", "instances.json"), pipeline=pipe))
    assert isinstance(ds, DATASETS.get("Mixed")) and len(ds) == 2 and ds.num_token == -1              # the visual-genome record is dropped
    assert ds.flag.tolist() == [1, 0]                                                                  # 80x60 landscape, 64x90 portrait
    item = ds[1]
    assert item["filename"].endswith(os.path.join("flickr", "2.jpg")) and item["expression"] == "left dog on the grass"
    val = build_dataset(dict(type="RefCOCOUNC", which_set="testA", img_source=["coco"], imgsfile=os.path.join(root, "coco"),
                             annsfile=os.path.join(root, "anns", "RefCOCOUNC", "instances.json"),
                             pipeline=[dict(type="LoadImageAnnotationsFromFile", dataset="RefCOCOUNC", max_token=8, with_bbox=True,
                                            device="cpu")]))
    assert val.num_token == len(fx["vocab"]["RefCOCOUNC"]) and not hasattr(val, "flag")
    assert val[0]["ref_expr_inds"].tolist()[:3] == [fx["vocab"]["RefCOCOUNC"][w] for w in ("zebra", "xylophone", "quartz")]
    with pytest.raises(ValueError):
        build_dataset(dict(type="RefCOCOUNC", which_set="nope", imgsfile=root, annsfile=annsfile, pipeline=pipe))
    with pytest.raises(NotImplementedError):
        build_dataset(dict(type="MixedSeg", which_set="train"))


def test_xlmr_alignment_rules(mini):
    from simvg_amd.datasets.tokenizer import XLMRTokenizer
    fx, root = mini
    tok = XLMRTokenizer(os.path.join(root, "beit3.spm"))
    n = len(tok.sp_model)
    assert (tok.bos_token_id, tok.pad_token_id, tok.eos_token_id, tok.unk_token_id) == (0, 1, 2, 3)
    assert tok.vocab_size == n + 2 and tok.mask_token_id == n + 1
    pieces = tok.tokenize("the man in the red shirt")
    assert tok.convert_tokens_to_ids(pieces) == [tok.sp_model.PieceToId(p) + 1 for p in pieces]
    assert tok.convert_tokens_to_ids(["<s>", "<pad>", "</s>", "<unk>", "<mask>", "▁qqqqzzzz"]) == [0, 1, 2, 3, n + 1, 3]
    ids, mask = tok.encode_expression("kid with a kite", 12)
    assert ids[0] == 0 and ids[mask.index(1) - 1] == 2 and set(ids[mask.index(1):]) == {1} and len(ids) == len(mask) == 12
    ids, mask = tok.encode_expression("very " * 40, 6)
    assert len(ids) == 6 and ids[0] == 0 and ids[-1] == 2 and mask == [0] * 6
    with pytest.raises(RuntimeError):
        tok.encode_expression("", 6)
    with pytest.raises(FileNotFoundError):
        XLMRTokenizer(os.path.join(root, "missing.spm"))


def test_aspect_group_sampler_batches_share_a_group_and_ranks_partition_the_epoch():
    from simvg_amd.datasets import AspectGroupSampler
    flags = np.array([0] * 11 + [1] * 21, dtype=np.uint8)
    single = AspectGroupSampler(flags, 4, seed=3)
    order = list(single)
    assert len(order) == len(single) == 12 + 24
    for i in range(0, len(order), 4):
        assert len({int(flags[j]) for j in order[i:i + 4]}) == 1                  # one aspect group per batch
    assert set(order) == set(range(32))                                           # everyone appears (some twice: padding)
    single.set_epoch(1)
    assert list(single) != order                                                  # reshuffled per epoch
    ranks = [AspectGroupSampler(flags, 4, world_size=2, rank=r, seed=3) for r in range(2)]
    a, b = list(ranks[0]), list(ranks[1])
    assert len(a) == len(b) == len(ranks[0]) == (16 + 24) // 2
    assert set(a) | set(b) == set(range(32))
    for part in (a, b):
        for i in range(0, len(part), 4):
            assert len({int(flags[j]) for j in part[i:i + 4]}) == 1


def test_two_stage_loader_runs_the_host_stage_in_seeded_workers(mini):
    """decode + tokenisation in DataLoader worker processes (host stage), the remaining transforms in the consumer; the
    expression draw of a worker follows the reference's worker seed (num_workers * rank + worker_id + seed)"""
    from simvg_amd.datasets import build_dataset, build_dataloader
    from simvg_amd.datasets.refsets import TwoStageLoader
    from oracle.mock_loop import Cfg
    fx, root = mini
    pipe = [dict(type="LoadImageAnnotationsFromFile", dataset="RefCOCOUNC", max_token=12, with_bbox=True, use_token_type="beit3",
                 spm_path=os.path.join(root, "beit3.spm"), device="cpu"),
            dict(type="CollectData", keys=["img", "ref_expr_inds", "text_attention_mask", "gt_bbox"])]
    ds = build_dataset(dict(type="RefCOCOUNC", which_set="train", img_source=["coco"], imgsfile=os.path.join(root, "coco"),
                            annsfile=os.path.join(root, "anns", "RefCOCOUNC", "instances.json"), pipeline=pipe))
    assert ds.host_steps() == 2                                # the loader and every (deferrable) transform after it
    cfg = Cfg(distributed=False, seed=5, rank=0, world_size=1, data=Cfg(samples_per_gpu=2, workers_per_gpu=2))
    loader = build_dataloader(cfg, ds)
    assert isinstance(loader, TwoStageLoader) and loader.dataset is ds and len(loader) == 2 and hasattr(loader.sampler, "set_epoch")
    loader.collate = lambda items: items                       # frames of different sizes: no stacking in this test
    seen = []
    for epoch in range(2):
        loader.sampler.set_epoch(epoch)
        for batch in loader:
            assert len(batch) == 2
            for item in batch:
                assert item["img"].dtype == torch.uint8 and item["img"].dim() == 3 and not item["img"].is_cuda
                assert len(item["ref_expr_inds"]) == 12 and item["img_metas"]["expression"]
                seen.append((os.path.basename(item["img_metas"]["filename"]), item["img_metas"]["expression"]))
    assert {n for n, _ in seen} == {"COCO_train2014_%012d.jpg" % i for i in (1, 2, 3)}
    again = []
    loader2 = build_dataloader(cfg, ds)
    loader2.collate = lambda items: items
    for epoch in range(2):
        loader2.sampler.set_epoch(epoch)
        for batch in loader2:
            again += [(os.path.basename(i["img_metas"]["filename"]), i["img_metas"]["expression"]) for i in batch]
    assert again == seen                                       # same seed -> same order and the same expression draws


def test_decode_image_applies_the_exif_orientation(tmp_path):
    """mmcv.imfrombytes(flag='color') = cv2.imdecode(IMREAD_COLOR), which rotates by the EXIF orientation tag (reference
    loading.py:157-160): a 40 x 20 JPEG tagged "rotate 90 CW" (6) must come out 40 high and 20 wide, its red left edge on top."""
    import numpy as np
    from PIL import Image
    from simvg_amd.datasets.loading import decode_image
    a = np.zeros((20, 40, 3), np.uint8)
    a[:, :8] = (255, 0, 0)                         # red band at the LEFT of the stored raster
    exif = Image.Exif()
    exif[0x0112] = 6                               # the viewer has to rotate the raster 90 degrees clockwise
    Image.fromarray(a).save(tmp_path / "o6.jpg", quality=95, exif=exif.tobytes())
    Image.fromarray(a).save(tmp_path / "plain.jpg", quality=95)
    plain, turned = decode_image(str(tmp_path / "plain.jpg")), decode_image(str(tmp_path / "o6.jpg"))
    assert plain.shape == (20, 40, 3) and turned.shape == (40, 20, 3)
    assert plain[:, :6, 2].mean() > 200 and plain[:, 12:, 2].mean() < 40          # BGR: red is channel 2
    assert turned[:6, :, 2].mean() > 200 and turned[12:, :, 2].mean() < 40       # left edge -> top edge
