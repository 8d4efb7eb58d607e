"""Golden vectors for the file-bound head of the data pipeline, produced by EXECUTING the
reference's own `LoadImageAnnotationsFromFile` (simvg/datasets/pipelines/loading.py) and vocabulary builder
(`tokenize`, simvg/datasets/utils.py:136-190) on a miniature dataset written to a temporary directory: a few JPEG frames,
one annotation json per dataset flavour (RefCOCO-style single box, ReferItGame-style file names, `Mixed` with two image
sources, GRefCOCO with per-expression box lists and no-target records) and a sentencepiece model trained here (the real
`beit3.spm` is not in this image).

    python -m oracle.make_golden_loading        -> tests/golden/loading_golden.pt

The fixture carries the DATA (json records, JPEG bytes, the sentencepiece model, seeds) and the reference's outputs, so the
test rebuilds the miniature dataset anywhere.  Leaves under the reference code: mmcv.FileClient / imfrombytes (PIL decode,
BGR), transformers 4.x XLMRobertaTokenizer (oracle/leaf.py restatement) -- parity unpinned at that boundary.
TEST INFRASTRUCTURE ONLY."""
import importlib.util
import io
import json
import os
import sys
import tempfile

import numpy as np
import torch

from . import leaf, ref_loader

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "loading_golden.pt")

CORPUS = ["the man in the red shirt", "left dog on the grass", "woman holding an umbrella near the bus",
          "second giraffe from the right", "a cup of coffee on the wooden table", "person wearing blue jeans",
          "the tallest tree behind the house", "small white car parked on left", "kid with a kite", "two people on a bench",
          "umpire behind the catcher", "plate of food closest to us", "bike leaning on the wall", "girl in pink dress"]


def jpeg_bytes(h, w, seed):
    from PIL import Image
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 3 + yy * 2 + 60 * c) % 256 for c in range(3)], -1) + rng.randint(-20, 21, size=(h, w, 3))
    buf = io.BytesIO()
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(buf, format="JPEG", quality=90)
    return buf.getvalue()


def make_dataset():
    """-> dict(images={relative path: jpeg bytes}, sets={name: dict(dataset, annsfile json, imgsfile, img_source)})"""
    images, sets = {}, {}
    frames = [(1, 60, 80), (2, 90, 64), (3, 48, 48), (7, 70, 100), (12, 64, 96)]          # (image_id, h, w)
    for iid, h, w in frames:
        images["coco/COCO_train2014_%012d.jpg" % iid] = jpeg_bytes(h, w, iid)
        images["flickr/%d.jpg" % iid] = jpeg_bytes(h, w, 100 + iid)
    def rec(iid, h, w, exprs, bbox, **kw):
        return dict(image_id=iid, height=h, width=w, expressions=exprs, bbox=bbox, **kw)
    refcoco = dict(
        train=[rec(1, 60, 80, ["The man in the RED shirt!", "guy, left-most (red)"], [10.5, 5.0, 40.0, 50.0]),
               rec(2, 90, 64, ["woman holding an umbrella/parasol near the bus"], [30.0, 20.0, 60.0, 90.0]),       # box leaves the frame
               rec(3, 48, 48, ["kid with a kite", "a 'kite' kid?", "small kid; kite: red"], [0.0, 0.0, 47.5, 47.9])],
        val=[rec(7, 70, 100, ["second giraffe from the right " + "very " * 30 + "far"], [5.0, 6.0, 50.0, 30.0])],   # truncation
        testA=[rec(12, 64, 96, ["zebra xylophone quartz"], [1.0, 2.0, 3.0, 4.0])],                                   # unseen words / pieces
        testB=[rec(1, 60, 80, ["bike leaning on the wall"], [20.0, 10.0, 30.0, 20.0])])
    sets["RefCOCOUNC"] = dict(dataset="RefCOCOUNC", anns=refcoco, imgsfile="coco", img_source=["coco"])
    sets["ReferItGameBerkeley"] = dict(dataset="ReferItGameBerkeley", imgsfile="flickr", img_source=["saiaprtc12"],
                                       anns=dict(train=[rec(2, 90, 64, ["plate of food closest to us"], [3.0, 4.0, 20.0, 30.0]),
                                                        rec(7, 70, 100, ["umpire behind the catcher", "ump"], [50.0, 10.0, 49.0, 59.0])],
                                                 val=[rec(3, 48, 48, ["girl in pink dress"], [4.0, 4.0, 10.0, 10.0])],
                                                 test=[rec(12, 64, 96, ["two people on a bench"], [0.0, 0.0, 96.0, 64.0])]))
    mixed = dict(train=[rec(1, 60, 80, ["person wearing blue jeans"], [1.0, 1.0, 10.0, 10.0], data_source="coco"),
                        rec(2, 90, 64, ["left dog on the grass"], [2.0, 2.0, 20.0, 20.0], data_source="flickr"),
                        rec(3, 48, 48, ["a cup of coffee on the wooden table"], [3.0, 3.0, 30.0, 30.0], data_source="visual-genome")],
                 val_refcoco_unc=[rec(7, 70, 100, ["the tallest tree behind the house"], [9.0, 9.0, 20.0, 20.0], data_source="coco")])
    sets["Mixed"] = dict(dataset="Mixed", anns=mixed, imgsfile={"coco": "coco", "flickr": "flickr"}, img_source=["coco", "flickr"])
    grec = dict(train=[rec(1, 60, 80, ["two people on a bench", "nobody here"],
                           [[[5.0, 5.0, 20.0, 30.0], [40.0, 8.0, 30.0, 40.0]], [[0.0, 0.0, 0.0, 0.0]]],
                           annotations=[[dict(category_id=1, id=11), dict(category_id=1, id=12)], [dict(category_id=-1, id=-1)]]),
                       rec(12, 64, 96, ["small white car parked on left"], [[[70.0, 30.0, 40.0, 40.0]]],
                           annotations=[[dict(category_id=3, id=31)]])],
                val=[rec(3, 48, 48, ["girl in pink dress", "the tallest tree"], [[[4.0, 4.0, 10.0, 10.0]], [[0.0, 0.0, 0.0, 0.0]]],
                         annotations=[[dict(category_id=1, id=5)], [dict(category_id=-1, id=-1)]])])
    sets["GRefCOCO"] = dict(dataset="GRefCOCO", anns=grec, imgsfile="coco", img_source=["coco"])
    return images, sets


def train_spm(workdir):
    import sentencepiece as spm
    corpus = os.path.join(workdir, "corpus.txt")
    with open(corpus, "w") as f:
        f.write("\n".join(CORPUS * 20))
    prefix = os.path.join(workdir, "tiny")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=prefix, vocab_size=150, model_type="unigram", character_coverage=1.0,
                                   bos_id=-1, eos_id=-1, unk_id=0, pad_id=-1, hard_vocab_limit=False, minloglevel=2)
    return open(prefix + ".model", "rb").read()


def materialise(root, images, sets, spm_bytes):
    for rel, data in images.items():
        os.makedirs(os.path.dirname(os.path.join(root, rel)), exist_ok=True)
        with open(os.path.join(root, rel), "wb") as f:
            f.write(data)
    for name, s in sets.items():
        d = os.path.join(root, "anns", name)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "instances.json"), "w") as f:
            json.dump(s["anns"], f)
    os.makedirs(os.path.join(root, "pretrain_weights"), exist_ok=True)
    with open(os.path.join(root, "pretrain_weights", "beit3.spm"), "wb") as f:
        f.write(spm_bytes)


def load_reference_loading():
    from transformers import BertTokenizer, XLMRobertaTokenizer      # noqa: F401  resolve transformers' lazy imports BEFORE the stand-in packages (timm ...) enter sys.modules
    ref_loader.load()
    mm = sys.modules["mmcv"]

    class FileClient:
        def __init__(self, backend="disk", **kw):
            assert backend == "disk"

        def get(self, path):
            with open(path, "rb") as f:
                return f.read()

    def imfrombytes(content, flag="color", backend=None):
        from PIL import Image
        assert flag == "color"
        return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(content)).convert("RGB"))[:, :, ::-1])

    mm.FileClient, mm.imfrombytes = FileClient, imfrombytes
    ref_loader._mod("mmcv.parallel", DataContainer=type("DataContainer", (), {}))
    if "simvg.datasets" not in sys.modules:
        ref_loader._pkg("simvg.datasets")
    sys.modules["simvg.datasets"].__path__ = []
    if "simvg.datasets.builder" not in sys.modules:
        ref_loader._mod("simvg.datasets.builder", PIPELINES=leaf.Registry("PIPELINES"), DATASETS=leaf.Registry("DATASETS"))
    ref_loader._pkg("simvg.datasets.pipelines")
    out = {}
    for key, name, rel in [("loading", "simvg.datasets.pipelines.loading", "simvg/datasets/pipelines/loading.py"),
                           ("utils", "simvg.datasets.utils", "simvg/datasets/utils.py")]:
        spec = importlib.util.spec_from_file_location(name, os.path.join(ref_loader.REF_ROOT, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        out[key] = m
    out["loading"].XLMRobertaTokenizer = leaf.XLMRobertaTokenizer          # the 4.x sentencepiece-backed class
    return out


def to_plain(v):
    if isinstance(v, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(v))
    if isinstance(v, list) and v and isinstance(v[0], np.ndarray):
        return [to_plain(x) for x in v]
    return v


def main():
    R = load_reference_loading()
    images, sets = make_dataset()
    cases = []
    with tempfile.TemporaryDirectory() as root:
        spm_bytes = train_spm(root)
        materialise(root, images, sets, spm_bytes)
        cwd = os.getcwd()
        os.chdir(root)                    # the reference opens "pretrain_weights/beit3.spm" relative to the working directory
        try:
            vocab, decoded = {}, {}
            for name, s in sets.items():
                annsfile = os.path.join("anns", name, "instances.json")
                anns_all = json.load(open(annsfile))
                token2idx, idx2token, word_emb = R["utils"].tokenize(annsfile, anns_all, None)
                for f in ("token_to_ix.pkl", "ix_to_token.pkl", "word_emb.npz"):       # the cache the reference writes: not part of the fixture
                    os.remove(os.path.join("anns", name, f))
                vocab[name] = token2idx
                if anns_all["train"][0].get("data_source") is not None:       # what BaseDataset.__init__ does (base.py:43-44)
                    anns_all["train"] = [a for a in anns_all["train"] if a["data_source"] in s["img_source"]]
                for token_type in ("beit3", "default"):
                    for max_token in (20, 6):
                        loader = R["loading"].LoadImageAnnotationsFromFile(dataset=s["dataset"], max_token=max_token, with_bbox=True,
                                                                            use_token_type=token_type)
                        for which_set, records in anns_all.items():
                            for idx in range(len(records)):
                                seed = 1000 * len(cases) + 7
                                np.random.seed(seed)
                                imgs = s["imgsfile"] if isinstance(s["imgsfile"], dict) else s["imgsfile"]
                                res = loader(dict(ann=json.loads(json.dumps(records[idx])), which_set=which_set, token2idx=token2idx,
                                                  imgsfile=imgs))
                                decoded.setdefault(os.path.relpath(res["filename"]), to_plain(res["img"]))
                                keep = {k: to_plain(res[k]) for k in ("filename", "img_shape", "ori_shape", "ref_expr_inds", "expression",
                                                                        "max_token", "gt_bbox", "with_bbox", "with_mask") if k in res}
                                for k in ("text_attention_mask", "target"):
                                    if k in res:
                                        keep[k] = to_plain(res[k])
                                cases.append(dict(set=name, which_set=which_set, index=idx, token_type=token_type, max_token=max_token,
                                                  seed=seed, random_ind=int(loader.random_ind), out=keep))
        finally:
            os.chdir(cwd)
    torch.save(dict(images=images, decoded=decoded, sets=sets, spm=spm_bytes, vocab=vocab, cases=cases, corpus=CORPUS), OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(cases), "cases")
    for c in cases[:4]:
        print(c["set"], c["which_set"], c["index"], c["token_type"], c["max_token"], c["out"]["expression"][:40],
              np.asarray(c["out"]["ref_expr_inds"]).tolist()[:10], c["out"].get("gt_bbox"))


if __name__ == "__main__":
    main()
