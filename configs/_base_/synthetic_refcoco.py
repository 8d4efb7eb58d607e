# RefCOCO-shaped synthetic pairs generated on the fly (no dataset files exist in this image): 640x640 image noise,
# 20 XLM-R style token ids + padding mask, one xyxy box.  Same `data` layout as the reference's dataset bases
# (train / val / testA / testB), so tools/train.py and tools/test.py walk it unchanged.
dataset = "RefCOCOUNC"
max_token = 20
img_size = 640
data = dict(
    samples_per_gpu=64,
    workers_per_gpu=0,
    train=dict(type="SyntheticRefDataset", which_set="train", length=6400, img_size=img_size, max_token=max_token, seed=1),
    val=dict(type="SyntheticRefDataset", which_set="val", length=640, img_size=img_size, max_token=max_token, seed=2),
    testA=dict(type="SyntheticRefDataset", which_set="testA", length=640, img_size=img_size, max_token=max_token, seed=3),
    testB=dict(type="SyntheticRefDataset", which_set="testB", length=640, img_size=img_size, max_token=max_token, seed=4),
)
