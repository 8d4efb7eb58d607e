"""Dump the head's two hipGraphs (forward, backward) as dot files and count node types: python tools/dev/graph_nodes.py"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch          # noqa: E402
import bench          # noqa: E402

graphs = []
orig_begin = torch.cuda.CUDAGraph.capture_begin


def begin(self, *a, **k):
    self.enable_debug_mode()
    graphs.append(self)
    return orig_begin(self, *a, **k)


torch.cuda.CUDAGraph.capture_begin = begin


def main():
    from simvg_amd.models import build_model
    from simvg_amd.graphs import train_stream
    device = torch.device("cuda", 0)
    torch.manual_seed(1234)
    model = build_model(bench.model_cfg(1, "base")).to(device).train()
    batch = bench.synthetic_batch(64, 1000, device)
    with torch.cuda.stream(train_stream(device)):
        for _ in range(6):
            losses, _ = model(batch["img"], batch["ref_expr_inds"], batch["img_metas"], return_loss=True,
                              text_attention_mask=batch["text_attention_mask"], gt_bbox=batch["gt_bbox"], rescale=False)
            for p in model.parameters():
                p.grad = None
            losses["loss_total"].backward()
    torch.cuda.synchronize()
    print("graphs captured:", len(graphs))
    for i, g in enumerate(graphs):
        path = f"/tmp/head_graph_{i}.dot"
        g.debug_dump(path)
        txt = open(path).read()
        kinds = collections.Counter(re.findall(r"(KERNEL|MEMCPY|MEMSET|EMPTY|HOST|EVENT|WAIT|RECORD|kernel|memcpy|memset)", txt))
        print(i, len(txt), dict(kinds))
        labels = collections.Counter(re.findall(r'label="([^"\\]{0,40})', txt))
        for k, v in labels.most_common(12):
            print("     ", v, k)


if __name__ == "__main__":
    main()
