"""Data-parallel gradient reduction for the MI355X hot path: one process per GPU, RCCL over xGMI through
`torch.distributed` (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).

Replaces the reference's `MMDistributedDataParallel(find_unused_parameters=True)` (tools/train.py:102-104, C2 in
SURVEY.md 2.4).  MI355X-first: gradients already live in flat fp32 arenas, so the exchange is a handful of large
contiguous all-reduces -- one per encoder layer, issued asynchronously from inside the hand-sequenced backward the
moment that layer's wgrads are written (overlapping the remaining backward), then one for the embeddings and one
for the head.  No per-tensor buckets, no unused-parameter search.
"""
import os

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, model, bucket_bytes=None):
        self.model = model
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # SIMVG_FORCE_REDUCE=1 exercises the exchange even with a single rank (all-reduce over 1 rank == identity)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("SIMVG_FORCE_REDUCE") == "1")
        self.pending, self._scale, self._head = [], [], None
        # RCCL averages inside the collective (ncclAvg): no separate 1/world pass over the 640 MB of gradients; gloo (CPU
        # tests) has no AVG -> SUM, then one division per message
        self._avg = dist.is_initialized() and dist.get_backend() == "nccl"
        self.enc = getattr(model, "vis_enc", None)
        self._done_layers = set()
        if self.enc is not None:
            self.enc._grad_ready_hook = self._on_layer_done

    # called by BEIT3._engine_backward after layer i (i = L-1 .. 0), then with -1 after the embedding stage
    def _on_layer_done(self, i):
        if not self.active:
            return
        # the head's backward is complete before the encoder's starts (the encoder output is upstream of every head
        # node): its gradients go first, as ONE packed message, and travel under the whole encoder backward
        self._launch_head()
        A = self.enc._arena
        if i >= 0:
            lo, hi = A.slice_of(self.enc.layer_param_names(i))
            self._launch(A.flat_grad[lo:hi])
            self._done_layers.add(i)
        else:
            # everything that is not a layer slice: embeddings, position tables, final LayerNorm
            spans = sorted(A.slice_of(self.enc.layer_param_names(l)) for l in range(self.enc.L))
            cur = 0
            for lo, hi in spans:
                if lo > cur:
                    self._launch(A.flat_grad[cur:lo])
                cur = max(cur, hi)
            if cur < A.total:
                self._launch(A.flat_grad[cur:])

    def _launch(self, t):
        if t.numel():
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            self.pending.append(dist.all_reduce(t, op=op, async_op=True))
            if not self._avg:
                self._scale.append(t)

    def _launch_head(self):
        if self._head is not None:
            return
        grads = [p.grad for n, p in self.model.named_parameters() if not n.startswith("vis_enc.") and p.grad is not None]
        if not grads:
            self._head = ([], None)
            return
        flat = torch.cat([g.reshape(-1) for g in grads])           # one batched copy kernel (torch.cat over a list)
        self._head = (grads, flat)
        self._launch(flat)

    def begin(self):
        self.pending, self._scale, self._head = [], [], None

    def finish(self):
        """Call after loss.backward(): waits for every message, averages, scatters the head gradients back."""
        if not self.active:
            return
        self._launch_head()              # models without an encoder hook (or a frozen encoder) reduce the head here
        for w in self.pending:
            w.wait()
        for t in self._scale:
            t.div_(self.world)
        grads, flat = self._head
        if grads:
            torch._foreach_copy_(grads, [v.view_as(g) for g, v in zip(grads, flat.split([g.numel() for g in grads]))])
        self.pending, self._scale, self._head = [], [], None
