"""Data-parallel gradient reduction for the MI355X hot path: one process per GPU, RCCL over xGMI through
`torch.distributed` (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).

Replaces the reference's `MMDistributedDataParallel(find_unused_parameters=True)` (tools/train.py:102-104, C2 in
SURVEY.md 2.4).  MI355X-first: gradients already live in flat fp32 arenas, so the exchange is a handful of large
contiguous all-reduces -- one per encoder layer, issued asynchronously from inside the hand-sequenced backward the
moment that layer's wgrads are written (overlapping the remaining backward), then one for the embeddings and one
for the head.  No per-tensor buckets, no unused-parameter search.

Message format: fp32 by default (the reference's DDP arithmetic).  `GradReducer(model, message_dtype="bf16")` (or
SIMVG_GRAD_MESSAGE=bf16) halves the bytes on xGMI: every message is rounded to bf16 (fp32's exponent range: no scaling
needed), averaged in bf16 by RCCL and written back into the fp32 gradient arena, which stays the master copy (the clip
norm and Adam run on fp32 as before).  Off by default: it changes the arithmetic (8 significand bits per addend) and the
fp32 exchange already hides under the backward on one node.
"""
import os

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, model, message_dtype=None):
        self.model = model
        message_dtype = message_dtype or os.environ.get("SIMVG_GRAD_MESSAGE") or None
        if message_dtype not in (None, "fp32", "bf16"):
            raise ValueError(f"message_dtype must be None / 'fp32' / 'bf16', got {message_dtype!r}")
        self.message_dtype = torch.bfloat16 if message_dtype == "bf16" else None
        self._lowp = []             # (fp32 destination, 16-bit message) pairs of the step in flight
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # SIMVG_FORCE_REDUCE=1 exercises the exchange even with a single rank (all-reduce over 1 rank == identity)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("SIMVG_FORCE_REDUCE") == "1")
        self.pending, self._scale, self._head, self._lowp = [], [], None, []
        self._all_ids, self._ids_done, self._text_rows = None, False, None
        self.last_sparse_rows = 0
        self.last_late = 0          # head gradients that missed the early message in the last step (diagnostic)
        # which branches the last step took (tests / bench line): messages sent, averaging inside the collective, ids gathered
        # with all_gather_into_tensor, rows of the sparse text-table message
        self.last_stats = {}
        self._n_msgs = 0
        # the step's message schedule, in issue order: (what, bytes) with what in {"head", "ids", "layer:<i>", "rest", "text_rows",
        # "late"} -- the overlap design of DESIGN.md section 7 as data (tests assert it: head first, layers L-1 .. 0 from inside
        # the backward, everything else after the last layer).  `timing` = True brackets finish()'s waits with two events on the
        # training stream: the time that stream stalls on RCCL = the EXPOSED (non-overlapped) part of the exchange.
        self.schedule, self.last_schedule, self._what = [], [], "rest"
        self.timing, self._exposed = False, []
        # SIMVG_DIST_CHECK=1: verify every step (one host synchronisation) that all ranks hold the same gathered id list --
        # the sparse text-row exchange is correct only then (rows are matched by position in that list)
        self.check_ids = os.environ.get("SIMVG_DIST_CHECK") == "1"
        # RCCL averages inside the collective (ncclAvg): no separate 1/world pass over the 640 MB of gradients; gloo (CPU
        # tests) has no AVG -> SUM, then one division per message
        self._nccl = dist.is_initialized() and dist.get_backend() == "nccl"
        # SIMVG_REDUCE_OP=sum: SUM in the collective + one division per message (what gloo gets).  With ONE rank RCCL's SUM is a
        # no-op while its AVG launches a pre-multiply kernel over every message (`oneRankReduce`, 2.3 ms of kernel time per step
        # beside the backward): bench.py's one-GPU overhead run uses this switch to separate the cost of ISSUING the exchange
        # (this file's) from that one-rank artefact
        self._avg = self._nccl and os.environ.get("SIMVG_REDUCE_OP", "avg") != "sum"
        self._dry = self.world == 1 and os.environ.get("SIMVG_REDUCE_DRY") == "1"
        self.enc = getattr(model, "vis_enc", None)
        self._done_layers = set()
        if self.enc is not None:
            # (an inactive reducer installs no hook; SIMVG_LAYER_HOOK=1 installs it for A/B runs of its host cost)
            self.enc._grad_ready_hook = self._on_layer_done if (self.active or os.environ.get("SIMVG_LAYER_HOOK") == "1") else None   # (=1: A/B)

    TEXT_TABLE = "beit3.text_embed.weight"

    # called by BEIT3._engine_backward after layer i (i = L-1 .. 0), then with -1 after the embedding stage
    def _on_layer_done(self, i):
        if not self.active:
            return
        # the head's backward is complete before the encoder's starts (the encoder output is upstream of every head
        # node): its gradients go first, as ONE packed message, and travel under the whole encoder backward
        self._launch_head()
        self._gather_ids()
        A = self.enc._arena
        if i >= 0:
            first = self._group_of(i)        # None: layer i's gradients wait for the lower layers of their group
            if first is not None:
                lo, hi = self._layer_span(i)[0], self._layer_span(first)[1]
                self._launch(A.flat_grad[lo:hi], f"layer:{i}" if first == i else f"layer:{first}-{i}")
            self._done_layers.add(i)
        else:
            # everything that is not a layer slice: embeddings, position tables, final LayerNorm.  The text table
            # (64 010 x D: 197 MB for ViT-B, a third of all gradients, and the LAST thing the backward produces) is
            # exchanged as the rows this step's tokens touch on ANY rank -- all other rows are zero everywhere.
            spans = sorted(self._layer_span(l) for l in range(self.enc.L))
            sparse = self._all_ids is not None and self.TEXT_TABLE in A.params
            if sparse:
                t_lo = A.offsets[self.TEXT_TABLE]
                spans = sorted(spans + [(t_lo, t_lo + A.params[self.TEXT_TABLE].numel())])
            cur = 0
            for lo, hi in spans:
                if lo > cur:
                    self._launch(A.flat_grad[cur:lo])
                cur = max(cur, hi)
            if cur < A.total:
                self._launch(A.flat_grad[cur:])
            if sparse:
                work, ids = self._all_ids
                if work is not None:
                    work.wait()
                ids = ids.reshape(-1)
                table = A.grad(self.TEXT_TABLE)
                rows = table.index_select(0, ids)          # duplicates carry the same row: harmless, no unique() / sync
                self._launch(rows, "text_rows")
                self._text_rows = (table, ids, rows)
                self.last_sparse_rows = int(ids.numel())

    def _layer_names(self, i):
        """parameters whose gradients are final when layer i's backward returns: BEIT3 keeps its LayerNorm parameters (reduced
        once, after layer 0) outside the layers' spans -- `layer_message_names`; an encoder without that method: the whole layer"""
        f = getattr(self.enc, "layer_message_names", None)
        return f(i) if f is not None else self.enc.layer_param_names(i)

    def _group_of(self, i):
        """Layers are sent in groups of consecutive layers (their Linears are one contiguous span of the arena): the message leaves
        when the group's LOWEST layer is done.  -> the group's highest layer if layer i closes a group, else None.
        SIMVG_REDUCE_GROUPS="4,4,2,1,1" (sizes in backward order, top layers first; "1,1,...": every layer its own message).
        Default (round 6): five groups of L/3, L/3, L/6, L/12, L/12 layers where 12 divides L (ViT-B: 4,4,2,1,1; ViT-L: 8,8,4,2,2) --
        large messages while most of the backward is still ahead to travel under, single layers at the end so that the exposed tail
        stays one layer's message; issuing 9 instead of 16 collectives costs a step 0.33 instead of 0.47 ms on one GPU
        (profiles/r06_reduce_ab.txt).  Any other depth: every layer its own message."""
        cache = self.__dict__.setdefault("_groups_cache", {})
        key = (self.enc.L, os.environ.get("SIMVG_REDUCE_GROUPS"))
        if key not in cache:
            L, spec = key
            if spec:
                sizes = [int(x) for x in spec.split(",")]
            elif L % 12 == 0:
                sizes = [L // 3, L // 3, L // 6, L // 12, L // 12]
            else:
                sizes = [1] * L
            if sum(sizes) != L or min(sizes) < 1:
                raise ValueError(f"SIMVG_REDUCE_GROUPS={spec!r} must list positive group sizes summing to {L} layers")
            closes, top = {}, L - 1
            for n in sizes:
                closes[top - n + 1] = top
                top -= n
            cache[key] = closes
        return cache[key].get(i)

    def _layer_span(self, i):
        """(lo, hi) of layer i's message inside the flat gradient arena; cached per arena (the name walk costs 0.1 ms of host time)"""
        A = self.enc._arena
        cache = self.__dict__.setdefault("_span_cache", {})
        if cache.get("arena") is not A:
            cache.clear()
            cache["arena"] = A
        if i not in cache:
            cache[i] = A.slice_of(self._layer_names(i))
        return cache[i]

    def _gather_ids(self):
        """token ids of every rank for this step (a few KB), gathered asynchronously when the backward starts"""
        if self._ids_done or self.enc is None:
            return
        self._ids_done = True
        ids = getattr(self.enc, "_last_ids", None)
        if ids is None or os.environ.get("SIMVG_DENSE_EMBED_REDUCE") == "1" or self.TEXT_TABLE not in self.enc._arena.params:
            return
        ids = ids.reshape(-1).contiguous()
        out = torch.empty(self.world * ids.numel(), dtype=ids.dtype, device=ids.device)
        self.schedule.append(("ids", out.numel() * out.element_size()))
        if self._nccl:     # RCCL
            work = dist.all_gather_into_tensor(out, ids, async_op=True)
        else:              # gloo (CPU tests)
            parts = [torch.empty_like(ids) for _ in range(self.world)]
            dist.all_gather(parts, ids)
            out, work = torch.cat(parts), None
        self._all_ids = (work, out)

    def _launch(self, t, what="rest"):
        if t.numel():
            self.schedule.append((what, t.numel() * (2 if self.message_dtype is not None else t.element_size())))
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            if self.message_dtype is not None:
                msg = t.to(self.message_dtype)
                self._lowp.append((t, msg))
                t = msg
            self._n_msgs += 1
            if self._dry:                # (measurement only, one rank: everything but the collective call itself)
                return
            self.pending.append(dist.all_reduce(t, op=op, async_op=True))
            if not self._avg:
                self._scale.append(t)

    def _head_params(self):
        """the parameters outside the encoder arena (walked once: the model's parameter set does not change under a reducer)"""
        ps = self.__dict__.get("_head_param_list")
        if ps is None:
            ps = self._head_param_list = [p for n, p in self.model.named_parameters() if not n.startswith("vis_enc.")]
        return ps

    def _launch_head(self):
        """The head's gradients as ONE message.  With FlatAdam the head's parameters own a flat gradient buffer
        (`core.optimizer._RestArena`, registered on the model): the autograd-produced `.grad`s are gathered into it by its one
        multi-tensor copy (the copy the optimizer would make anyway), the buffer is all-reduced IN PLACE and finish() re-points
        every `.grad` at its slice of it -- no packing `torch.cat`, no scatter back (round 4: ~110 per-tensor copies per step).
        Any other optimizer: packed copy, scattered back in finish()."""
        if self._head is not None:
            return
        arenas = getattr(self.model, "_simvg_rest_arenas", None)
        if arenas and os.environ.get("SIMVG_HEAD_MESSAGE_PACKED") != "1":
            owned = {id(p) for ra in arenas for p in ra.params}
            loose = [p for p in self._head_params() if p.grad is not None and id(p) not in owned]
            if not loose and all(ra.intact() for ra in arenas):
                sent = []
                for ra in arenas:
                    if ra.gather_grads(check=False):     # (the graded-set bookkeeping is the optimizer's: a late gradient may follow)
                        self._launch(ra.flat_grad, "head")
                        # (slice, parameter, the autograd-produced tensor, its version): re-pointed in finish(), where a tensor
                        # that autograd accumulated into after this copy is recognised by its version counter
                        sent += [(v, p, p.grad, p.grad._version) for v, p in ra.last_gathered]
                self._head = ("arena", arenas, sent)
                return
        grads = [p.grad for p in self._head_params() if p.grad is not None]
        if not grads:
            self._head = ([], None)
            return
        flat = torch.cat([g.reshape(-1) for g in grads])           # one batched copy kernel (torch.cat over a list)
        self._head = (grads, flat)
        self._launch(flat, "head")

    def _assert_same_ids(self, ids):
        """every rank must hold the SAME gathered id list, in the same order: row r of the compact message belongs to
        ids[r] on every rank.  An order-sensitive checksum is gathered from every rank and compared."""
        w = torch.arange(1, ids.numel() + 1, device=ids.device, dtype=ids.dtype)
        c = (ids * w).sum().reshape(1)                     # order-sensitive checksum (int64, wraps consistently)
        if self._nccl:
            every = torch.empty(self.world, dtype=c.dtype, device=c.device)
            dist.all_gather_into_tensor(every, c)
        else:
            parts = [torch.empty_like(c) for _ in range(self.world)]
            dist.all_gather(parts, c)
            every = torch.cat(parts)
        if bool((every != every[0]).any()):
            raise RuntimeError("GradReducer: the ranks hold different gathered token-id lists; the sparse text-row exchange "
                               "would mix rows (SIMVG_DENSE_EMBED_REDUCE=1 selects the dense message)")

    def begin(self):
        self.pending, self._scale, self._head, self._lowp = [], [], None, []
        self._all_ids, self._ids_done, self._text_rows = None, False, None
        self._n_msgs = 0
        self.schedule = []

    def exposed_ms(self):
        """mean / max time per step the training stream waited for RCCL inside finish() (`timing` = True), or None"""
        if not self._exposed:
            return None
        torch.cuda.synchronize()
        t = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed = []
        return dict(mean_ms=sum(t) / len(t), max_ms=max(t), steps=len(t))

    def _drain(self):
        """wait for every message in flight; gloo: divide by the world size; 16-bit messages: back into the fp32 gradients"""
        ev = None
        if self.timing and torch.cuda.is_available() and self.pending:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self.pending:
            w.wait()
        if ev is not None:
            ev[1].record()
            self._exposed.append(ev)
        if self.world > 1:
            for t in self._scale:
                t.div_(self.world)
        for dst, msg in self._lowp:          # 16-bit messages: back into the fp32 master gradients
            dst.copy_(msg)
        self.pending, self._scale, self._lowp = [], [], []

    def finish(self):
        """Call after loss.backward(): waits for every message, averages, scatters the head gradients back."""
        if not self.active:
            return
        self._launch_head()              # models without an encoder hook (or a frozen encoder) reduce the head here
        # safety net: a head gradient that the autograd engine accumulated only AFTER the early head message was packed
        # (none with the current graph -- every head node is upstream of the encoder's backward node -- but a silent
        # omission would desynchronise the replicas) goes in a second message
        late_arenas, late, late_flat = [], [], None
        if self._head[0] == "arena":
            # gradients that autograd produced after the gather: a parameter that had none then goes in a second message (below);
            # one that was accumulated INTO its (already travelling) slice cannot be repaired
            for v, p, g, ver in self._head[2]:
                if p.grad is not g or g._version != ver:
                    raise RuntimeError("GradReducer: a head gradient changed after the head message had left (a head node "
                                       "downstream of the encoder's backward?); SIMVG_HEAD_MESSAGE_PACKED=1 selects the packed "
                                       "message with its late-gradient pass")
                p.grad = v               # from here on the gradient IS its slice of the (all-reduced) flat buffer
            for ra in self._head[1]:
                fresh = [(v, p) for v, p in zip(ra.grad_views, ra.params) if p.grad is not None and p.grad is not v]
                if fresh:
                    late_arenas.append((ra, fresh))
            self.last_late = sum(len(f) for _, f in late_arenas)
        else:
            packed = {id(g) for g in self._head[0]}
            late = [p.grad for p in self._head_params() if p.grad is not None and id(p.grad) not in packed]
            self.last_late = len(late)
            if late:
                late_flat = torch.cat([g.reshape(-1) for g in late])
                self._launch(late_flat, "late")
        self._drain()
        if late_arenas:
            # their slices are part of the buffer that has just come back: written only now, sent as messages of their own
            for ra, fresh in late_arenas:
                torch._foreach_copy_([v for v, _ in fresh], [p.grad for _, p in fresh])
                for v, p in fresh:
                    p.grad = v
                    self._launch(v, "late")
            self._drain()
        grads, flat = self._head[:2]
        if grads == "arena":
            pass                         # reduced in place: nothing to scatter
        elif grads:
            torch._foreach_copy_(grads, [v.view_as(g) for g, v in zip(grads, flat.split([g.numel() for g in grads]))])
        if late_flat is not None:
            torch._foreach_copy_(late, [v.view_as(g) for g, v in zip(late, late_flat.split([g.numel() for g in late]))])
        if self._text_rows is not None:
            table, ids, rows = self._text_rows
            if self.check_ids:
                self._assert_same_ids(ids)
            table.index_copy_(0, ids, rows)
        self.last_stats = dict(messages=self._n_msgs, avg_in_collective=bool(self._avg),
                               ids_gathered=self._all_ids is not None, gather_into_tensor=bool(self._nccl and self._all_ids is not None),
                               sparse_rows=int(self._text_rows[1].numel()) if self._text_rows is not None else 0,
                               message_dtype="bf16" if self.message_dtype is not None else "fp32", world=self.world,
                               bytes=sum(b for _, b in self.schedule))
        self.last_schedule = list(self.schedule)
        self.pending, self._scale, self._head, self._lowp = [], [], None, []
        self._all_ids, self._ids_done, self._text_rows = None, False, None
