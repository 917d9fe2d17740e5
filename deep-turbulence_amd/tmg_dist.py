"""Data-parallel harness for the TM-Glow path: one process per GPU, RCCL (torch.distributed backend
"nccl") over xGMI, gradient all-reduce on the optimizer step only.

Replaces the reference's single-process thread-per-GPU wrapper (utils/parallel.py:74-241): there is
no per-window parameter broadcast, no per-step LSTM-state gather and no loss gather -- replicas are
persistent, states and losses stay rank-local, and the only exchange is one bucketed all-reduce of
the gradients (mean over ranks == the reference's mean of per-GPU losses, trainFlowParallel.py:285).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available() and not os.environ.get("TMG_SINGLE_DEVICE"):
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))   # kernels launch on the current device's stream
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("TMG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_parameters(module, src=0):
    """One-time parameter / buffer broadcast so every replica starts identical."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


class GradBucket:
    """Bucketed gradient all-reduce (mean over ranks), overlapped with backward.

    The first call of `allreduce_mean()` runs synchronously after backward and learns which parameters receive gradients at
    all (the reference's dead `norm2`, SURVEY fact 8, never does: no find_unused_parameters pass, no hang).  From then on
    every live parameter carries a post-accumulate-grad hook; a bucket (parameters in reverse registration order, <=
    `bucket_mb`) owns ONE persistent flat buffer: its gradients are copied into it with one multi-tensor launch and the
    all-reduce (RCCL over xGMI with backend "nccl", async) is issued the moment its last gradient has been produced, i.e. while
    backward is still running on the earlier layers.  `allreduce_mean()` then only waits for the handles, divides each flat
    bucket by the world size (one launch) and re-binds every `p.grad` to its slice of the reduced bucket - no copy back, no
    allocation per step.  Works unchanged with gloo on CPU tensors.

    Contract: ONE backward per `allreduce_mean()` (no gradient accumulation, no retain_graph second pass) - a second gradient
    for a parameter whose bucket has already been handed to the collective raises.  The live set may change between steps: a
    parameter that first receives a gradient later triggers a rebuild of the buckets (all ranks run the same graph, so they
    rebuild together); a bucketed parameter without a gradient in some step contributes zeros.

    `measure=True`: event pairs around the exchange (see `overlap_report`)."""

    def __init__(self, params, bucket_mb=32, measure=False):
        self.params = [p for p in params if p.requires_grad]
        self.bucket_elems = int(bucket_mb * 1024 * 1024 // 4)
        self.buckets = None        # list of parameter lists once the live set is known
        self._flat = []            # persistent flat buffer per bucket
        self._views = []           # per bucket: views of the flat buffer, one per parameter
        self._where = {}           # id(p) -> bucket index
        self._pending = []         # gradients still missing per bucket in this backward
        self._work = []            # (bucket index, async handle)
        self._launched = set()
        self._hooks = []
        self.launched_during_backward = 0   # diagnostics: buckets whose all-reduce was issued from a hook
        self.rebuilds = 0
        self.measure = bool(measure)
        self._ev = []              # per step: (first bucket ready, backward done, last all-reduce done) events
        self._ev_first = None

    # ---- bucket construction (after the first backward, or when the live set grew) ----------------------------------
    def _build(self, live):
        for h in self._hooks:
            h.remove()
        self._hooks, self._where = [], {}
        order = list(reversed(live))   # backward produces the last-registered parameters' gradients first (roughly)
        self.buckets, cur, n = [], [], 0
        for p in order:
            if cur and n + p.numel() > self.bucket_elems:
                self.buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(cur)
        self._flat, self._views = [], []
        for bi, bk in enumerate(self.buckets):
            flat = torch.empty(sum(p.numel() for p in bk), device=bk[0].device, dtype=bk[0].dtype)
            views, o = [], 0
            for p in bk:
                views.append(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
                self._where[id(p)] = bi
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            self._flat.append(flat)
            self._views.append(views)
        self._reset()

    def _reset(self):
        self._pending = [len(bk) for bk in self.buckets]
        self._work = []
        self._launched = set()
        self._ev_first = None

    def _launch(self, bi):
        bk, views = self.buckets[bi], self._views[bi]
        have = [(v, p.grad) for v, p in zip(views, bk) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in zip(views, bk):
            if p.grad is None:      # no gradient this step: this rank contributes zeros (the other ranks run the same graph)
                v.zero_()
        if self.measure and self._ev_first is None and self._flat[bi].is_cuda:
            self._ev_first = torch.cuda.Event(enable_timing=True)
            self._ev_first.record()
        self._launched.add(bi)
        self._work.append((bi, dist.all_reduce(self._flat[bi], op=dist.ReduceOp.SUM, async_op=True)))

    def _on_grad(self, p):
        bi = self._where[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] < 0 or (bi in self._launched and self._pending[bi] != 0):
            raise RuntimeError("GradBucket: a second gradient arrived for a parameter of bucket %d before allreduce_mean() - one "
                               "backward per all-reduce (no gradient accumulation / retain_graph passes)" % bi)
        if self._pending[bi] == 0:
            self._launch(bi)
            self.launched_during_backward += 1

    # ---- called between backward and the optimizer step ---------------------------------------------------------------
    def allreduce_mean(self):
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return 0
        world = dist.get_world_size()
        live = [p for p in self.params if p.grad is not None]
        ev_done = None
        if self.measure and live and live[0].is_cuda:
            ev_done = torch.cuda.Event(enable_timing=True)
            ev_done.record()                       # backward has been issued up to here on the compute stream
        if self.buckets is None or any(id(p) not in self._where for p in live):
            # first step, or a parameter outside the bucket layout received a gradient: (re)build from the union - every rank runs
            # the same graph, hence sees the same live set and rebuilds in the same step
            for bi, work in self._work:            # buckets already in flight from hooks: finish them, their values are re-reduced below
                work.wait()
                # p.grad still holds this rank's OWN gradient (re-binding to the reduced buffer has not happened), so a fresh pass
                # over the new layout is exact - unless the gradients were accumulated in place into the old buffers
                if any(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for v, p in zip(self._views[bi], self.buckets[bi])):
                    raise RuntimeError("GradBucket: the set of parameters with gradients changed while gradients alias the reduced "
                                       "buckets (zero_grad(set_to_none=False)); use set_to_none=True")
            if self.buckets is not None:
                self.rebuilds += 1
            known = set(self._where)
            self._build([p for p in self.params if p.grad is not None or id(p) in known])
            for bi in range(len(self.buckets)):
                self._launch(bi)
        else:
            for bi in range(len(self.buckets)):    # buckets a hook did not complete (a parameter without gradient this time)
                if bi not in self._launched:
                    self._launch(bi)
        for bi, work in self._work:
            work.wait()
            self._flat[bi].div_(world)
            for v, p in zip(self._views[bi], self.buckets[bi]):
                p.grad = v
        if ev_done is not None and self._ev_first is not None:
            ev_last = torch.cuda.Event(enable_timing=True)
            ev_last.record()
            self._ev.append((self._ev_first, ev_done, ev_last))
        nb = len(self._work)
        self._reset()
        return nb

    def overlap_report(self):
        """Event-pair measurement of the exchange (measure=True, CUDA tensors): per step, `exchange_ms` = first bucket ready ->
        last all-reduce complete on the compute stream, `exposed_ms` = end of backward -> last all-reduce complete (what the
        step actually waits for), `overlap` = 1 - exposed / exchange.  Means over the recorded steps; clears the record."""
        if not self._ev:
            return None
        torch.cuda.synchronize()
        ex = [a.elapsed_time(c) for a, _, c in self._ev]
        xp = [max(b.elapsed_time(c), 0.0) for _, b, c in self._ev]
        n = len(self._ev)
        self._ev = []
        exch, expo = sum(ex) / n, sum(xp) / n
        return {"steps": n, "exchange_ms": round(exch, 3), "exposed_ms": round(expo, 3),
                "overlap": round(1.0 - expo / exch, 4) if exch > 0 else None, "buckets": len(self.buckets),
                "bytes": int(sum(f.numel() * f.element_size() for f in self._flat))}


def shard(t, rank, world):
    """Rank-local slice of a global batch (dim 0); the batch must divide evenly (reference parallel.py:84-86)."""
    assert t.shape[0] % world == 0, "global batch must be divisible by the number of GPUs"
    n = t.shape[0] // world
    return t[rank * n:(rank + 1) * n]


def train_window(model, optimizer, xs, states, key_states, loss_fn, bucket=None, max_grad_norm=None, sample=None):
    """One BPTT window of the reference's inner loop (trainFlowParallel.py:256-297): `tback` time-steps of the
    generative direction, one backward, gradient mean over ranks, clip, optimizer step, then the LSTM states
    are re-anchored half-way to their seed states.  `sample(model, x_t, states, t)` defaults to model.sample."""
    optimizer.zero_grad(set_to_none=True)
    loss = 0.0
    outs = []
    for t in range(len(xs)):
        if sample is None:
            y, logp, states = model.sample(xs[t], states)
        else:
            y, logp, states = sample(model, xs[t], states, t)
        loss = loss + loss_fn(y, logp)
        outs.append((y.detach(), logp.detach()))
    loss.backward()
    if bucket is not None:
        bucket.allreduce_mean()
    gn = None
    if max_grad_norm is not None:
        gn = torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], max_grad_norm)
    optimizer.step()
    states = [(0.5 * h.detach() + 0.5 * hk, 0.5 * c.detach() + 0.5 * ck) for (h, c), (hk, ck) in zip(states, key_states)]
    return loss.detach(), gn, states, outs
