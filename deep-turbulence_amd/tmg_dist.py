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
    if (world > 1 or os.environ.get("TMG_FORCE_DIST")) and not dist.is_initialized():
        # (TMG_FORCE_DIST=1: a process group of ONE rank - legal with RCCL on one device - so that the collective path runs through the
        # real backend on a one-GPU box: bench.py --force-bucket, tests/test_dist_gpu.py)
        if backend is None:
            backend = os.environ.get("TMG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_parameters(module, src=0, force=False):
    """One-time parameter / buffer broadcast so every replica starts identical: ONE collective per dtype over a flat copy of all
    tensors of that dtype (fp32 parameters + buffers, the int64 BatchNorm counters) instead of one per tensor (~1 000 at the metric
    configuration: each a launch + a rendezvous on the RCCL stream)."""
    if not (dist.is_initialized() and (dist.get_world_size() > 1 or force)):
        return
    groups = {}
    for t in list(module.parameters()) + list(module.buffers()):
        groups.setdefault((t.dtype, t.device), []).append(t.data)
    for (_, _), ts in groups.items():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src)
        o = 0
        outs = []
        for t in ts:
            outs.append(flat[o:o + t.numel()].view(t.shape))
            o += t.numel()
        torch._foreach_copy_(ts, outs)
    # the copies went through `.data` aliases, whose version counters are not the parameters': everything derived from parameter
    # values (tmg_ops.DerivedCache) is keyed on this generation as well
    import tmg_ops
    tmg_ops.PARAM_GENERATION[0] += 1


class GradBucket:
    """Bucketed gradient all-reduce (mean over ranks), overlapped with backward.

    Layout (static, identical on every rank by construction): ALL parameters that require a gradient, in the order in which the
    FIRST backward pass delivered their gradients (rank 0's order, broadcast; `order="reverse"`: reverse registration order), cut into
    buckets of <= `bucket_mb` (4 MB: six buckets at the metric configuration); a bucket owns ONE persistent flat buffer [gradients |
    one has-gradient flag per parameter]; the LAST bucket's buffer ends with the n + 2 floats of the control vector.  TM-Glow's
    generative direction runs the flow levels deepest-first, so its backward delivers the first level's gradients first and the
    large deep levels' late - the opposite of reverse registration order, with which no bucket could leave before the end of backward
    (one-rank RCCL group, config M, 4-MB buckets: first bucket ready 7.1 ms before the last all-reduce completes with the arrival
    order, 2.6 ms with reverse registration order; 0.1 ms of the exchange exposed either way on one rank, tools/bucket_cost.py).  The collectives of a step are ALWAYS
    bucket 0, 1, .., n-1 in that order (the last one - the first-registered parameters, whose gradients arrive last anyway - always from
    allreduce_mean(), with the control values in its tail: no collective of their own, staged through a pinned buffer), then -
    only when the REDUCED control vector says so - a second pass of the buckets it names, in order.  No rank-local decision changes
    the sequence or the sizes of the collectives, so ranks whose live sets differ (or change at different times) can neither
    dead-lock each other nor bind different gradient sets.

    The first `allreduce_mean()` runs synchronously after backward and learns which parameters receive gradients (the reference's
    dead `norm2`, SURVEY fact 8, never does).  From then on every parameter carries a post-accumulate-grad hook and a bucket is
    handed to the collective (RCCL over xGMI with backend "nccl", async) the moment its expected gradients exist and every
    earlier bucket has gone - while backward is still running on the earlier layers.  `allreduce_mean()` then issues what is
    left, the last bucket with the control vector `[late[0..n-1] | changed | aliased]` in its tail, and reads the n + 2 reduced values
    on the host (the one host read of a steady step):
      * late[b] > 0: on SOME rank a gradient arrived for bucket b after it had gone (a parameter's first-ever gradient: the live set
        grew) - EVERY rank sends bucket b again with all its gradients (round 4 re-sent it on the rank that saw it only: one collective
        more than its peers);
      * changed > 0: SOME rank's live set differs from its previous step's - every rank re-reads the reduced has-gradient flags
        (round 4 re-read them only on the rank whose own set changed: a peer whose set grew left this rank binding `grad = None` for a
        parameter the peer stepped);
      * aliased > 0 together with a late bucket: some rank's gradients live inside the flat buffers the first pass has overwritten
        (zero_grad(set_to_none=False)) - every rank raises.
    Then it divides each flat bucket by the world size (one launch) and re-binds every `p.grad` to its slice of the reduced bucket -
    no copy back, no allocation per step.  The reduced flags say which parameters had a gradient on ANY rank: the others get
    `p.grad = None`, exactly as in a single-process run (the optimizer skips them: no moment decay, no weight decay).  A parameter
    that is live somewhere but has no gradient on this rank contributes zeros.  Works unchanged with gloo on CPU.

    Host synchronisation: the reduced control vector is read on the host (a synchronisation) only in the first steps, until the live
    set has been quiet for SYNC_STEPS consecutive steps; from then on it is examined one step late (see allreduce_mean): a steady
    step contains NO host synchronisation.  TMG_BUCKET_SYNC=1 reads it in every step.

    Contract: ONE backward per `allreduce_mean()` (no gradient accumulation, no retain_graph second pass) - a second gradient for
    the same parameter raises.

    `measure=True`: event pairs around the exchange (see `overlap_report`)."""

    SYNC_STEPS = 3

    def __init__(self, params, bucket_mb=4, measure=False, force=False, defer_on_cpu=False, order="arrival"):
        """force: run the collectives also in a process group of ONE rank (the real backend on a one-GPU box).  defer_on_cpu: the
        one-step-late examination of the control vector also for CPU tensors (where there is nothing to gain: the protocol's tests)."""
        self.force = bool(force)
        self.defer_on_cpu = bool(defer_on_cpu)
        self.always_sync = bool(os.environ.get("TMG_BUCKET_SYNC"))   # A / B switch: read the control vector on the host in every step
        self._sync_left = self.SYNC_STEPS      # steps of the synchronous protocol still to run (see allreduce_mean)
        self._pending = None                   # event behind the asynchronous copy of the previous step's reduced control vector
        self.deferred_steps = 0                # diagnostics: steps whose control vector was examined one step later
        self.params = [p for p in params if p.requires_grad]
        self.order = order
        self._arrival, self._arrival_seen = [], set()
        self._order_hooks = [p.register_post_accumulate_grad_hook(self._record_arrival) for p in self.params] if order == "arrival" else []
        self.bucket_elems = int(bucket_mb * 1024 * 1024 // 4)
        self.buckets = None        # list of parameter lists (built on first use: the parameters may still move device)
        self._flat = []            # persistent flat buffer per bucket: gradients, then one flag per parameter
        self._views = []           # per bucket: views of the flat buffer, one per parameter
        self._flags = []           # per bucket: the flag tail of the flat buffer
        self._flag_src = {}        # (bucket, has-gradient pattern) -> device tensor of 0 / 1
        self._ctrl = None          # control vector [late per bucket | changed | aliased], all-reduced once per step after the buckets
        self._where = {}           # id(p) -> (bucket index, position)
        self._expected = None      # per bucket: ids of the parameters that had a gradient in the previous step (None: not learned)
        self._arrived = set()      # ids of the parameters whose gradient arrived in this backward
        self._got = []             # per bucket: expected gradients that arrived so far
        self._work = []            # (bucket index, async handle)
        self._launched = set()
        self._sent_pat = {}        # bucket -> has-gradient pattern it was sent with in this step
        self._late = set()         # buckets that received a first-ever gradient after they had gone (THIS rank's view: goes into the control vector)
        self._next = 0             # next bucket index in the fixed collective order
        self._hooks = []
        self._live_local = None    # ids with a gradient on this rank in the last step
        self._live_any = None      # ids with a gradient on ANY rank (from the reduced flags)
        self.launched_during_backward = 0   # diagnostics: buckets whose all-reduce was issued from a hook
        self.flag_reads = 0        # diagnostics: host reads of the reduced flags
        self.second_passes = 0     # diagnostics: buckets sent a second time (the live set grew behind a bucket that had gone)
        self.paused = False        # True: the hooks launch nothing (CapturedWindow: a replayed backward runs no hooks, and the recording
                                   # one must not put collectives into the graph) - allreduce_mean() issues every bucket
        self.measure = bool(measure)
        self._ev = []              # per step: (first bucket ready, backward done, last all-reduce done) events
        self._ev_first = None

    # ---- static layout ------------------------------------------------------------------------------------------------------
    def _record_arrival(self, p):
        """Order-recording hook of the FIRST backward pass (registered at construction): the bucket layout follows it."""
        if self.buckets is None and p.grad is not None and id(p) not in self._arrival_seen:
            self._arrival_seen.add(id(p))
            self._arrival.append(p)

    def _layout_order(self):
        """The parameters in the order the buckets are cut from.  "reverse": reverse registration order (what a feed-forward net's
        backward produces, roughly).  "arrival" (default): the order in which the FIRST backward pass delivered the gradients - TM-Glow's
        generative direction runs the flow levels deepest-first, so its backward delivers level 1 first and the big deep levels late,
        the opposite of reverse registration order: with that layout NO bucket could go before the end of backward - followed by the
        parameters that delivered none (reverse registration order).  Rank 0's order is broadcast (indices into the registration
        order): the layout is identical on every rank by construction."""
        if self.order != "arrival" or not self._arrival:
            return list(reversed(self.params))
        seen = self._arrival_seen
        order = list(self._arrival) + [p for p in reversed(self.params) if id(p) not in seen]
        if dist.is_initialized() and dist.get_world_size() > 1:
            pos = {id(p): i for i, p in enumerate(self.params)}
            idx = torch.tensor([pos[id(p)] for p in order], dtype=torch.int64, device=self.params[0].device)
            dist.broadcast(idx, 0)
            order = [self.params[i] for i in idx.tolist()]
        return order

    def _build(self):
        order = self._layout_order()
        for h in self._order_hooks:
            h.remove()
        self._order_hooks = []
        self.buckets, cur, n = [], [], 0
        for p in order:
            if cur and n + p.numel() > self.bucket_elems:
                self.buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(cur)
        nctrl = len(self.buckets) + 2
        for bi, bk in enumerate(self.buckets):
            ng = sum(p.numel() for p in bk)
            last = bi == len(self.buckets) - 1
            flat = torch.zeros(ng + len(bk) + (nctrl if last else 0), device=bk[0].device, dtype=bk[0].dtype)
            views, o = [], 0
            for k, p in enumerate(bk):
                views.append(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
                self._where[id(p)] = (bi, k)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            self._flat.append(flat)
            self._views.append(views)
            self._flags.append(flat[ng:ng + len(bk)])
            if last:
                self._ctrl = flat[ng + len(bk):]              # rides on the last bucket's collective
        self._ctrl_host = torch.zeros(nctrl, dtype=self._ctrl.dtype)
        self._ctrl_read = torch.zeros(nctrl, dtype=self._ctrl.dtype)
        if self._ctrl.is_cuda:
            self._ctrl_host = self._ctrl_host.pin_memory()    # (a pageable source makes the copy a synchronising one)
            self._ctrl_read = self._ctrl_read.pin_memory()
        self._reset()

    def _reset(self):
        self._got = [0] * len(self.buckets)
        self._arrived = set()
        self._work = []
        self._launched = set()
        self._late = set()
        self._sent_pat = {}
        self._next = 0
        self._ev_first = None

    def _launch(self, bi):
        bk, views = self.buckets[bi], self._views[bi]
        have = [(v, p.grad) for v, p in zip(views, bk) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        none = [v for v, p in zip(views, bk) if p.grad is None]     # no gradient here this step: zeros
        if none:
            torch._foreach_zero_(none)
        pat = tuple(p.grad is not None for p in bk)
        src = self._flag_src.get((bi, pat))
        if src is None:                            # (one upload per distinct pattern: the same every step in practice)
            src = self._flag_src[(bi, pat)] = torch.tensor([1.0 if f else 0.0 for f in pat], dtype=self._flat[bi].dtype).to(self._flat[bi].device)
        self._flags[bi].copy_(src)
        if self.measure and self._ev_first is None and self._flat[bi].is_cuda:
            self._ev_first = torch.cuda.Event(enable_timing=True)
            self._ev_first.record()
        self._launched.add(bi)
        self._sent_pat[bi] = pat
        self._work.append((bi, dist.all_reduce(self._flat[bi], op=dist.ReduceOp.SUM, async_op=True)))

    def _ready(self, bi):
        return self._expected is not None and len(self._expected[bi]) > 0 and self._got[bi] == len(self._expected[bi])

    def _on_grad(self, p):
        if self.buckets is None or self._expected is None or self.paused:
            return                      # first step (or hooks switched off): everything goes after backward
        if p.grad is None:
            # the hook also fires when the node in front handed autograd NO gradient for this parameter (tmg_ops.fused_grad_accumulation
            # collects the window's parameter gradients itself and binds them after backward): nothing has arrived yet
            return
        if id(p) in self._arrived:
            raise RuntimeError("GradBucket: a second gradient arrived for a parameter before allreduce_mean() - one backward per "
                               "all-reduce (no gradient accumulation / retain_graph passes)")
        self._arrived.add(id(p))
        bi, _ = self._where[id(p)]
        if id(p) in self._expected[bi]:
            self._got[bi] += 1
        elif bi in self._launched:
            self._late.add(bi)          # a first-ever gradient for a bucket that has gone: reported in the control vector, every rank
                                        # sends the bucket again in allreduce_mean()
        # (the last bucket never goes from a hook: it carries the control vector, which is known only after backward)
        while self._next < len(self.buckets) - 1 and self._ready(self._next) and self._next not in self._launched:
            self._launch(self._next)
            self._next += 1
            self.launched_during_backward += 1

    # ---- called between backward and the optimizer step ---------------------------------------------------------------
    def allreduce_mean(self):
        if not (dist.is_initialized() and (dist.get_world_size() > 1 or self.force)):
            return 0
        world = dist.get_world_size()
        if self.buckets is None:
            self._build()
        nbk = len(self.buckets)
        ev_done = None
        if self.measure and self.params and self.params[0].is_cuda:
            ev_done = torch.cuda.Event(enable_timing=True)
            ev_done.record()                       # backward has been issued up to here on the compute stream
        for bi in range(nbk - 1):                  # what the hooks did not hand over, in the fixed order
            if bi not in self._launched:
                self._launch(bi)
        # gradients bound after backward (tmg_ops.fused_grad_accumulation) never pass a hook: a bucket that went from a hook without one
        # of them is late too - compare every bucket's has-gradient pattern now with the one it was sent with
        for bi in range(nbk - 1):
            if self._sent_pat.get(bi) != tuple(p.grad is not None for p in self.buckets[bi]):
                self._late.add(bi)
        local = {id(p) for p in self.params if p.grad is not None}
        aliased = any(p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                      for views, bk in zip(self._views, self.buckets) for v, p in zip(views, bk))
        ctrl = [1.0 if bi in self._late else 0.0 for bi in range(nbk)]
        ctrl += [1.0 if (self._live_local is None or local != self._live_local) else 0.0, 1.0 if aliased else 0.0]
        # the control values ride in the tail of the last bucket's flat buffer (round 5 sent them as a collective of their own, from a
        # pageable host tensor: a synchronising copy and a rendezvous more per step)
        self._ctrl_host.copy_(torch.tensor(ctrl, dtype=self._ctrl_host.dtype))
        self._ctrl.copy_(self._ctrl_host, non_blocking=True)
        self._launch(nbk - 1)
        for bi, work in self._work:
            work.wait()
        # Reading the reduced control vector is a HOST SYNCHRONISATION: the host cannot run ahead into the next step while the GPU
        # finishes this one, and the next step's ~300 short launches (encoder, deepest levels: 10-20 us of GPU time each, ~20 us of
        # host time each) then reach an empty queue - measured on a one-rank RCCL group: 43.1 -> 49.7 ms per step at config M with the
        # collective itself patched OUT (tools/bucket_cost.py).  So the synchronous protocol runs only while the live set is still
        # settling (SYNC_STEPS consecutive quiet steps); afterwards the vector of step N is copied to pinned memory asynchronously and
        # examined at step N + 1 - by every rank alike, it is the REDUCED vector - where a non-zero entry (the live set changed after
        # all: never in TM-Glow, whose live set is static from the first step) sends all ranks back to the synchronous protocol.  The one
        # step in between went without the second pass: the late gradient of the newly live parameter is missing from that step's sum
        # on every rank alike (replicas stay identical), and a warning says so.
        deferred = self._sync_left <= 0 and (self._ctrl.is_cuda or self.defer_on_cpu) and not self.always_sync
        if deferred:
            if self._pending is not None:
                if self._ctrl.is_cuda:
                    self._pending.synchronize()    # recorded a whole step ago: complete unless the host is more than a step ahead
                prev = self._ctrl_read.tolist()
                if any(v > 0.0 for v in prev):
                    import warnings
                    warnings.warn("GradBucket: the set of parameters with gradients changed after it had settled (reduced control "
                                  "vector %s of the previous step); the previous step's exchange ran without a second pass, the "
                                  "synchronous protocol is back on for %d steps" % (prev, self.SYNC_STEPS))
                    self._sync_left = self.SYNC_STEPS
                    self._live_any = None          # the flags of THIS step are read below: the live set is learned again
                    deferred = False
                    self._pending = None
        if deferred:
            self._ctrl_read.copy_(self._ctrl, non_blocking=True)
            if self._ctrl.is_cuda:
                self._pending = torch.cuda.Event()
                self._pending.record()
            else:
                self._pending = True
            red = [0.0] * (nbk + 2)
            self.deferred_steps += 1
        else:
            red = self._ctrl.tolist()              # the one host read of a settling step (n + 2 floats; identical on every rank)
            self._pending = None
            quiet = not any(v > 0.0 for v in red)
            self._sync_left = self._sync_left - 1 if (quiet and self._live_any is not None) else self.SYNC_STEPS
        late_any = [bi for bi in range(nbk) if red[bi] > 0.0]
        if late_any:
            # the live set grew behind a bucket that had gone - on some rank.  p.grad still holds every rank's OWN gradient (re-binding
            # happens below), so a second pass is exact - unless the gradients were accumulated in place into the buckets the first
            # pass has just overwritten with the sum
            if red[nbk + 1] > 0.0:
                raise RuntimeError("GradBucket: the set of parameters with gradients grew while gradients alias the reduced "
                                   "buckets (zero_grad(set_to_none=False)); use set_to_none=True")
            self._work = [(bi, w) for bi, w in self._work if bi not in late_any]
            for bi in late_any:
                self._launch(bi)
                self.second_passes += 1
            for bi, work in self._work:
                if bi in late_any:
                    work.wait()
        if self._live_any is None or red[nbk] > 0.0 or late_any:
            # which parameters have a gradient on ANY rank: a host read of the reduced flags - first step and whenever ANY rank's live
            # set changed (the reduced control vector says so on every rank alike)
            self.flag_reads += 1
            self._live_any = set()
            for bk, fl in zip(self.buckets, self._flags):
                for p, f in zip(bk, fl.tolist()):
                    if f > 0.0:
                        self._live_any.add(id(p))
        self._live_local = local
        done = set()
        for bi, _ in self._work:
            if bi in done:
                continue
            done.add(bi)
            self._flat[bi].div_(world)
            for v, p in zip(self._views[bi], self.buckets[bi]):
                p.grad = v if id(p) in self._live_any else None
        # the next backward's hooks wait for the gradients this rank produced now
        self._expected = [{id(p) for p in bk if id(p) in local} for bk in self.buckets]
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


def _flat_tensors(obj, out=None):
    """The tensors of a nested list / tuple structure, depth first."""
    out = [] if out is None else out
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _flat_tensors(o, out)
    elif obj is not None:
        raise TypeError("CapturedWindow: arguments are tensors or nested lists / tuples of tensors (got %s)" % type(obj).__name__)
    return out


def _map_tensors(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(o, fn) for o in obj)
    return obj


class CapturedWindow:
    """The forward passes, the loss and the backward pass of ONE BPTT window (reference trainFlowParallel.py:256-281) recorded
    once into a hipGraph and replayed for every later window of the same shape.

    Why: a window is ~11 000 kernel launches; its T forward passes are short kernels issued from Python one by one and the host
    does not keep up with the GPU there (rocprofv3 of the trainer's window, round 4: 46.9 ms of kernels in 50.9 ms per time-step).
    A replay issues the same launches from the runtime with no Python in between.  The gradient exchange, clipping, the
    optimizer step and the state re-anchoring stay eager (a few dozen launches; any optimizer works unchanged).

        cw = CapturedWindow(model, body, example_args)     # runs `body` eagerly once (warm-up), then records it
        loss, outs = cw(*args)                              # args: same nested structure / shapes as example_args

    `body(*args) -> (loss, outs)` runs the T forward passes and returns the scalar loss and whatever tensors the caller wants
    afterwards (new states, predictions: a nested list / tuple).  `args` are copied into the graph's own input tensors before
    each replay; `loss` / `outs` are the graph's output tensors - overwritten by the next replay, so detach-and-use (or clone)
    before it.  After a replay every parameter's `.grad` is bound to the gradient of THIS window (the graph's gradient buffers are
    rewritten, not accumulated into, by each replay).

    Everything the capture touches must already be able to run without host-to-device copies or synchronisation: the grouped
    weight-gradient launches write their pointer tables with a kernel during capture (tmg_fill_i64), scratch comes from the
    graph's pool, latent draws use torch's graph-safe Philox offsets.  Parameters are read in place: an optimizer step between
    replays is seen by the next one.  The graph keeps the window's activations allocated (its private pool: ~70 GB at the metric
    shape, batch 64, T = 10).  Construction runs the body twice on the example arguments (eager warm-up, then the recording) and puts
    the module buffers (BatchNorm running statistics / counters) and the device random generator back afterwards: no gradient, no
    statistics update and no generator advance is left behind.  bucket: the GradBucket of a
    multi-GPU run - its hooks are switched off for good (the gradient
    exchange then runs after the replay instead of overlapping the backward pass)."""

    def __init__(self, model, body, example_args, warmup=1, bucket=None):
        import tmg_ops
        if bucket is not None:
            bucket.paused = True     # the exchange follows the replay, un-overlapped (see GradBucket.paused)
        # graphs of earlier EAGER passes that are still alive (the derived-tensor caches of a window keep theirs until the next
        # window rebuilds them) pin AccumulateGrad nodes created on the stream those passes ran on; a recording on another stream
        # would meet them ("AccumulateGrad node's stream does not match": a cross-stream dependency inside the capture)
        import gc
        tmg_ops.invalidate_derived(model)
        gc.collect()
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.static_in = _map_tensors(tuple(example_args), lambda t: t.detach().clone())
        self._in_flat = _flat_tensors(self.static_in)
        dev = self._in_flat[0].device
        assert dev.type == "cuda", "CapturedWindow records a hipGraph: HIP tensors only"
        saved = [p.grad for p in self.params]

        def run():
            for p in self.params:
                p.grad = None
            with tmg_ops.bptt_window() as win:
                loss, outs = body(*self.static_in)
                win.backward(loss)
            return loss, outs
        # warm-up (lazy initialisation, code objects, cached operand tables) on the stream the capture will use: autograd's
        # AccumulateGrad nodes remember the stream they were created on
        # Construction must have NO observable side effect: the warm-up and the recording both run the body for real on the example
        # arguments, and the caller then replays the same window.  Module buffers (BatchNorm running statistics and counters, the mix
        # layers' `log_s_old`) and the device random generator are put back afterwards, so that the first replayed window starts from
        # the state an eager first window starts from (round 4 left the statistics updated twice and the Philox offset advanced)
        buffers = [b for b in model.buffers()]
        buf_saved = [b.detach().clone() for b in buffers]
        rng_saved = torch.cuda.get_rng_state(dev)
        self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        import warnings
        warn_always = torch.is_warn_always_enabled()
        # EVERY exit - success, the stream-mismatch refusal below, an allocation failure on the graph pool, an op that cannot be
        # captured, an exception out of `body` - goes through the `finally` at the end: module buffers, the device generator and the
        # caller's gradients are put back, so that a caller that falls back to eager windows (TrainFlow) continues on exactly the
        # trajectory an eager run would have taken (only two of the exits restored them)
        ok = False
        try:
            torch.set_warn_always(True)         # (the autograd engine reports the mismatch below once per process otherwise)
            try:
                with warnings.catch_warnings(record=True) as caught:
                    warnings.simplefilter("always")
                    with torch.cuda.stream(self.stream):
                        for _ in range(max(int(warmup), 1)):
                            run()
            finally:
                torch.set_warn_always(warn_always)
                torch.cuda.current_stream(dev).wait_stream(self.stream)
            if any("AccumulateGrad node's stream does not match" in str(w.message) for w in caught):
                # an autograd graph of an EARLIER pass on another stream is still referenced somewhere (a loss tensor, a state with a
                # grad_fn): it keeps the parameters' AccumulateGrad nodes alive, and those run on the stream they were created on -
                # inside a recording that is a dependency on a stream outside the capture (hipStreamEndCapture then crashes the process)
                raise RuntimeError("CapturedWindow: the autograd graph of an earlier pass (on another stream) is still alive - drop "
                                   "every tensor that carries a grad_fn (loss, states, outputs) before recording a window")
            for p in self.params:
                p.grad = None
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()            # the warm-up's activations go back to the device: the graph's pool is a separate one
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                loss, outs = self._record(run)
            self.loss = loss.detach()
            self.outs = _map_tensors(outs, lambda t: t.detach())
            self._grads = [(p, p.grad) for p in self.params if p.grad is not None]
            ok = True
        finally:
            if not ok:
                self.graph = None
                tmg_ops.invalidate_derived(model)      # graphs / cached tensors of the failed pass (they may live in the graph's pool)
            with torch.no_grad():
                if buffers:
                    torch._foreach_copy_(buffers, buf_saved)
            torch.cuda.set_rng_state(rng_saved, dev)
            for p, g in zip(self.params, saved):
                p.grad = g
        self.replays = 0
        # the graph reads the parameters in place: their storage must still be where it was recorded
        self._param_ptrs = [p.data_ptr() for p in self.params]

    @staticmethod
    def _record(run):
        """The recording pass itself (a seam for the failure-path tests: tests/test_model_parity.py makes it raise)."""
        return run()

    def __call__(self, *args):
        new = _flat_tensors(tuple(args))
        if len(new) != len(self._in_flat) or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(new, self._in_flat)):
            raise ValueError("CapturedWindow: arguments differ in structure / shape from the ones the window was recorded with")
        if [p.data_ptr() for p in self.params] != self._param_ptrs:
            raise RuntimeError("CapturedWindow: a parameter's storage moved since the window was recorded (model.to(...), `p.data = ...`, "
                               "a re-created parameter): the graph reads the old memory - record a new CapturedWindow")
        pairs = [(d, s) for d, s in zip(self._in_flat, new) if d is not s]
        if pairs:
            torch._foreach_copy_([d for d, _ in pairs], [s for _, s in pairs])
        self.graph.replay()
        for p, g in self._grads:
            p.grad = g
        self.replays += 1
        return self.loss, self.outs


def window_body(model, loss_fn, sample=None):
    """The forward part of train_window as a CapturedWindow body: (xs, states) -> (loss, (new states, [(y, logp) per step]))."""
    def body(xs, states):
        loss = 0.0
        outs = []
        for t in range(len(xs)):
            if sample is None:
                y, logp, states = model.sample(xs[t], states)
            else:
                y, logp, states = sample(model, xs[t], states, t)
            loss = loss + loss_fn(y, logp)
            outs.append((y, logp))
        return loss, (states, outs)
    return body


def train_window(model, optimizer, xs, states, key_states, loss_fn, bucket=None, max_grad_norm=None, sample=None, captured=None,
                 captured_extra=()):
    """One BPTT window of the reference's inner loop (trainFlowParallel.py:256-297): `tback` time-steps of the
    generative direction, one backward, gradient mean over ranks, clip, optimizer step, then the LSTM states
    are re-anchored half-way to their seed states.  `sample(model, x_t, states, t)` defaults to model.sample.
    captured: a CapturedWindow over window_body(model, loss_fn, sample) - forward passes, loss and backward are then one hipGraph
    replay (the outputs returned are the graph's own tensors: valid until the next replay); captured_extra: further arguments of its
    body after (xs, states), e.g. injected latents."""
    import tmg_ops
    optimizer.zero_grad(set_to_none=True)
    if captured is not None:
        loss, (states, outs) = captured(list(xs), states, *captured_extra)
        outs = [(y, lp) for y, lp in outs]
    else:
        loss = 0.0
        outs = []
        # one window = T forward passes on unchanged parameters + one backward: parameter-only tensors (folded mixes, padded weights) are
        # evaluated once, the T per-time-step parameter gradients are summed by T - 1 multi-tensor adds instead of ~900 T tiny ones
        with tmg_ops.bptt_window() as win:
            for t in range(len(xs)):
                if sample is None:
                    y, logp, states = model.sample(xs[t], states)
                else:
                    y, logp, states = sample(model, xs[t], states, t)
                loss = loss + loss_fn(y, logp)
                outs.append((y.detach(), logp.detach()))
            win.backward(loss)
    if bucket is not None:
        bucket.allreduce_mean()
    gn = None
    if max_grad_norm is not None:
        gn = torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], max_grad_norm)
    optimizer.step()
    states = [(0.5 * h.detach() + 0.5 * hk, 0.5 * c.detach() + 0.5 * ck) for (h, c), (hk, ck) in zip(states, key_states)]
    return loss.detach(), gn, states, outs
