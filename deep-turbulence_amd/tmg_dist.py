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
    `bucket_mb`) is flattened with ONE multi-tensor launch and its all-reduce (RCCL over xGMI with backend "nccl", async) is
    issued the moment its last gradient has been produced, i.e. while backward is still running on the earlier layers.
    `allreduce_mean()` then only waits for the handles, divides each flat bucket by the world size (one launch) and re-binds
    every `p.grad` to its slice of the reduced bucket - no copy back.  Works unchanged with gloo on CPU tensors."""

    def __init__(self, params, bucket_mb=32):
        self.params = [p for p in params if p.requires_grad]
        self.bucket_elems = int(bucket_mb * 1024 * 1024 // 4)
        self.buckets = None        # list of parameter lists once the live set is known
        self._where = {}           # id(p) -> bucket index
        self._pending = []         # gradients still missing per bucket in this backward
        self._work = []            # (bucket index, flat tensor, async handle)
        self._hooks = []
        self.launched_during_backward = 0   # diagnostics: buckets whose all-reduce was issued from a hook

    # ---- bucket construction (after the first backward) ------------------------------------------------------------
    def _build(self, live):
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
        for bi, bk in enumerate(self.buckets):
            for p in bk:
                self._where[id(p)] = bi
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._reset()

    def _reset(self):
        self._pending = [len(bk) for bk in self.buckets]
        self._work = []

    def _launch(self, bi):
        bk = self.buckets[bi]
        flat = torch.cat([p.grad.reshape(-1) for p in bk])
        self._work.append((bi, flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)))

    def _on_grad(self, p):
        bi = self._where[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)
            self.launched_during_backward += 1

    # ---- called between backward and the optimizer step ---------------------------------------------------------------
    def allreduce_mean(self):
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return 0
        world = dist.get_world_size()
        if self.buckets is None:
            # every rank runs the same graph, hence has the same live set
            self._build([p for p in self.params if p.grad is not None])
            for bi in range(len(self.buckets)):
                self._launch(bi)
        else:
            for bi, left in enumerate(self._pending):   # buckets a hook did not complete (a parameter without gradient this time)
                if left > 0 and all(p.grad is not None for p in self.buckets[bi]):
                    self._launch(bi)
        for bi, flat, work in self._work:
            work.wait()
            flat.div_(world)
            o = 0
            for p in self.buckets[bi]:
                k = p.numel()
                p.grad = flat[o:o + k].view_as(p)
                o += k
        nb = len(self._work)
        self._reset()
        return nb


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
