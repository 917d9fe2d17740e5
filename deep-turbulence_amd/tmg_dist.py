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
    """Flat fp32 bucket(s) for the gradient all-reduce.  Parameters that never receive a gradient (the
    reference's dead `norm2`) are skipped, so no find_unused_parameters pass is needed."""

    def __init__(self, params, bucket_mb=32):
        self.params = [p for p in params if p.requires_grad]
        self.bucket_elems = int(bucket_mb * 1024 * 1024 // 4)

    def allreduce_mean(self):
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return 0
        world = dist.get_world_size()
        live = [p for p in self.params if p.grad is not None]
        # every rank has the same set of live grads (same graph); bucket in registration order
        i, nb = 0, 0
        while i < len(live):
            j, n = i, 0
            while j < len(live) and (n == 0 or n + live[j].numel() <= self.bucket_elems):
                n += live[j].numel()
                j += 1
            flat = torch.cat([p.grad.reshape(-1) for p in live[i:j]])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(world)
            o = 0
            for p in live[i:j]:
                k = p.numel()
                p.grad.copy_(flat[o:o + k].view_as(p.grad))
                o += k
            i = j
            nb += 1
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
