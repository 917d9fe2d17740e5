"""Optimizer step of the trainer (reference main.py:78: torch.optim.Adam(lr, weight_decay=1e-8, amsgrad=True)) as ONE launch.

`HipAdam` is torch.optim.Adam with `step()` replaced: same constructor, same state (`step`, `exp_avg`, `exp_avg_sq`, `max_exp_avg_sq` -
a workspace written with either loads into the other, utils/utils.py), same arithmetic in the same order (tests/test_hip_ops.py compares
three steps against torch's).  torch's multi-tensor implementations walk the model's ~1 000 small parameter tensors in ~100 launches
(1.4 ms per step at the metric configuration); the update itself moves 36 bytes per parameter (~40 us).  Optional: the trainer takes
whatever optimizer main.py constructs."""
import torch
from torch.optim.adam import adam as _torch_adam

import tmg_hip as H


class HipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, foreach=False, fused=False)
        self._chunks = {}
        self._stage = {}

    def _chunk_table(self, numels, device):
        key = (tuple(numels), device.index)
        t = self._chunks.get(key)
        if t is None:
            rows = []
            for i, n in enumerate(numels):
                for e0 in range(0, n, 4096):
                    rows.append((i, e0, min(4096, n - e0)))
            t = (torch.tensor(rows, dtype=torch.int32).to(device), len(rows))
            self._chunks = {key: t}
        return t

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            beta1, beta2 = group["betas"]
            self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            if not params:
                continue
            ok = (all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params)
                  and not group.get("maximize", False) and not group.get("capturable", False) and not group.get("differentiable", False)
                  and len({float(s) for s in steps[:1] + steps[-1:]}) == 1)
            if not ok:   # anything unusual: torch's own single-tensor path on this group (same state)
                _torch_adam(params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps, amsgrad=group["amsgrad"], has_complex=False,
                                      beta1=beta1, beta2=beta2, lr=group["lr"], weight_decay=group["weight_decay"], eps=group["eps"],
                                      maximize=group.get("maximize", False), foreach=False, capturable=False, differentiable=False,
                                      fused=False, grad_scale=None, found_inf=None, decoupled_weight_decay=False)
                continue
            torch._foreach_add_(steps, 1)
            step = float(steps[0])
            grads = [g if g.is_contiguous() else g.contiguous() for g in grads]
            dev = params[0].device
            amsgrad = bool(group["amsgrad"])
            # pointer table through a persistent PINNED staging buffer and an asynchronous copy: a pageable host-to-device copy would
            # make the host wait for everything enqueued before it, i.e. a host-device synchronisation per step
            n = len(params)
            stage = self._stage.get(dev.index)
            if stage is None or stage[0].numel() < 5 * n:
                stage = (torch.empty(5 * n, dtype=torch.int64).pin_memory(), torch.empty(5 * n, dtype=torch.int64, device=dev),
                         torch.cuda.Event())
                self._stage[dev.index] = stage
            host, tab, done = stage
            done.synchronize()          # the previous step's copy has left the staging buffer (it has, long ago)
            hv = host.numpy()
            hv[0:5 * n:5] = [p.data_ptr() for p in params]
            hv[1:5 * n:5] = [g.data_ptr() for g in grads]
            hv[2:5 * n:5] = [m.data_ptr() for m in exp_avgs]
            hv[3:5 * n:5] = [v.data_ptr() for v in exp_avg_sqs]
            hv[4:5 * n:5] = [m.data_ptr() for m in max_sqs] if amsgrad else 0
            tab.copy_(host, non_blocking=True)
            done.record()
            chunks, nchunks = self._chunk_table([p.numel() for p in params], dev)
            bc1 = 1.0 - beta1 ** step
            bc2s = (1.0 - beta2 ** step) ** 0.5
            H.adam_step(tab, chunks, nchunks, group["lr"], beta1, beta2, group["eps"], group["weight_decay"], bc1, bc2s, amsgrad)
            del grads   # (kept alive until the launch is enqueued)
        return loss
