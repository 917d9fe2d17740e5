"""Optimizer step of the trainer (reference main.py:78: torch.optim.Adam(lr, weight_decay=1e-8, amsgrad=True)) as ONE launch.

`HipAdam` is torch.optim.Adam with `step()` replaced: same constructor, same state (`step`, `exp_avg`, `exp_avg_sq`, `max_exp_avg_sq` -
a workspace written with either loads into the other, utils/utils.py), same arithmetic in the same order (tests/test_hip_ops.py compares
three steps against torch's).  torch's multi-tensor implementations walk the model's ~1 000 small parameter tensors in ~100 launches
(1.4 ms per step at the metric configuration); the update itself moves 36 bytes per parameter (~40 us).  Optional: the trainer takes
whatever optimizer main.py constructs."""
import torch
from torch.optim.adam import adam as _torch_adam

import tmg_hip as H


def _torch_adam_signature_ok():
    """`torch.optim.adam.adam` is a private functional API: the general (non-HIP) path below calls it only if it still accepts the
    keywords this file passes."""
    import inspect
    try:
        names = set(inspect.signature(_torch_adam).parameters)
    except (TypeError, ValueError):
        return False
    return {"amsgrad", "has_complex", "beta1", "beta2", "lr", "weight_decay", "eps", "maximize", "foreach", "capturable", "differentiable",
            "fused", "grad_scale", "found_inf", "decoupled_weight_decay"} <= names


_TORCH_ADAM_OK = _torch_adam_signature_ok()


class HipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, foreach=False, fused=False)
        self._chunks = {}
        self._fast = {}

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
        for gi, group in enumerate(self.param_groups):
            if self._fast_step(gi, group):
                continue
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            beta1, beta2 = group["betas"]
            self._prealloc_state(group)
            self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            if not params:
                continue
            ok = (all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params)
                  and all(g.dtype == torch.float32 and g.device == params[0].device and not g.is_sparse for g in grads)
                  and not group.get("maximize", False) and not group.get("capturable", False) and not group.get("differentiable", False)
                  and len({float(s) for s in steps}) == 1 and len({p.device for p in params}) == 1)
            if not ok:   # anything unusual: torch's own single-tensor path on this group (same state)
                if not _TORCH_ADAM_OK:
                    raise RuntimeError("HipAdam: this parameter group needs torch's own Adam path, whose functional signature differs in "
                                       "this torch version; construct torch.optim.Adam for it instead")
                _torch_adam(params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps, amsgrad=group["amsgrad"], has_complex=False,
                                      beta1=beta1, beta2=beta2, lr=group["lr"], weight_decay=group["weight_decay"], eps=group["eps"],
                                      maximize=group.get("maximize", False), foreach=False, capturable=False, differentiable=False,
                                      fused=False, grad_scale=None, found_inf=None, decoupled_weight_decay=False)
                continue
            # first step with this set of tensors: build the launch state once.  The pointer table goes through a persistent PINNED
            # staging buffer and an asynchronous copy: a pageable host-to-device copy would make the host wait for everything enqueued
            # before it, i.e. a host-device synchronisation per step.  Parameter and state pointers are written once; only the
            # gradient column changes from step to step.
            dev = params[0].device
            amsgrad = bool(group["amsgrad"])
            n = len(params)
            host = torch.empty(5 * n, dtype=torch.int64).pin_memory()
            hv = host.numpy()
            hv[0:5 * n:5] = [p.data_ptr() for p in params]
            hv[2:5 * n:5] = [m.data_ptr() for m in exp_avgs]
            hv[3:5 * n:5] = [v.data_ptr() for v in exp_avg_sqs]
            hv[4:5 * n:5] = [m.data_ptr() for m in max_sqs] if amsgrad else 0
            chunks, nchunks = self._chunk_table([p.numel() for p in params], dev)
            self._fast[gi] = dict(params=params, states=[self.state[p] for p in params], steps=steps, host=host, hv=hv,
                                  tab=torch.empty(5 * n, dtype=torch.int64, device=dev), done=torch.cuda.Event(), chunks=chunks,
                                  nchunks=nchunks, amsgrad=amsgrad, keep=(exp_avgs, exp_avg_sqs, max_sqs), step=int(float(steps[0])),
                                  ptrs=[p.data_ptr() for p in params] + [t.data_ptr() for ts in ((exp_avgs, exp_avg_sqs, max_sqs) if amsgrad
                                                                                                else (exp_avgs, exp_avg_sqs)) for t in ts])
            ok = self._fast_step(gi, group)
            assert ok
        return loss

    def _prealloc_state(self, group):
        """First step: the moment tensors of all parameters out of ONE zero-filled allocation (torch's lazy init is a `zeros_like`
        launch per tensor: ~1 800 fills for this model).  Same keys, shapes and values as torch.optim.Adam's own state; every tensor is
        a tensor of its own on the shared storage, 256-byte aligned."""
        new = [p for p in group["params"] if p.grad is not None and len(self.state[p]) == 0]
        if len(new) < 8 or not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.device == new[0].device for p in new):
            return
        if group.get("capturable", False) or group.get("fused", False) or group.get("differentiable", False):
            return
        kinds = ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if group["amsgrad"] else ())
        offs, total = [], 0
        for p in new:
            offs.append(total)
            total += (p.numel() + 63) & ~63
        flat = torch.zeros(len(kinds) * total, dtype=torch.float32, device=new[0].device)
        store = flat.untyped_storage()
        for p, o in zip(new, offs):
            st = self.state[p]
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            for ki, kind in enumerate(kinds):
                st[kind] = torch.empty(0, dtype=torch.float32, device=p.device).set_(store, ki * total + o, p.shape, p.stride())

    def _fast_step(self, gi, group):
        """The steady-state step: everything that does not change between steps (parameter / state pointers, chunk table, pinned
        staging buffer) is cached, so the host side is one pass over the gradients (~0.3 ms for ~600 tensors instead of ~3 ms of
        list building, during which the GPU - done with backward - would sit idle).  False: not applicable, take the general path."""
        c = self._fast.get(gi)
        if c is None:
            return False
        params = c["params"]
        gp = [p for p in group["params"] if p.grad is not None]
        if (len(gp) != len(params) or any(a is not b for a, b in zip(gp, params)) or bool(group["amsgrad"]) != c["amsgrad"]
                or group.get("maximize", False) or group.get("capturable", False) or group.get("differentiable", False)):
            self._fast.pop(gi)
            return False
        grads = [p.grad for p in params]
        dev = c["tab"].device
        if any(g is None or g.is_sparse or g.dtype != torch.float32 or g.device != dev or not g.is_contiguous() for g in grads):
            self._fast.pop(gi)
            return False
        st = c["states"]
        if any(self.state.get(p) is not s for p, s in zip(params, st)):   # load_state_dict replaced the state
            self._fast.pop(gi)
            return False
        # the cached pointer table must still describe the live storage: model.to(...), `p.data = ...` or a state tensor swapped by hand
        # re-materialise storage under the same Parameter / dict objects (one cheap pass, like the gradient pointer pass below)
        kinds = ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if c["amsgrad"] else ())
        ptrs = [p.data_ptr() for p in params] + [s[k].data_ptr() for k in kinds for s in st]
        if ptrs != c["ptrs"]:
            self._fast.pop(gi)
            return False
        if int(float(c["steps"][0])) != c["step"] or c["steps"][0] is not st[0]["step"]:
            self._fast.pop(gi)
            return False
        torch._foreach_add_(c["steps"], 1)
        c["step"] += 1
        step = float(c["step"])
        beta1, beta2 = group["betas"]
        n = len(params)
        c["done"].synchronize()          # the previous step's copy has left the staging buffer (it has, long ago)
        c["hv"][1:5 * n:5] = [g.data_ptr() for g in grads]
        c["tab"].copy_(c["host"], non_blocking=True)
        c["done"].record()
        bc1 = 1.0 - beta1 ** step
        bc2s = (1.0 - beta2 ** step) ** 0.5
        H.adam_step(c["tab"], c["chunks"], c["nchunks"], group["lr"], beta1, beta2, group["eps"], group["weight_decay"], bc1, bc2s,
                    c["amsgrad"])
        # the kernel writes the parameters behind torch's back: no version counter moves, so everything cached from parameter
        # values (tmg_ops.DerivedCache) is keyed on this generation as well
        import tmg_ops
        tmg_ops.PARAM_GENERATION[0] += 1
        del grads   # (kept alive until the launch is enqueued)
        return True


def adopt(optimizer):
    """Turns a plain `torch.optim.Adam` (what the reference's main.py:78 constructs) into a HipAdam IN PLACE when its arguments allow:
    the object, its param_groups and its state stay the ones the caller, the LR scheduler and saveWorkspace hold - only `step()`
    changes (one launch instead of ~100).  Returns True when the optimizer is (now) a HipAdam.  Not adopted: subclasses, fused /
    capturable / differentiable / maximize / decoupled-weight-decay groups, parameters that are not fp32 on one HIP device."""
    import functools
    import weakref
    if isinstance(optimizer, HipAdam):
        return True
    if type(optimizer) is not torch.optim.Adam:
        return False
    for g in optimizer.param_groups:
        if g.get("fused") or g.get("capturable") or g.get("differentiable") or g.get("maximize") or g.get("decoupled_weight_decay"):
            return False
        if torch.is_tensor(g.get("lr")) or not all(p.is_cuda and p.dtype == torch.float32 for p in g["params"]):
            return False
    optimizer.__class__ = HipAdam
    optimizer._chunks, optimizer._fast = {}, {}
    optimizer._patch_step_function()        # profile / pre- / post-step hooks of torch.optim.Optimizer on the new class
    w = optimizer.__dict__.get("step")
    if w is not None and getattr(w, "_wrapped_by_lr_sched", False):
        # an LR scheduler constructed earlier has wrapped the bound step() of the OLD class (torch/optim/lr_scheduler.py,
        # patch_track_step_called): the same wrapper around the new one
        ref = weakref.ref(optimizer)
        func = HipAdam.step

        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            opt = ref()
            opt._opt_called = True
            return func.__get__(opt, opt.__class__)(*args, **kwargs)

        wrapper._wrapped_by_lr_sched = True
        optimizer.step = wrapper
    return True
