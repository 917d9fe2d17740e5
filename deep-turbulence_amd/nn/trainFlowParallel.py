"""Trainer of TM-Glow on the HIP path.  API mirror of the reference's nn/trainFlowParallel.py: `TMGLowPredictionItem`
(:29-101), `TMGLowLoss` (:104-177) and `TrainFlow` (:180-382: `trainParallel`, `test`)."""
import math
import os

import torch
import torch.nn as nn

import tmg_hip as H
import tmg_ops as ops


_INV_COUNTS = {}


class _PhysLossFn(torch.autograd.Function):
    """beta * (vPres + vDiv + vL1 + vRMS) of trainFlowParallel.py:131-149 as three fused kernels (forward sums,
    per-pixel time statistics, backward)."""

    @staticmethod
    def forward(ctx, y, target, trms, sd, mu, beta, dx, dy):
        B, T, C, Hh, Ww = y.shape
        assert C == 3, "the physics loss is defined for (u_x, u_y, p) fields"
        y = y.contiguous()
        target = target.contiguous()
        trms = trms.contiguous()
        N = B * T
        sums = torch.zeros(4, device=y.device, dtype=torch.float32)
        H.phys_fwd(y.view(N, 3, Hh, Ww), target.view(N, 3, Hh, Ww), sums, sd, mu, dx, dy)
        mean = torch.empty((B, 3, Hh, Ww), device=y.device, dtype=torch.float32)
        coef = torch.empty((B, 3, Hh, Ww), device=y.device, dtype=torch.float32)
        H.phys_rms(y, trms, mean, coef, sums[3:4])
        cnt = (N * (Hh - 2) * (Ww - 2), N * (Hh - 2) * Ww, N * 3 * Hh * Ww, B * 3 * Hh * Ww)
        inv = _INV_COUNTS.get((cnt, y.device))
        if inv is None:      # once per shape: no host-to-device copy per window (and none for a hipGraph capture to trip over)
            inv = _INV_COUNTS[(cnt, y.device)] = torch.tensor([1.0 / c for c in cnt], dtype=torch.float32).to(y.device)
        ctx.save_for_backward(y, target, mean, coef)
        ctx.cfg = (sd, mu, beta, dx, dy, cnt, T)
        return beta * (sums * inv).sum()

    @staticmethod
    def backward(ctx, g):
        y, target, mean, coef = ctx.saved_tensors
        sd, mu, beta, dx, dy, cnt, T = ctx.cfg
        B, _, _, Hh, Ww = y.shape
        dyo = torch.empty_like(y)
        c = [beta * 2.0 / n for n in cnt]
        # the upstream gradient of the loss value stays on the device (the kernel multiplies it in): no host read-back
        H.phys_bwd(y.view(B * T, 3, Hh, Ww), target.view(B * T, 3, Hh, Ww), mean, coef, dyo.view(B * T, 3, Hh, Ww), T, sd, mu, dx, dy, 1.0,
                   c[0], c[1], c[2], c[3], upstream=g.reshape(1).to(torch.float32).contiguous())
        return dyo, None, None, None, None, None, None, None


class TMGLowPredictionItem(object):
    """Accumulates predictions / log-densities / targets over time-steps along dim 1 (reference :29-101); every field is a
    tensor or a list of per-device tensors, as in the reference."""

    def __init__(self, yPred=None, yTarget=None, logp=None, tback=1):
        self.yPred, self.yTarget, self.logp, self.tback = yPred, yTarget, logp, tback

    @staticmethod
    def unsqueeze(tensor, dim=1):
        return [t.unsqueeze(dim) for t in tensor] if isinstance(tensor, list) else tensor.unsqueeze(dim)

    @staticmethod
    def concat(tensor1, tensor2, dim=1):
        assert type(tensor1) == type(tensor2), "Tensor 1 of type {} is not the same as Tensor 2 with type {}".format(
            type(tensor1), type(tensor2))
        if isinstance(tensor1, list):
            assert len(tensor1) == len(tensor2), "List sizes of tensors are not equal."
            return [torch.cat([a, b], dim=dim) for a, b in zip(tensor1, tensor2)]
        return torch.cat([tensor1, tensor2], dim=dim)

    def add(self, yPred0, logp0, yTarget0):
        new = [self.unsqueeze(v, dim=1) for v in (yPred0, logp0, yTarget0)]
        if self.yPred is None or self.yTarget is None:
            self.yPred, self.logp, self.yTarget = new
        else:
            self.yPred, self.logp, self.yTarget = (self.concat(a, b) for a, b in zip((self.yPred, self.logp, self.yTarget), new))

    def getOutputs(self):
        if isinstance(self.yPred, list):
            return [(y, l) for y, l in zip(self.yPred, self.logp)]
        return (self.yPred, self.logp)

    def getTargets(self, *newTargets):
        return (self.yTarget,) + newTargets if len(newTargets) > 0 else self.yTarget

    def clear(self):
        self.yPred = self.logp = self.yTarget = None


class TMGLowLoss(nn.Module):
    """Reverse-KL loss with physics constraints: beta*(pressure-Poisson + divergence + L2 + RMS terms) + entropy
    (reference :121-151).  Same constructor contract: `args` carries beta, dx, dy; `model` (optionally wrapped, the
    reference reads `model.module`) carries the out_std / out_mu normalisation buffers."""

    def __init__(self, args, model, log=None):
        super().__init__()
        self.beta = args.beta
        self.dx, self.dy = args.dx, args.dy
        core = getattr(model, "module", model)
        self.register_buffer("output_std", core.out_std.detach().clone().view(1, -1, 1, 1))
        self.register_buffer("output_mu", core.out_mu.detach().clone().view(1, -1, 1, 1))
        # kernel arguments: read ONCE here (the reference also snapshots the buffers at construction, :117-118); reading them
        # per call was a device->host synchronisation inside every BPTT window
        self._sd = [float(v) for v in self.output_std.flatten().tolist()]
        self._mu = [float(v) for v in self.output_mu.flatten().tolist()]

    def forward(self, yPred, logp, target, target_mean, target_rms):
        data = _PhysLossFn.apply(yPred, target, target_rms, self._sd, self._mu, float(self.beta), float(self.dx), float(self.dy))
        n_out_pixels = yPred.size(-3) * yPred.size(-2) * yPred.size(-1)
        return data + logp.mean() / math.log(2.) / n_out_pixels


class TrainFlow(object):
    """Epoch driver with the reference's interface (trainFlowParallel.py:178-382): `TrainFlow(args, model, train_loader,
    test_loader, log)`, `trainParallel(model, optimizer, tback, epoch)` -> summed loss of the epoch, `test(model, samples,
    epoch, plot)` -> mean-squared error of the predictive mean.

    What is different by design: one process per GPU instead of DataParallel threads.  Every
    rank runs this loop on ITS SHARD of each global batch (the loaders of `DataLoaderAuto` shard themselves; any other loader's
    batches are split here along dim 0, the reference's `scatter`, parallel.py:118), replicas are persistent (no per-window
    `replicate`), LSTM states stay rank-local (no per-step gather) and the only exchange is one bucketed gradient all-reduce
    (RCCL, mean) per BPTT window, which equals the reference's mean over per-GPU losses (:285).  No `synchronize()` /
    `empty_cache()` per window.  The returned loss is the mean over ranks (one scalar all-reduce per epoch)."""

    def __init__(self, args, model, train_loader, test_loader, log=None):
        self.args = args
        self.trainingLoader, self.testingLoader, self.log = train_loader, test_loader, log
        core = getattr(model, "module", model)
        self.loss = TMGLowLoss(args, model).to(next(core.parameters()).device)
        self._bucket = None
        # hipGraph replay of the windows (tmg_dist.CapturedWindow): "on" / "off" when args.capture_window or TMG_CAPTURE_WINDOW (1 / 0)
        # says so; otherwise "auto" - on HIP devices a window shape is recorded when it comes round the SECOND time (the replay is
        # worth 5 % on every box measured, round 4) and the trainer falls back to eager windows for a shape whose recording fails
        env = os.environ.get("TMG_CAPTURE_WINDOW")
        flag = getattr(args, "capture_window", None)
        self._capture = ("on" if flag else "off") if flag is not None else ({"1": "on", "0": "off"}.get(env, "auto"))
        self._captured = {}     # window shape -> tmg_dist.CapturedWindow
        self._seen = {}         # window shape -> eager windows run so far
        self._capture_failed = {}
        self.use_hip_adam = not os.environ.get("TMG_NO_HIP_ADAM")

    def _window_body(self, core):
        """The forward passes and the loss of one BPTT window (reference trainFlowParallel.py:256-281)."""
        def body(xin, ytarget, target_mean, target_rms, a0):
            ys, lps = [], []
            for tstep in range(xin.size(1)):
                y, logp, a0 = core.sample(xin[:, tstep], a0)
                ys.append(y)
                lps.append(logp)
            return self.loss(torch.stack(ys, dim=1), torch.stack(lps, dim=1), ytarget, target_mean, target_rms), a0
        return body

    @staticmethod
    def _world():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def _grad_bucket(self, core):
        if self._world()[1] == 1:
            return None
        if self._bucket is None:
            import tmg_dist
            self._bucket = tmg_dist.GradBucket([p for p in core.parameters() if p.requires_grad])
        return self._bucket

    def trainParallel(self, model, optimizer, tback=1, epoch=0, **kwargs):
        import tmg_dist
        core = getattr(model, "module", model)
        dev = next(core.parameters()).device
        core.train()
        rank, world = self._world()
        split_here = world > 1 and getattr(self.trainingLoader, "world", 1) == 1   # loader not sharded: shard each batch here
        total_loss = 0
        if self.use_hip_adam and dev.type == "cuda":
            # main.py:78 constructs torch.optim.Adam(lr, weight_decay=1e-8, amsgrad=True): the same update as ONE launch where its
            # arguments allow (same object, same state, same state_dict: the scheduler and saveWorkspace keep their references)
            import tmg_optim
            tmg_optim.adopt(optimizer)
        optimizer.zero_grad()
        for mbIdx, (input0, target0, lstm_seeds) in enumerate(self.trainingLoader):
            if split_here:
                input0, target0, lstm_seeds = (tmg_dist.shard(t, rank, world) for t in (input0, target0, lstm_seeds))
            input0, target0 = input0.to(dev), target0.to(dev)
            aKey = core.initLSTMStates(lstm_seeds.cpu(), [target0.size(-2), target0.size(-1)])
            a0 = [(h.clone(), c.clone()) for h, c in aKey]
            tmax = target0.size(1)
            tback = 10  # the reference overrides its own argument with this constant (trainFlowParallel.py:229)
            tback = min(tback, tmax)
            target0_mean = target0.mean(dim=1)
            target0_rms = torch.sqrt(((target0 - target0_mean.unsqueeze(1)) ** 2).mean(dim=1))
            for i in range(tmax // tback):
                xin, ytarget = input0[:, i * tback:(i + 1) * tback], target0[:, i * tback:(i + 1) * tback]
                bucket = self._grad_bucket(core)
                cw = None
                if self._capture != "off" and dev.type == "cuda":
                    # forward passes + loss + backward of the window as one hipGraph replay (tmg_dist.CapturedWindow), recorded once
                    # per window shape
                    key = (tuple(xin.shape), tuple(ytarget.shape))
                    cw = self._captured.get(key)
                    if cw is None and key not in self._capture_failed and (self._capture == "on" or (self._seen.get(key, 0) >= 1 and len(self._captured) < 2)):
                        try:
                            cw = self._captured[key] = tmg_dist.CapturedWindow(core, self._window_body(core), (xin, ytarget, target0_mean, target0_rms, a0),
                                                                               bucket=bucket)
                        except Exception as e:  # noqa: BLE001  (out of memory for the graph's pool, an op that cannot be captured)
                            if self._capture == "on":
                                raise
                            self._capture_failed[key] = repr(e)
                            for p_ in core.parameters():
                                p_.grad = None
                            torch.cuda.synchronize(dev)
                            torch.cuda.empty_cache()
                            if bucket is not None:
                                bucket.paused = False
                            if self.log is not None:
                                self.log.warning("window %s stays eager, recording it as a hipGraph failed: %s" % (key, e))
                    self._seen[key] = self._seen.get(key, 0) + 1
                if cw is not None:
                    loss, a0 = cw(xin, ytarget, target0_mean, target0_rms, a0)
                else:
                    # one window = `tback` forward passes on unchanged parameters + ONE backward: parameter-only tensors (folded mixes,
                    # padded weights) are evaluated once, and the per-time-step parameter gradients of the custom nodes are summed with a
                    # few multi-tensor launches instead of ~900 one-block adds per time-step (tmg_ops.bptt_window)
                    with ops.bptt_window() as win:
                        loss, a0 = self._window_body(core)(xin, ytarget, target0_mean, target0_rms, a0)
                        win.backward(loss)
                if bucket is not None:
                    bucket.allreduce_mean()
                torch.nn.utils.clip_grad_norm_([p for p in core.parameters() if p.grad is not None], self.args.max_grad_norm)
                optimizer.step()
                optimizer.zero_grad()
                # pull the LSTM states half-way back to their seed states (reference :283-287)
                a0 = [(0.5 * h.detach() + 0.5 * hk, 0.5 * c.detach() + 0.5 * ck) for (h, c), (hk, ck) in zip(a0, aKey)]
                total_loss = total_loss + loss.detach()
                loss = None      # (a tensor with a grad_fn keeps the window's autograd nodes - and the parameters' AccumulateGrad nodes, bound
                                 # to the stream they were created on - alive: the next window may be RECORDED, on another stream)
            if self.log is not None and (mbIdx + 1) % 5 == 0:
                self.log.log('Train Epoch: {}; Mini-batch: {}/{} ({:.0f}%); \t Current Loss: {:.6f}'.format(
                    epoch, mbIdx, len(self.trainingLoader), 100. * mbIdx / len(self.trainingLoader), float(total_loss)))
        if world > 1 and torch.is_tensor(total_loss):
            import torch.distributed as dist
            dist.all_reduce(total_loss, op=dist.ReduceOp.SUM)
            total_loss = total_loss / world
        return total_loss

    def test(self, model, samples=1, epoch=0, plot=True, tmax=40):
        """Prediction error on the testing loader (reference :313-382): `samples` independent roll-outs of tmax+1 time-steps
        per batch from freshly drawn LSTM seeds (states pulled half-way back to their seed states every 10 steps), fields
        un-normalised with out_std / out_mu; returns sum((mean over samples - target)^2) over steps 1..tmax divided by
        ntest * tmax * H * W.  `tmax` is clipped to the series length (the reference hard-codes 40 and needs >= 41 steps).
        Plotting (`utils.viz`, matplotlib) is outside this path: panels are drawn only if that module can be imported."""
        core = getattr(model, "module", model)
        dev = next(core.parameters()).device
        was_training = core.training
        core.eval()
        out_std = core.out_std.to(dev).view(1, 1, -1, 1, 1)
        out_mu = core.out_mu.to(dev).view(1, 1, -1, 1, 1)
        total = torch.zeros((), device=dev)
        hw, tdiv = 1, tmax
        with torch.no_grad():
            for mbIdx, (input0, target0, u0) in enumerate(self.testingLoader):
                inp = input0.to(dev)
                ytarget = out_std * target0.to(dev) + out_mu
                steps = min(tmax, inp.size(1) - 1)
                yPred = torch.zeros((samples,) + tuple(ytarget.shape), device=dev, dtype=ytarget.dtype)
                for i in range(samples):
                    seeds = torch.LongTensor(inp.size(0)).random_(0, int(1e8))
                    aKey = core.initLSTMStates(seeds, [ytarget.size(-2), ytarget.size(-1)], cache=False)   # fresh seeds every batch: never resident
                    a0 = [(h.clone(), c.clone()) for h, c in aKey]
                    for tstep in range(steps + 1):
                        yPred0, logp, a0 = core.sample(inp[:, tstep], a0)
                        yPred[i, :, tstep] = out_std[:, 0] * yPred0 + out_mu[:, 0]
                        if tstep % 10 == 0:
                            a0 = [(0.5 * h + 0.5 * hk, 0.5 * c + 0.5 * ck) for (h, c), (hk, ck) in zip(a0, aKey)]
                if plot and mbIdx == 0:
                    try:
                        from utils.viz import plotVelocityPred
                        for bidx in range(min(2, inp.size(0))):
                            plotVelocityPred(self.args, inp, yPred, ytarget, bidx=bidx, stride=4, epoch=epoch)
                    except Exception as e:  # noqa: BLE001  (no matplotlib / no reference viz module on this path)
                        if self.log is not None:
                            self.log.warning('Skipping prediction plots: {}'.format(e))
                total = total + ((yPred[:, :, 1:steps + 1].mean(dim=0) - ytarget[:, 1:steps + 1]) ** 2).sum()
                hw = yPred.size(-2) * yPred.size(-1)
                tdiv = steps
        core.train(was_training)
        return total / (self.args.ntest * max(tdiv, 1) * hw)
