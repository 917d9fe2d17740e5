"""Loss of the TM-Glow trainer on the HIP path.  API mirror of `TMGLowLoss` in the reference's
nn/trainFlowParallel.py:104-177 (the trainer loop itself is tmg_dist.train_window)."""
import math

import torch
import torch.nn as nn

import tmg_hip as H


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
        inv = torch.tensor([1.0 / c for c in cnt], device=y.device, dtype=torch.float32)
        ctx.save_for_backward(y, target, mean, coef)
        ctx.cfg = (sd, mu, beta, dx, dy, cnt, T)
        return beta * (sums * inv).sum()

    @staticmethod
    def backward(ctx, g):
        y, target, mean, coef = ctx.saved_tensors
        sd, mu, beta, dx, dy, cnt, T = ctx.cfg
        B, _, _, Hh, Ww = y.shape
        up = float(g)  # scalar upstream gradient of the loss value (host read: the trainer's loss is the graph root)
        dyo = torch.empty_like(y)
        c = [up * beta * 2.0 / n for n in cnt]
        H.phys_bwd(y.view(B * T, 3, Hh, Ww), target.view(B * T, 3, Hh, Ww), mean, coef, dyo.view(B * T, 3, Hh, Ww), T, sd, mu, dx, dy, 1.0,
                   c[0], c[1], c[2], c[3])
        return dyo, None, None, None, None, None, None, None


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

    def forward(self, yPred, logp, target, target_mean, target_rms):
        sd = [float(v) for v in self.output_std.flatten().tolist()]
        mu = [float(v) for v in self.output_mu.flatten().tolist()]
        data = _PhysLossFn.apply(yPred, target, target_rms, sd, mu, float(self.beta), float(self.dx), float(self.dy))
        n_out_pixels = yPred.size(-3) * yPred.size(-2) * yPred.size(-1)
        return data + logp.mean() / math.log(2.) / n_out_pixels


class TrainFlow(object):
    """Epoch driver with the reference's interface (trainFlowParallel.py:178-311): `TrainFlow(args, model, train_loader,
    test_loader, log)`, `trainParallel(model, optimizer, tback, epoch)` -> summed loss of the epoch.

    What is different by design: one process per GPU instead of DataParallel threads - every
    rank runs this loop on its shard of the batch, replicas are persistent (no per-window `replicate`), LSTM states stay
    rank-local (no per-step gather) and the only exchange is one bucketed gradient all-reduce (RCCL, mean) per BPTT
    window, which equals the reference's mean over per-GPU losses.  No `synchronize()` / `empty_cache()` per window."""

    def __init__(self, args, model, train_loader, test_loader, log=None):
        self.args = args
        self.trainingLoader, self.testingLoader, self.log = train_loader, test_loader, log
        core = getattr(model, "module", model)
        self.loss = TMGLowLoss(args, model).to(next(core.parameters()).device)
        self._bucket = None

    def _grad_bucket(self, core):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return None
        if self._bucket is None:
            import tmg_dist
            self._bucket = tmg_dist.GradBucket([p for p in core.parameters() if p.requires_grad])
        return self._bucket

    def trainParallel(self, model, optimizer, tback=1, epoch=0, **kwargs):
        core = getattr(model, "module", model)
        dev = next(core.parameters()).device
        core.train()
        total_loss = 0
        optimizer.zero_grad()
        for mbIdx, (input0, target0, lstm_seeds) in enumerate(self.trainingLoader):
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
                ys, lps = [], []
                for tstep in range(tback):
                    y, logp, a0 = core.sample(xin[:, tstep], a0)
                    ys.append(y)
                    lps.append(logp)
                loss = self.loss(torch.stack(ys, dim=1), torch.stack(lps, dim=1), ytarget, target0_mean, target0_rms)
                loss.backward()
                bucket = self._grad_bucket(core)
                if bucket is not None:
                    bucket.allreduce_mean()
                torch.nn.utils.clip_grad_norm_([p for p in core.parameters() if p.grad is not None], self.args.max_grad_norm)
                optimizer.step()
                optimizer.zero_grad()
                # pull the LSTM states half-way back to their seed states (reference :283-287)
                a0 = [(0.5 * h.detach() + 0.5 * hk, 0.5 * c.detach() + 0.5 * ck) for (h, c), (hk, ck) in zip(a0, aKey)]
                total_loss = total_loss + loss.detach()
            if self.log is not None and (mbIdx + 1) % 5 == 0:
                self.log.log('Train Epoch: {}; Mini-batch: {}/{} ({:.0f}%); \t Current Loss: {:.6f}'.format(
                    epoch, mbIdx, len(self.trainingLoader), 100. * mbIdx / len(self.trainingLoader), float(total_loss)))
        return total_loss
