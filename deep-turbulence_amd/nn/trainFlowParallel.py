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
