"""CPU oracle for the physics-constrained reverse-KL loss  --  TEST INFRASTRUCTURE ONLY.

Plain-PyTorch restatement of `TMGLowLoss.forward` (reference tmglow/nn/trainFlowParallel.py:121-177) and of the
residuals it calls (tmglow/pc/physicsConstrained.py:42-94 with the 3x3 / 5x5 stencils of pc/grad1Filter.py:37-57 and
pc/grad2Filter.py:28-49).  Pinned by tests/golden/phys_loss.npz and tests/golden/phys_fields.npz, recorded from the
reference itself (tests/golden/make_golden.py).  Only tests/ may import this file.
"""
import math

import torch
import torch.nn.functional as F

_G1 = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]]) / 8.0     # d/dx, smoothed central difference
_G2 = torch.tensor([[1., -2., 1.], [2., -4., 2.], [1., -2., 1.]]) / 4.0     # d2/dx2


# the 5x5 stencils as the reference writes them (grad1Filter.py:43-47, grad2Filter.py:33-37) - including the last COLUMN of the
# second-derivative stencil, which reads -1 in every row where the symmetric stencil would have -1, -2, -3, -2, -1
_G1_5 = torch.tensor([[1., -8., 0., 8., -1.], [2., -16., 0., 16., -2.], [3., -24., 0., 24., -3.], [2., -16., 0., 16., -2.],
                      [1., -8., 0., 8., -1.]]) / (9 * 12.)
_G2_5 = torch.tensor([[-1., 16., -30., 16., -1.], [-2., 32., -60., 32., -1.], [-3., 48., -90., 48., -1.], [-2., 32., -60., 32., -1.],
                      [-1., 16., -30., 16., -1.]]) / (9 * 12.)


def _stencil(order, ksize):
    if ksize not in (3, 5):
        raise ValueError('kernel_size size {:d} is not supported!'.format(ksize))     # grad1Filter.py:57, grad2Filter.py:49
    return {(1, 3): _G1, (2, 3): _G2, (1, 5): _G1_5, (2, 5): _G2_5}[(order, ksize)]


def _conv(u, k):
    """zero-padded correlation of a [N,1,H,W] field with a k x k stencil (grad1Filter.py:69-70)."""
    r = k.shape[0] // 2
    return F.conv2d(F.pad(u, (r, r, r, r)), k.to(u.dtype).view(1, 1, k.shape[0], k.shape[1]))


def grad1x(u, dx, ksize=3):
    return _conv(u, _stencil(1, ksize)) / dx


def grad1y(u, dy, ksize=3):
    return _conv(u, _stencil(1, ksize).t()) / dy


def grad2x(u, dx, ksize=3):
    return _conv(u, _stencil(2, ksize)) / dx ** 2


def grad2y(u, dy, ksize=3):
    return _conv(u, _stencil(2, ksize).t()) / dy ** 2


def divergence(u, dx, dy, k1=3, scale=True):
    """physicsConstrained.py:42-60: first/last column replicated before the stencils, result scaled by dx (scale=True) and
    clamped to [-1, 1]; the output is two columns wider than the input."""
    u = torch.cat((u[:, :, :, :1], u, u[:, :, :, -1:]), dim=-1)
    star = grad1y(u[:, 1:2], dy, k1) + grad1x(u[:, 0:1], dx, k1)
    return torch.clamp(dx * star if scale else star, -1, 1)


def pressure_poisson(u, p, dx, dy, rho=1.0, k1=3, k2=3, scale=True):
    """physicsConstrained.py:62-94."""
    ddp = (grad2x(p, dx, k2) + grad2y(p, dy, k2)) / rho
    rhs = grad1x(u[:, 0:1], dx, k1) ** 2 + 2 * grad1y(u[:, 0:1], dy, k1) * grad1x(u[:, 1:2], dx, k1) + grad1y(u[:, 1:2], dy, k1) ** 2
    return torch.clamp(dx * dy * (ddp + rhs) if scale else ddp + rhs, -1, 1)


def tmglow_loss(y_pred, logp, target, target_mean, target_rms, out_std, out_mu, beta, dx, dy):
    """TMGLowLoss.forward (trainFlowParallel.py:121-151).  y_pred/target: [B,T,3,H,W]; logp: [B,T];
    out_std/out_mu: [3].  (target_mean only enters dead code in the reference, :139.)"""
    std = out_std.view(1, 3, 1, 1)
    mu = out_mu.view(1, 3, 1, 1)
    flat = y_pred.reshape(-1, y_pred.size(-3), y_pred.size(-2), y_pred.size(-1))
    hat = std * flat + mu
    p_star = pressure_poisson(hat[:, :2], hat[:, 2:], dx, dy)
    v_pres = torch.mean(p_star[:, :, 1:-1, 1:-1] ** 2)
    u_star = divergence(hat[:, :2], dx, dy)
    v_div = torch.mean(u_star[:, :, 1:-1, 1:-1] ** 2)
    v_l1 = torch.mean((y_pred - target) ** 2)
    pred_rms = torch.sqrt(torch.mean((y_pred - torch.mean(y_pred, dim=1).unsqueeze(1)) ** 2, dim=1))
    v_rms = torch.mean((pred_rms - target_rms) ** 2)
    n_out = y_pred.size(-3) * y_pred.size(-2) * y_pred.size(-1)
    neg_entropy = logp.mean() / math.log(2.0) / n_out
    return beta * (v_pres + v_div + v_l1 + v_rms) + neg_entropy
