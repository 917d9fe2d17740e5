"""Invertible 1x1 convolutions on the HIP path.  API mirror of the reference's nn/modules/glowConv.py."""
import numpy as np
import scipy.linalg
import torch
import torch.nn as nn

import tmg_hip as H
import tmg_ops as ops


class _TriInv(torch.autograd.Function):
    """Inverse of a triangular [C,C] parameter matrix.  Forward is one triangular solve against I;
    backward is the closed form dA = -X^T G X^T as two tiny matmuls (ROCm's solve_triangular
    backward gave wrong gradients on gfx950, matmul does not)."""

    @staticmethod
    def forward(ctx, a, upper):
        eye = torch.eye(a.shape[-1], device=a.device, dtype=a.dtype)
        x = torch.linalg.solve_triangular(a.detach(), eye.expand_as(a), upper=upper)
        ctx.save_for_backward(x)
        return x

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        xt = x.transpose(-1, -2)
        return -(xt @ g @ xt), None


def _host_inverse_fp64(w):
    """General inverse for the plain (non-LU) variant, which is off the TM-Glow path: done on the host in
    fp64 as the reference does (glowConv.py:58), gradient by the closed form."""
    return _HostInv.apply(w)


class _HostInv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w):
        x = torch.inverse(w.detach().double().cpu()).float().to(w.device)
        ctx.save_for_backward(x)
        return x

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        xt = x.t()
        return -(xt @ g @ xt)


def _mix(x, w, bias=None):
    c = w.shape[0]
    return H.nchw(ops.conv([H.nhwc(x)], w.reshape(c, c, 1, 1), bias, ksize=1))


class InvertibleConv1x1(nn.Module):
    """Plain (non-LU) variant, reference glowConv.py:17-102.  Not instantiated by TMGlow; kept for
    API completeness.  The inverse / determinant are O(C^3) parameter-side work done with torch in
    fp64 exactly as the reference does; the channel mix itself is the MFMA 1x1 kernel."""

    def __init__(self, in_features, train_sampling=True):
        super().__init__()
        w_shape = (in_features, in_features)
        w_init = np.linalg.qr(np.random.randn(*w_shape))[0].astype(np.float32)
        self.w_shape = w_shape
        self.train_sampling = train_sampling
        self.weight = nn.Parameter(torch.Tensor(w_init))

    def _matrix(self, inverse):
        return _host_inverse_fp64(self.weight) if inverse else self.weight

    def log_determinant(self, x, W):
        h, w = x.shape[2:]
        # O(C^3) scalar bookkeeping on the host in fp64, as the reference (glowConv.py:98-101)
        det = torch.det(W.detach().to(torch.float64).cpu()).to(torch.float32)
        if det.item() == 0:
            det = det + 1e-6
        return (h * w * det.abs().log()).to(W.device)

    def forward(self, x):
        W = self._matrix(self.train_sampling)
        return _mix(x, W), self.log_determinant(x, W)

    def reverse(self, y):
        W = self._matrix(not self.train_sampling)
        return _mix(y, W), self.log_determinant(y, W)


class InvertibleConv1x1LU(nn.Module):
    """PLU-parameterised invertible 1x1 convolution, reference glowConv.py:105-222.

    Same parameters / buffers (l, u, log_s, p, sign_s, l_mask, u_mask, eye, log_s_old) and the
    same quirks: U carries an extra +0.01*I that the log-det ignores; with train_sampling=True
    `forward` applies W^-1, `reverse` applies W, and both report -HW*sum(log_s).
    W is rebuilt on every call (the reference's `self.W` cache, :157,209-212, cannot survive two
    backward passes; `log_s_old` is still maintained for state_dict compatibility).
    """

    def __init__(self, in_channels, train_sampling=True):
        super().__init__()
        dtype = np.float32
        w_shape = (in_channels, in_channels)
        w_init = np.linalg.qr(np.random.randn(*w_shape))[0].astype(dtype)
        self.w_shape = w_shape
        self.train_sampling = train_sampling
        p_np, l_np, u_np = scipy.linalg.lu(w_init)
        s_np = np.diag(u_np)
        self.register_buffer("p", torch.Tensor(p_np.astype(dtype)))
        self.l = nn.Parameter(torch.Tensor(l_np.astype(dtype)))
        self.u = nn.Parameter(torch.Tensor(np.triu(u_np, k=1).astype(dtype)))
        self.log_s = nn.Parameter(torch.Tensor(np.log(abs(s_np)).astype(dtype)))
        self.register_buffer("sign_s", torch.Tensor(np.sign(s_np).astype(dtype)))
        self.register_buffer("l_mask", torch.Tensor(np.tril(np.ones_like(w_init), -1)))
        self.register_buffer("u_mask", torch.Tensor(np.triu(np.ones_like(w_init), k=1)))
        self.register_buffer("eye", torch.Tensor(np.eye(*w_shape, dtype=dtype)))
        self.register_buffer("log_s_old", torch.Tensor(np.log(abs(s_np)).astype(dtype) + 1.0))

    def _factors(self):
        lower = self.l * self.l_mask + self.eye
        upper = self.u * self.u_mask + torch.diag(self.log_s.exp() * self.sign_s) + 0.01 * self.eye
        return lower, upper

    def weight(self):
        with torch.no_grad():
            self.log_s_old.copy_(self.log_s)
        lower, upper = self._factors()
        return self.p @ (lower @ upper)

    def inv_weight(self):
        """W^-1 = U^-1 L^-1 P^-1 (reference glowConv.py:171-174).  The reference takes three general
        `inverse()`s; the factors are triangular and P is a permutation, so two triangular inverses times
        P^T give the same matrix (to fp32 rounding) without a pivoted GPU factorisation."""
        lower, upper = self._factors()
        return _TriInv.apply(upper, True) @ (_TriInv.apply(lower, False) @ self.p.t())

    def matrix(self, reverse):
        """Effective channel-mix matrix of the requested direction."""
        use_weight = reverse if self.train_sampling else not reverse
        return self.weight() if use_weight else self.inv_weight()

    def logdet(self, x):
        # glowConv.py:186-190 / :207-215: the sign flips with train_sampling, not with the direction
        ld = self.log_s.sum() * (x.shape[2] * x.shape[3])
        return -ld if self.train_sampling else ld

    def forward(self, x):
        return _mix(x, self.matrix(False)), self.logdet(x)

    def reverse(self, y):
        return _mix(y, self.matrix(True)), self.logdet(y)
