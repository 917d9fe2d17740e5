"""Squeeze / split / prior utilities on the HIP path.  API mirror of the reference's
nn/modules/flowUtils.py (Squeeze :18-74, CheckerSqueeze :76-145, GaussianDiag :147-209,
Conv2dZeros :211-247, LatentEncoder :249-276, Split :278-335)."""
import math

import numpy as np
import torch
import torch.nn as nn

import tmg_hip as H
import tmg_ops as ops


class Squeeze(nn.Module):
    """Glow-style reshape squeeze (reference :37-74).  Pure index shuffling that TMGlow never uses
    (squeeze_type=0 selects CheckerSqueeze); it stays a view/reshape, no kernel involved."""

    def __init__(self, factor=2):
        super().__init__()
        assert factor >= 1
        self.factor = factor

    def forward(self, x):
        f = self.factor
        if f == 1:
            return x
        B, C, Hh, Ww = x.shape
        assert Hh % f == 0 and Ww % f == 0
        x = x.reshape(-1, C, f, Hh // f, f, Ww // f).transpose(3, 4)
        return x.reshape(-1, C * f * f, Hh // f, Ww // f)

    def reverse(self, y):
        f = self.factor
        if f == 1:
            return y
        B, C, Hh, Ww = y.shape
        assert C >= f * f and C % (f * f) == 0
        y = y.reshape(-1, C // (f * f), f, f, Hh, Ww).transpose(3, 4)
        return y.reshape(-1, C // (f * f), Hh * f, Ww * f)


class CheckerSqueeze(nn.Module):
    """Checkerboard squeeze: channel block k of the output holds the pixels at (row, col) offset
    (0,0), (1,0), (1,1), (0,1) (reference :114-122).  Factor is forced to 2 as in the reference (:96)."""

    def __init__(self, factor=2):
        super().__init__()
        assert factor >= 1
        self.factor = 2

    def forward(self, x):
        assert x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
        return H.nchw(ops.CheckerFn.apply(H.nhwc(x), True))

    def reverse(self, y):
        assert y.shape[1] >= 4 and y.shape[1] % 4 == 0
        return H.nchw(ops.CheckerFn.apply(H.nhwc(y), False))


class GaussianDiag(object):
    """Diagonal Gaussian over a feature map (reference :147-209).  The log-std is clamped to [-10, ln 5] (the reference clamps in
    place at construction, :163).  On the HIP path the two maps usually arrive as ONE [B,h,w,2C] tensor (the encoder's output,
    tmGlow.py:399-403: `hz=`): the kernels read (mean | log-std) from it in place and apply the clamp themselves, so neither the
    chunk, nor the clamp, nor the re-concatenation of the two halves is ever launched; `mean` / `log_stddev` are evaluated on demand."""
    Log2PI = float(np.log(2 * np.pi))

    def __init__(self, mean=None, log_stddev=None, hz=None):
        if hz is not None:
            self._hz_map = hz                    # NHWC [B,h,w,2C], un-clamped
            self._mean = self._lsd = None
        else:
            self._hz_map = None
            self._mean = mean
            self._lsd = log_stddev.clamp(min=-10., max=math.log(5.))

    @property
    def mean(self):
        if self._mean is None:
            self._mean = H.nchw(self._hz_map[..., :self._hz_map.shape[3] // 2])
        return self._mean

    @property
    def log_stddev(self):
        if self._lsd is None:
            self._lsd = H.nchw(self._hz_map[..., self._hz_map.shape[3] // 2:]).clamp(min=-10., max=math.log(5.))
        return self._lsd

    def _hz(self):
        if self._hz_map is not None:
            return self._hz_map
        return torch.cat([H.nhwc(self._mean), H.nhwc(self._lsd)], 3)

    def likelihood(self, x):
        # element-wise map; only used for inspection in the reference (:167-179)
        return -0.5 * (GaussianDiag.Log2PI + self.log_stddev * 2. + (x - self.mean) ** 2 / (self.log_stddev * 2.).exp())

    def log_prob(self, x, return_eps=False):
        """[B] summed log-likelihood; with return_eps also eps = (x - mean) / exp(log_stddev) from the same kernel."""
        logp, eps = ops.GaussLogpFn.apply(self._hz(), H.nhwc(x), 0, ops.TOP_LIMITS, bool(return_eps))
        return (logp, H.nchw(eps)) if return_eps else logp

    def sample(self, eps=None, rng=None):
        """mean + exp(log_stddev) eps; eps None: drawn inside the kernel (rng = (nonce, site) of the model call, or a fresh nonce)."""
        if eps is None and rng is None:
            rng = (ops.latent_nonce(self._hz().device), 0)
        z, _ = ops.GaussDrawFn.apply(self._hz(), None, None if eps is None else H.nhwc(eps), rng, 0, ops.TOP_LIMITS)
        return H.nchw(z)


class Conv2dZeros(nn.Module):
    """Zero-initialised 3x3 convolution with replicate padding and a learned exp-scale
    (reference :211-247) -- one launch of the MFMA implicit-GEMM kernel with the padding rule,
    bias and exp(clamp(scale,-4,ln4)) fused."""

    def __init__(self, in_features, out_features, logscale_factor=1):
        super().__init__()
        self.conv = nn.Conv2d(in_features, out_features, kernel_size=3, stride=1, padding=0, bias=True)
        self.conv.weight.data.zero_()
        self.conv.bias.data.zero_()
        self.scale = nn.Parameter(torch.zeros(1, 1, 1, 1))
        self.logscale_factor = logscale_factor

    def run(self, inputs, relu_in=False):
        """NHWC entry: `inputs` is a list of NHWC tensors standing for their channel concatenation."""
        kappa = self.scale if self.logscale_factor == 1 else self.scale * self.logscale_factor
        return ops.conv(inputs, self.conv.weight, self.conv.bias, kappa=kappa, pad_rep=True, relu_in=relu_in)

    def forward(self, x):
        return H.nchw(self.run([H.nhwc(x)]))


class LatentEncoder(nn.Module):
    """Prior network of a split (reference :249-276): hardtanh(Conv2dZeros(x), -2, ln 5) -> (mean, log-std).
    The clip is applied inside the Gaussian kernels; `raw` returns the un-clipped conv output."""

    def __init__(self, in_features):
        super().__init__()
        self.conv2d = Conv2dZeros(in_features, in_features * 2)
        self.hardtanh = nn.Hardtanh(min_val=-2.0, max_val=np.log(5.0), inplace=False)

    def raw(self, x_nhwc):
        return self.conv2d.run([x_nhwc])

    def forward(self, x):
        h = H.nchw(self.raw(H.nhwc(x)))
        mean, log_stddev = torch.clamp(h, -2.0, math.log(5.0)).chunk(2, 1)
        return GaussianDiag(mean, log_stddev)


class Split(nn.Module):
    """Split half of the channels off into a learned Gaussian (reference :278-335)."""

    def __init__(self, in_features):
        super().__init__()
        self.latent_encoder = LatentEncoder(in_features // 2)

    def forward(self, z, return_eps=False):
        zn = H.nhwc(z)
        ch = zn.shape[3] // 2
        z1, z2 = zn[..., :ch], zn[..., ch:]
        hz = self.latent_encoder.raw(z1)
        logp, eps = ops.GaussLogpFn.apply(hz, z2, 1, ops.SPLIT_LIMITS, bool(return_eps))
        return H.nchw(z1.contiguous()), logp, (H.nchw(eps) if return_eps else None)

    def reverse(self, z1, eps=None, rng=None):
        """cat(z1, z2 = mean + exp(log-std) eps) and its log-prob; the sample is written into the second half of the result by the
        kernel (no torch.cat), eps None: drawn inside it (rng = (nonce, site) of the model call, default a fresh nonce)."""
        z1n = H.nhwc(z1)
        hz = self.latent_encoder.raw(z1n)
        if eps is None and rng is None:
            rng = (ops.latent_nonce(z1n.device), 0)
        out, logp = ops.GaussDrawFn.apply(hz, z1n, None if eps is None else H.nhwc(eps), rng, 1, ops.SPLIT_LIMITS)
        return H.nchw(out), logp
