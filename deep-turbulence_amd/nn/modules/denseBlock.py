"""Dense blocks on the HIP path.  API mirror of the reference's nn/modules/denseBlock.py
(_DenseLayer :15-67, DenseBlock :69-100, _DenseLayerNoNorm :102-152, NoNormDenseBlock :154-185)."""
import os

import torch
import torch.nn as nn

import tmg_hip as H
import tmg_ops as ops


def drop_features(owner, yn, name='dropout'):
    """The reference's optional `nn.Dropout3d` after a conv, applied to the NCHW VIEW of the NHWC activation: the very same
    torch module, so its semantics on a 4-D input (whole samples are dropped: torch reads [N,C,H,W] as an un-batched
    [C,D,H,W]) and its random-number consumption are the reference's by construction.  Off the hot path (rate 0 everywhere
    the paper trains); an element-wise torch op, no kernel of ours."""
    drop = getattr(owner, name, None)
    if drop is None:
        return yn
    return drop(yn.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)


class _DenseLayer(nn.Sequential):
    """BN -> ReLU -> 3x3 conv (growth_rate outputs), concatenated onto the input.  BatchNorm's
    per-channel affine and the ReLU are folded into the conv kernel's input staging; the batch moments
    come from a two-pass reduction kernel.  Only the non-bottleneck branch exists (main.py:72 always
    passes bottleneck=False; the reference's bottleneck branch has a channel bug, denseBlock.py:45)."""

    def __init__(self, in_features, growth_rate, drop_rate=0., bn_size=8, bottleneck=False, padding=1):
        super().__init__()
        if bottleneck and in_features > bn_size * growth_rate:
            raise NotImplementedError("bottleneck dense layers are not on the TM-Glow path (reference denseBlock.py:37-47)")
        self.add_module('norm1', nn.BatchNorm2d(in_features))
        self.add_module('conv1', nn.Conv2d(in_features, growth_rate, kernel_size=(2 * padding + 1), stride=1, padding=padding,
                                           bias=False, padding_mode='zeros'))
        if drop_rate > 0:
            self.add_module('dropout', nn.Dropout3d(p=drop_rate))   # reference :54-55 (main.py --drop-rate)

    def grow(self, xn):
        """xn: NHWC tensor or channel-slice view; returns the growth_rate new channels (NHWC)."""
        bn = self.norm1
        if self.training or not bn.track_running_stats:
            mean, rstd, a, bsh = ops.bn_batch_stats(xn.detach(), bn)
            # a / bsh carry no autograd history: BNReLUConvFn's hand-written backward returns d(gamma), d(beta) itself
            training = True
        else:
            mean = bn.running_mean
            rstd = torch.rsqrt(bn.running_var + bn.eps)
            a = bn.weight.detach() * rstd
            bsh = bn.bias.detach() - mean * a
            training = False
        return drop_features(self, ops.BNReLUConvFn.apply(xn, bn.weight, bn.bias, self.conv1.weight, mean, rstd, a, bsh, training))

    def forward(self, x):
        xn = H.nhwc(x)
        return H.nchw(torch.cat([xn, self.grow(xn)], 3))


class DenseBlock(nn.Sequential):
    def __init__(self, num_layers, in_features, growth_rate, drop_rate, bn_size=4, bottleneck=False):
        super().__init__()
        for i in range(num_layers):
            self.add_module('denselayer%d' % (i + 1),
                            _DenseLayer(in_features + i * growth_rate, growth_rate, drop_rate=drop_rate, bn_size=bn_size,
                                        bottleneck=bottleneck))

    def run(self, xn):
        """NHWC entry: writes every layer's new channels into one pre-sized buffer instead of
        re-concatenating (the reference re-allocates the whole map per layer, denseBlock.py:67)."""
        layers = list(self._modules.values())
        c0 = xn.shape[3]
        growth = [l.conv1.weight.shape[0] for l in layers]
        if not torch.is_grad_enabled() or not (xn.requires_grad or any(p.requires_grad for p in self.parameters())):
            buf = torch.empty(xn.shape[:3] + (c0 + sum(growth),), device=xn.device, dtype=xn.dtype)
            buf[..., :c0] = xn
            c = c0
            for l, g in zip(layers, growth):
                buf[..., c:c + g] = l.grow(buf[..., :c])
                c += g
            return buf
        if all(getattr(l, 'dropout', None) is None for l in layers) and os.environ.get("TMG_NO_DENSE_BLOCK_NODE") is None:
            # training path: the same pre-sized buffer, as one autograd node (tmg_ops.DenseBlockFn)
            params = [t for l in layers for t in (l.norm1.weight, l.norm1.bias, l.conv1.weight)]
            return ops.DenseBlockFn.apply(xn, [l.norm1 for l in layers], self.training, *params)
        out = xn
        for l in layers:
            out = torch.cat([out, l.grow(out)], 3)
        return out

    def forward(self, x):
        return H.nchw(self.run(H.nhwc(x)))


class _DenseLayerNoNorm(nn.Sequential):
    """ReLU -> 3x3 conv, concatenated onto the UN-rectified input (reference :135-152)."""

    def __init__(self, in_features, growth_rate, drop_rate=0., bn_size=8, bottleneck=False, padding=1):
        super().__init__()
        if bottleneck and in_features > bn_size * growth_rate:
            raise NotImplementedError("bottleneck dense layers are not on the TM-Glow path")
        self.add_module('conv1', nn.Conv2d(in_features, growth_rate, kernel_size=(2 * padding + 1), stride=1, padding=padding,
                                           bias=False, padding_mode='zeros'))
        if drop_rate > 0:
            self.add_module('dropout', nn.Dropout3d(p=drop_rate))   # reference :139-140

    def grow(self, inputs):
        return drop_features(self, ops.conv(inputs, self.conv1.weight, relu_in=True))

    def forward(self, x):
        xn = H.nhwc(x)
        return H.nchw(torch.cat([xn, self.grow([xn])], 3))


class NoNormDenseBlock(nn.Sequential):
    def __init__(self, num_layers, in_features, growth_rate, drop_rate, bn_size=4, bottleneck=False):
        super().__init__()
        for i in range(num_layers):
            self.add_module('denselayer%d' % (i + 1),
                            _DenseLayerNoNorm(in_features + i * growth_rate, growth_rate, drop_rate=drop_rate, bn_size=bn_size,
                                              bottleneck=bottleneck))

    def run(self, inputs):
        """inputs: list of NHWC tensors (their concatenation is the block input).  Returns the list
        extended by each layer's new channels -- nothing is concatenated."""
        cur = list(inputs)
        for layer in self._modules.values():
            if len(cur) > H_MAX_SEG:
                cur = cur[:2] + [torch.cat(cur[2:], 3)]
            cur = cur + [layer.grow(cur)]
        return cur

    def forward(self, x):
        return H.nchw(torch.cat(self.run([H.nhwc(x)]), 3))


H_MAX_SEG = 3
