"""ActNorm on the HIP path.  API mirror of the reference's nn/modules/actNorm.py:15-85."""
import torch
import torch.nn as nn

import tmg_hip as H
import tmg_ops as ops


class ActNorm(nn.Module):
    """Per-channel affine y = w*x + b with log-det HW*sum(log|w|) (same value both directions,
    no batch dimension -- reference actNorm.py:66-67, :82-83).

    Stand-alone calls run the 1x1 channel-mix kernel with a diagonal weight; inside a coupling block
    the transform is folded into the block's invertible 1x1 convolution instead
    (flowLSTMBlock.AffineCouplingBlock), so no separate pass over the activations is made.
    """

    def __init__(self, in_features, return_logdet=True, data_init=False):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(in_features, 1, 1))
        self.bias = nn.Parameter(torch.zeros(in_features, 1, 1))
        self.data_init = data_init
        self.data_initialized = False
        self.return_logdet = return_logdet

    def _init_parameters(self, input):
        # reference actNorm.py:39-50 (never reached through TMGlow: data_init is always False)
        flat = input.detach().transpose(0, 1).reshape(input.shape[1], -1)
        std = flat.std(1) + 1e-6
        self.bias.data = -(flat.mean(1) / std).view(-1, 1, 1)
        self.weight.data = (1.0 / std).view(-1, 1, 1)

    def logdet(self, x):
        return self.weight.abs().log().sum() * (x.shape[-1] * x.shape[-2])

    def folded(self, reverse):
        """(scale, shift) vectors of the per-channel map in the requested direction."""
        w, b = self.weight.view(-1), self.bias.view(-1)
        return (1.0 / w, -b / w) if reverse else (w, b)

    def _run(self, x, reverse):
        s, t = self.folded(reverse)
        c = s.shape[0]
        y = ops.conv([H.nhwc(x)], torch.diag(s).view(c, c, 1, 1), t, ksize=1)
        return H.nchw(y)

    def forward(self, x):
        if self.data_init and not self.data_initialized:
            self._init_parameters(x)
            self.data_initialized = True
        y = self._run(x, False)
        return (y, self.logdet(x)) if self.return_logdet else y

    def reverse(self, y):
        x = self._run(y, True)
        return (x, self.logdet(y)) if self.return_logdet else x
