"""API mirror of the reference's nn/modules/misc.py."""
import torch.nn as nn

import tmg_hip as H
import tmg_ops as ops


class UpsamplingLinear(nn.Module):
    """Bilinear up-sampling with align_corners=True (reference misc.py:14-35) as a HIP gather kernel
    (forward) and its exact adjoint gather (backward)."""

    def __init__(self, scale_factor=2.):
        super().__init__()
        self.scale_factor = scale_factor

    def forward(self, x):
        return H.nchw(ops.UpsampleFn.apply(H.nhwc(x), self.scale_factor))
