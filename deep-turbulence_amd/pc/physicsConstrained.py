"""Navier-Stokes residuals on the HIP path.  API mirror of the reference's pc/physicsConstrained.py:17-94 (the
stencils of pc/grad1Filter.py and pc/grad2Filter.py are evaluated inside the fused kernel; only the 3x3 variants the
trainer instantiates, trainFlowParallel.py:115, exist here)."""
import torch
import torch.nn as nn

import tmg_hip as H


class PhysConstrainedLES(nn.Module):
    def __init__(self, dx, dy, rho=1.0, grad_kernels=[3, 3]):
        super().__init__()
        if list(grad_kernels) != [3, 3]:
            raise NotImplementedError("only the 3x3 stencils used by the TM-Glow trainer are on this path")
        self.rho, self.dx, self.dy = rho, dx, dy

    def _fields(self, u, p_):
        n, _, hh, ww = u.shape
        y = torch.cat([u, p_ if p_ is not None else torch.zeros((n, 1, hh, ww), device=u.device, dtype=u.dtype)], 1).contiguous()
        pstar = torch.empty((n, 1, hh, ww), device=u.device, dtype=torch.float32)
        ustar = torch.empty((n, 1, hh, ww + 2), device=u.device, dtype=torch.float32)
        H.phys_fwd(y, None, None, (1., 1., 1.), (0., 0., 0.), self.dx, self.dy, self.rho, pstar=pstar, ustar=ustar)
        return pstar, ustar

    def calcDivergence(self, uPred, scale=True):
        """[B,2,H,W] velocity -> [B,1,H,W+2] clamped, dx-scaled divergence (first/last column replicated, reference :42-60)."""
        if not scale:
            raise NotImplementedError("the trainer always scales the residual")
        return self._fields(uPred[:, :2], None)[1]

    def calcPressurePoisson(self, uPred, pPred, scale=True):
        """Residual of the pressure Poisson equation, dx*dy-scaled and clamped to [-1,1] (reference :62-94)."""
        if not scale:
            raise NotImplementedError("the trainer always scales the residual")
        return self._fields(uPred[:, :2], pPred)[0]
