"""Navier-Stokes residuals on the HIP path.  API mirror of the reference's pc/physicsConstrained.py:17-94: the stencils of
pc/grad1Filter.py and pc/grad2Filter.py (kernel_size 3 or 5, any combination) are evaluated inside the kernels, with or without the
cell-size scaling.  The trainer's loss (3x3, scaled: trainFlowParallel.py:115) runs on the fused tile kernel of tmg_phys_fwd; every
other combination on tmg_phys_fields."""
import torch
import torch.nn as nn

import tmg_hip as H


class PhysConstrainedLES(nn.Module):
    def __init__(self, dx, dy, rho=1.0, grad_kernels=[3, 3]):
        super().__init__()
        for k in grad_kernels[:2]:
            if int(k) not in (3, 5):
                raise ValueError('kernel_size size {:d} is not supported!'.format(int(k)))      # as grad1Filter.py:57 / grad2Filter.py:49
        self.k1, self.k2 = int(grad_kernels[0]), int(grad_kernels[1])
        self.rho, self.dx, self.dy = rho, dx, dy

    def _fast(self, scale):
        return scale and self.k1 == 3 and self.k2 == 3

    def _fields(self, u, p_):
        n, _, hh, ww = u.shape
        y = torch.cat([u, p_ if p_ is not None else torch.zeros((n, 1, hh, ww), device=u.device, dtype=u.dtype)], 1).contiguous()
        pstar = torch.empty((n, 1, hh, ww), device=u.device, dtype=torch.float32)
        ustar = torch.empty((n, 1, hh, ww + 2), device=u.device, dtype=torch.float32)
        H.phys_fwd(y, None, None, (1., 1., 1.), (0., 0., 0.), self.dx, self.dy, self.rho, pstar=pstar, ustar=ustar)
        return pstar, ustar

    def calcDivergence(self, uPred, scale=True):
        """[B,2,H,W] velocity -> [B,1,H,W+2] clamped divergence, dx-scaled when `scale` (first/last column replicated, reference :42-60)."""
        if self._fast(scale):
            return self._fields(uPred[:, :2], None)[1]
        u = uPred[:, :2].contiguous()
        ustar = torch.empty((u.shape[0], 1, u.shape[2], u.shape[3] + 2), device=u.device, dtype=torch.float32)
        H.phys_fields(u, None, ustar, None, self.dx, self.dy, self.rho, self.k1, self.k2, scale)
        return ustar

    def calcPressurePoisson(self, uPred, pPred, scale=True):
        """Residual of the pressure Poisson equation, dx*dy-scaled when `scale`, clamped to [-1,1] (reference :62-94)."""
        if self._fast(scale):
            return self._fields(uPred[:, :2], pPred)[0]
        u, p_ = uPred[:, :2].contiguous(), pPred.contiguous()
        pstar = torch.empty((u.shape[0], 1, u.shape[2], u.shape[3]), device=u.device, dtype=torch.float32)
        H.phys_fields(u, p_, None, pstar, self.dx, self.dy, self.rho, self.k1, self.k2, scale)
        return pstar
