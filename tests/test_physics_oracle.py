"""Pin the physics-loss oracle (oracle/physics_oracle.py) against the fixture recorded from the reference's
TMGLowLoss (tests/golden/make_golden.py loss).  CPU only."""
import numpy as np
import pytest
import torch

import common as C
from oracle import physics_oracle as PO


@pytest.mark.parametrize("tag", ["small", "clamped"])
def test_loss_and_gradients_match_reference(tag):
    d = C.load_npz("phys_loss.npz")
    t = lambda k: torch.from_numpy(d[tag + "." + k])  # noqa: E731
    y = t("y").clone().requires_grad_(True)
    logp = t("logp").clone().requires_grad_(True)
    beta, dx, dy = (float(v) for v in d[tag + ".cfg"])
    hat = t("std").view(1, 3, 1, 1) * y.detach().reshape(-1, 3, y.shape[-2], y.shape[-1]) + t("mu").view(1, 3, 1, 1)
    C.assert_field(PO.pressure_poisson(hat[:, :2], hat[:, 2:], dx, dy), d[tag + ".pstar"], "pstar", atol=1e-5)
    C.assert_field(PO.divergence(hat[:, :2], dx, dy), d[tag + ".ustar"], "ustar", atol=1e-5)
    loss = PO.tmglow_loss(y, logp, t("target"), t("tmean"), t("trms"), t("std"), t("mu"), beta, dx, dy)
    assert abs(loss.item() - float(d[tag + ".loss"])) <= 1e-5 * abs(float(d[tag + ".loss"]))
    loss.backward()
    C.assert_grads({"y": y.grad, "logp": logp.grad}, {"y": d[tag + ".dy"], "logp": d[tag + ".dlogp"]}, "loss grads",
                   global_tol=1e-5, tensor_tol=1e-4)


@pytest.mark.parametrize("k1", [3, 5])
@pytest.mark.parametrize("k2", [3, 5])
@pytest.mark.parametrize("scale", [True, False])
def test_residual_fields_match_reference_for_every_stencil_and_scaling(k1, k2, scale):
    """PhysConstrainedLES.calcDivergence / calcPressurePoisson with the 3x3 and 5x5 stencils and scale = True / False
    (reference pc/physicsConstrained.py:42-94, grad1Filter.py:37-88, grad2Filter.py:28-101): the oracle against phys_fields.npz."""
    d = C.load_npz("phys_fields.npz")
    dx, dy, rho = (float(v) for v in d["cfg"])
    tag = "k%d%d.%s" % (k1, k2, "scaled" if scale else "raw")
    au, ap = (float(v) for v in d[tag + ".amp"])
    u, p = torch.from_numpy(d["u"]), torch.from_numpy(d["p"])
    C.assert_field(PO.divergence(au * u, dx, dy, k1, scale), d[tag + ".ustar"], tag + " ustar", atol=2e-5)
    C.assert_field(PO.pressure_poisson(ap * u, ap * p, dx, dy, rho, k1, k2, scale), d[tag + ".pstar"], tag + " pstar", atol=2e-5)
    assert int(d["k7_raises"]) == 1
    with pytest.raises(ValueError):
        PO.grad1x(u[:, :1], dx, 7)
