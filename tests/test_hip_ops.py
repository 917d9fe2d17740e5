"""GPU parity of every HIP primitive against a plain PyTorch fp32/fp64 CPU reference of the same op.
Runs through the C ABI (ctypes -> libtmglow_hip.so)."""
import math

import pytest
import torch
import torch.nn.functional as F

import common as C  # noqa: F401  (sets sys.path)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def _back(t):
    return t.detach().cpu().permute(0, 3, 1, 2)


def _close(a, b, tol=2e-5, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, "%s: max err %.3e (scale %.2e)" % (what, err, scale)


CONV_CASES = [
    # (B, H, W, seg channels, Cout, k, stride, relu_in, pad_rep, bias, kappa, relu_out)
    (2, 8, 8, [6], 5, 3, 1, False, False, False, False, False),
    (2, 16, 12, [4, 8, 4], 16, 3, 1, True, True, True, True, False),
    (1, 9, 7, [3, 5, 2], 7, 3, 1, True, True, True, True, False),
    (2, 32, 32, [8, 32, 4], 16, 3, 1, True, True, True, True, False),
    (2, 16, 16, [16, 32, 64], 256, 3, 1, False, False, True, False, False),
    (2, 16, 16, [20, 64], 40, 3, 1, False, False, True, False, True),
    (2, 8, 8, [64, 32, 4], 128, 3, 1, True, True, True, True, False),
    (2, 40, 72, [8, 4], 16, 3, 1, True, True, True, True, False),     # several tiles per image, partial tiles on two sides
    (3, 1, 8, [8], 16, 3, 1, True, True, True, False, False),         # one-row image: top and bottom ring fold onto the same pixels
    (2, 5, 1, [4], 8, 3, 1, True, True, False, False, False),
    (2, 16, 16, [8], 16, 3, 2, True, False, False, False, False),
    (3, 10, 6, [5], 9, 3, 2, True, False, False, False, False),
    (2, 16, 16, [16], 16, 1, 1, False, False, True, False, False),
    (2, 8, 12, [6, 6], 12, 1, 1, False, False, True, False, False),
    (1, 4, 4, [128], 128, 1, 1, False, False, False, False, False),
    (2, 12, 20, [24], 96, 3, 1, False, False, True, False, False),
    (1, 6, 6, [160], 96, 3, 1, False, False, True, False, True),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_and_grads(case):
    import tmg_ops as ops
    B, Hh, Ww, segs, Cout, k, stride, relu_in, pad_rep, has_b, has_k, relu_out = case
    g = torch.Generator().manual_seed(sum(segs) * 7 + Cout + Hh)
    xs = [torch.randn(B, c, Hh, Ww, generator=g) for c in segs]
    w = 0.2 * torch.randn(Cout, sum(segs), k, k, generator=g)
    b = 0.3 * torch.randn(Cout, generator=g) if has_b else None
    kap = torch.tensor([[[[0.3]]]]) if has_k else None
    # ---- reference (fp64 on CPU)
    xr = [t.double().requires_grad_(True) for t in xs]
    wr = w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if has_b else None
    kr = kap.double().requires_grad_(True) if has_k else None
    t = torch.cat(xr, 1)
    if relu_in:
        t = F.relu(t)
    if k == 3:
        t = F.pad(t, (1, 1, 1, 1), mode="replicate" if pad_rep else "constant")
    yr = F.conv2d(t, wr, br, stride=stride)
    if has_k:
        yr = yr * torch.exp(torch.clamp(kr, -4.0, math.log(4.0)))
    if relu_out:
        yr = F.relu(yr)
    gy = torch.randn(yr.shape, generator=g).double()
    (yr * gy).sum().backward()
    # ---- HIP
    xd = [_nhwc(t_).requires_grad_(True) for t_ in xs]
    wd = w.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True) if has_b else None
    kd = kap.to(DEV).requires_grad_(True) if has_k else None
    y = ops.conv(xd, wd, bd, kappa=kd, ksize=k, stride=stride, relu_in=relu_in, pad_rep=pad_rep, relu_out=relu_out)
    _close(_back(y), yr, what="conv out")
    (y * _nhwc(gy.float())).sum().backward()
    _close(wd.grad, wr.grad, tol=5e-5, what="dW")
    if has_b:
        _close(bd.grad, br.grad, tol=5e-5, what="db")
    if has_k:
        _close(kd.grad, kr.grad, tol=1e-4, what="dkappa")
    for a, r in zip(xd, xr):
        _close(_back(a.grad), r.grad, tol=5e-5, what="dx")


def test_conv_on_channel_slice_views():
    """Segments that are channel slices of wider NHWC buffers (the no-concat path)."""
    import tmg_ops as ops
    g = torch.Generator().manual_seed(3)
    big = torch.randn(2, 10, 12, 16, generator=g)  # NHWC
    cond = torch.randn(2, 10, 12, 8, generator=g)
    w = 0.2 * torch.randn(12, 16, 3, 3, generator=g)
    bigd, condd = big.to(DEV), cond.to(DEV)
    y = ops.conv([bigd[..., :8], condd], w.to(DEV), relu_in=True)
    ref = F.conv2d(F.relu(torch.cat([big[..., :8], cond], 3).permute(0, 3, 1, 2)), w, padding=1)
    _close(_back(y), ref, what="sliced conv")


def test_bn_relu_conv():
    import tmg_ops as ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 12, 10, 14, generator=g) * 2 + 0.5
    gamma, beta = 1 + 0.3 * torch.randn(12, generator=g), 0.2 * torch.randn(12, generator=g)
    w = 0.2 * torch.randn(4, 12, 3, 3, generator=g)
    xr, gr, br, wr = (t.double().requires_grad_(True) for t in (x, gamma, beta, w))
    yr = F.conv2d(F.relu(F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)), wr, padding=1)
    gy = torch.randn(yr.shape, generator=g).double()
    (yr * gy).sum().backward()
    xd = _nhwc(x).requires_grad_(True)
    gd, bd, wd = (t.to(DEV).requires_grad_(True) for t in (gamma, beta, w))
    mean, var, n = ops.batch_moments(xd.detach())
    _close(mean, x.double().mean((0, 2, 3)), what="bn mean")
    _close(var, x.double().var((0, 2, 3), unbiased=False), what="bn var")
    # the fused statistics kernel: same moments, folded affine, and nn.BatchNorm2d's running-statistics update
    bn = torch.nn.BatchNorm2d(12).to(DEV)
    ref_bn = torch.nn.BatchNorm2d(12).double()
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
        ref_bn.weight.copy_(gamma); ref_bn.bias.copy_(beta)
    ref_bn.train()(x.double())
    mean2, rstd, a, bsh = ops.bn_batch_stats(xd.detach(), bn)
    _close(mean2, mean, what="fused mean")
    _close(rstd, torch.rsqrt(var + 1e-5), what="fused rstd")
    _close(a, gamma.to(DEV) * rstd, what="fused a")
    _close(bsh, beta.to(DEV) - mean * a, what="fused bsh")
    _close(bn.running_mean, ref_bn.running_mean, what="running mean")
    _close(bn.running_var, ref_bn.running_var, what="running var")
    assert int(bn.num_batches_tracked) == 1
    y = ops.BNReLUConvFn.apply(xd, gd, bd, wd, mean2, rstd, a, bsh, True)
    _close(_back(y), yr, what="bnreluconv")
    (y * _nhwc(gy.float())).sum().backward()
    _close(_back(xd.grad), xr.grad, tol=1e-4, what="bn dx")
    _close(gd.grad, gr.grad, tol=1e-4, what="dgamma")
    _close(bd.grad, br.grad, tol=1e-4, what="dbeta")
    _close(wd.grad, wr.grad, tol=1e-4, what="bn dW")


@pytest.mark.parametrize("reverse", [False, True])
def test_affine(reverse):
    import tmg_ops as ops
    g = torch.Generator().manual_seed(7)
    B, C_, Hh, Ww = 3, 12, 6, 10
    x = torch.randn(B, C_, Hh, Ww, generator=g)
    hh = torch.randn(B, C_, Hh, Ww, generator=g)
    xr, hr = x.double().requires_grad_(True), hh.double().requires_grad_(True)
    x1, x2 = xr.chunk(2, 1)
    shift, scale = hr[:, 0::2], torch.exp(2 * F.softsign(hr[:, 1::2]))
    o2 = x2 / scale - shift if reverse else (x2 + shift) * scale
    yr = torch.cat([x1, o2], 1)
    ldr = scale.abs().log().reshape(B, -1).sum(1)
    gy, gl = torch.randn(yr.shape, generator=g).double(), torch.randn(B, generator=g).double()
    ((yr * gy).sum() + (ldr * gl).sum()).backward()
    xd, hd = _nhwc(x).requires_grad_(True), _nhwc(hh).requires_grad_(True)
    y, ld = ops.AffineFn.apply(hd, xd, reverse)
    _close(_back(y), yr, what="affine y")
    _close(ld, ldr, what="affine logdet")
    ((y * _nhwc(gy.float())).sum() + (ld * gl.float().to(DEV)).sum()).backward()
    _close(_back(xd.grad), xr.grad, what="affine dx")
    _close(_back(hd.grad), hr.grad, what="affine dh")


@pytest.mark.parametrize("with_c", [True, False])
def test_lstm_pointwise(with_c):
    import tmg_ops as ops
    g = torch.Generator().manual_seed(9)
    B, R, Hh, Ww = 2, 6, 5, 7
    gates = torch.randn(B, 4 * R, Hh, Ww, generator=g)
    c = torch.randn(B, R, Hh, Ww, generator=g)
    gr = gates.double().requires_grad_(True)
    cr = c.double().requires_grad_(True)
    i, f, o, gg = torch.split(gr, R, 1)
    cn = torch.sigmoid(f) * (cr if with_c else 0) + torch.sigmoid(i) * torch.tanh(gg)
    hn = torch.sigmoid(o) * torch.tanh(cn)
    w1, w2 = torch.randn(hn.shape, generator=g).double(), torch.randn(hn.shape, generator=g).double()
    ((hn * w1).sum() + (cn * w2).sum()).backward()
    gd = _nhwc(gates).requires_grad_(True)
    cd = _nhwc(c).requires_grad_(True) if with_c else None
    h2, c2 = ops.LSTMPointwiseFn.apply(gd, cd)
    assert torch.equal(gd.detach(), _nhwc(gates))      # the forward launch only reads the pre-activation gates
    _close(_back(h2), hn, what="h")
    _close(_back(c2), cn, what="c")
    ((h2 * _nhwc(w1.float())).sum() + (c2 * _nhwc(w2.float())).sum()).backward()
    _close(_back(gd.grad), gr.grad, what="dgates")
    if with_c:
        _close(_back(cd.grad), cr.grad, what="dc_prev")


@pytest.mark.parametrize("clip", [1, 0])
def test_gauss_logp_and_sample(clip):
    import tmg_ops as ops
    g = torch.Generator().manual_seed(11)
    B, Ch, Hh, Ww = 3, 5, 4, 6
    hz = 1.5 * torch.randn(B, 2 * Ch, Hh, Ww, generator=g)
    z2 = torch.randn(B, Ch, Hh, Ww, generator=g)
    limits = ops.SPLIT_LIMITS if clip else ops.TOP_LIMITS

    def prior(h):
        if clip:
            h = F.hardtanh(h, -2.0, math.log(5.0))
        m, s = h.chunk(2, 1)
        return m, s.clamp(-10.0, math.log(5.0))

    hr, zr = hz.double().requires_grad_(True), z2.double().requires_grad_(True)
    m, s = prior(hr)
    lp = (-0.5 * (math.log(2 * math.pi) + 2 * s + (zr - m) ** 2 / torch.exp(2 * s))).reshape(B, -1).sum(1)
    epsr = (zr - m) / torch.exp(s)
    gl = torch.randn(B, generator=g).double()
    (lp * gl).sum().backward()
    hd, zd = _nhwc(hz).requires_grad_(True), _nhwc(z2).requires_grad_(True)
    logp, eps = ops.GaussLogpFn.apply(hd, zd, clip, limits, True)
    _close(logp, lp, what="logp")
    _close(_back(eps), epsr, what="eps")
    (logp * gl.float().to(DEV)).sum().backward()
    _close(_back(hd.grad), hr.grad, what="dhz")
    _close(_back(zd.grad), zr.grad, what="dz2")
    # sample direction
    hr2 = hz.double().requires_grad_(True)
    m, s = prior(hr2)
    e = torch.randn(B, Ch, Hh, Ww, generator=g)
    zs = m + torch.exp(s) * e.double()
    lps = (-0.5 * (math.log(2 * math.pi) + 2 * s + (zs - m) ** 2 / torch.exp(2 * s))).reshape(B, -1).sum(1)
    gz = torch.randn(zs.shape, generator=g).double()
    ((zs * gz).sum() + (lps * gl).sum()).backward()
    hd2 = _nhwc(hz).requires_grad_(True)
    z_, lp_ = ops.GaussSampleFn.apply(hd2, _nhwc(e), clip, limits)
    _close(_back(z_), zs, what="sample z")
    _close(lp_, lps, what="sample logp")
    ((z_ * _nhwc(gz.float())).sum() + (lp_ * gl.float().to(DEV)).sum()).backward()
    _close(_back(hd2.grad), hr2.grad, what="sample dhz")


@pytest.mark.parametrize("clip,Ch,with_z1", [(1, 5, True), (0, 8, False), (1, 6, True), (0, 64, True)])
def test_gauss_draw_node_given_latents(clip, Ch, with_z1):
    """Round 6: Split.reverse / GaussianDiag.sample as ONE launch (tmg_gauss_sample via ops.GaussDrawFn) - the sample written into the
    second half of the [.., 2 Ch] result beside the pass-through half - against the fp64 statement of flowUtils.py:194-209, :325-335
    (cat(z1, mean + exp(log-std) eps), its log-prob) with gradients for hz and z1; channel counts that are not multiples of 4 and
    inputs that are channel-slice views included."""
    import tmg_ops as ops
    g = torch.Generator().manual_seed(19)
    B, Hh, Ww = 3, 5, 7
    hz = 1.5 * torch.randn(B, 2 * Ch, Hh, Ww, generator=g)
    e = torch.randn(B, Ch, Hh, Ww, generator=g)
    z1 = torch.randn(B, Ch, Hh, Ww, generator=g)
    limits = ops.SPLIT_LIMITS if clip else ops.TOP_LIMITS
    hr, z1r = hz.double().requires_grad_(True), z1.double().requires_grad_(True)
    h = F.hardtanh(hr, -2.0, math.log(5.0)) if clip else hr
    m, sd = h.chunk(2, 1)
    sd = sd.clamp(-10.0, math.log(5.0))
    z2 = m + torch.exp(sd) * e.double()
    out_r = torch.cat([z1r, z2], 1) if with_z1 else z2
    lp_r = (-0.5 * (math.log(2 * math.pi) + 2 * sd + e.double() ** 2)).reshape(B, -1).sum(1)
    go = torch.randn(out_r.shape, generator=g).double()
    gl = torch.randn(B, generator=g).double()
    ((out_r * go).sum() + (lp_r * gl).sum()).backward()
    # HIP: hz / z1 / eps as channel-slice views of wider buffers (the kernels address them in place)
    wide = torch.zeros(B, Hh, Ww, 2 * Ch + 3, device=DEV)
    wide[..., :2 * Ch] = _nhwc(hz)
    hd = wide[..., :2 * Ch].detach().requires_grad_(True)
    z1d = _nhwc(z1).requires_grad_(True) if with_z1 else None
    out, lp = ops.GaussDrawFn.apply(hd, z1d, _nhwc(e), None, clip, limits)
    _close(_back(out), out_r, what="out")
    _close(lp, lp_r, what="logp")
    ((out * _nhwc(go.float())).sum() + (lp * gl.float().to(DEV)).sum()).backward()
    _close(_back(hd.grad), hr.grad, what="dhz")
    if with_z1:
        _close(_back(z1d.grad), z1r.grad, what="dz1")


def test_gauss_draw_node_in_kernel_latents():
    """The latents tmg_gauss_sample draws itself (Philox4x32-10 keyed by a nonce from torch's generator + Box-Muller): the stored eps
    reproduces the written sample exactly, the same (nonce, site) gives the same numbers, another site / nonce independent ones, the
    moments are those of N(0, 1) (mean, variance, skewness, kurtosis, tails) and neighbouring elements / quads are uncorrelated; the
    nonce follows torch.manual_seed."""
    import tmg_ops as ops
    B, Hh, Ww, Ch = 4, 64, 64, 16
    hz = torch.zeros(B, Hh, Ww, 2 * Ch, device=DEV)
    hz[..., :Ch] = 0.25
    hz[..., Ch:] = math.log(2.0)
    torch.manual_seed(1234)
    nonce = ops.latent_nonce(torch.device(DEV))
    torch.manual_seed(1234)
    assert torch.equal(ops.latent_nonce(torch.device(DEV)), nonce)         # follows torch's generator
    assert not torch.equal(ops.latent_nonce(torch.device(DEV)), nonce)     # ... and advances it

    def draw(nonce, site):
        ctx_out, lp = ops.GaussDrawFn.apply(hz.clone().requires_grad_(True), None, None, (nonce, site), 0, ops.TOP_LIMITS)
        eps = ctx_out.grad_fn.saved_tensors[1]
        return ctx_out.detach(), lp.detach(), eps

    z, lp, eps = draw(nonce, 2)
    assert torch.allclose(z, 0.25 + 2.0 * eps, rtol=0, atol=1e-6)
    ref_lp = (-0.5 * (math.log(2 * math.pi) + 2 * math.log(2.0) + eps.double() ** 2)).reshape(B, -1).sum(1)
    _close(lp, ref_lp, what="logp of drawn latents")
    z_b, _, eps_b = draw(nonce, 2)
    assert torch.equal(eps_b, eps)
    _, _, eps_c = draw(nonce, 3)
    _, _, eps_d = draw(ops.latent_nonce(torch.device(DEV)), 2)
    e = eps.double().reshape(-1)
    n = e.numel()                                                         # 262 144 draws: standard errors ~ 2e-3 (mean) .. 1e-2 (kurtosis)
    assert abs(float(e.mean())) < 1e-2 and abs(float(e.var()) - 1.0) < 1.5e-2
    assert abs(float((e ** 3).mean())) < 3e-2 and abs(float((e ** 4).mean()) - 3.0) < 8e-2
    assert abs(float((e.abs() > 2.0).double().mean()) - 0.0455) < 3e-3 and float(e.abs().max()) > 3.8
    for other in (eps_c, eps_d, torch.roll(eps, 1, 3), torch.roll(eps, 4, 3), torch.roll(eps, 1, 2), torch.roll(eps, 1, 0)):
        c = float((e * other.double().reshape(-1)).mean())
        assert abs(c) < 1e-2, c                                            # uncorrelated: |corr| ~ 1 / sqrt(n) = 2e-3
    assert n == B * Hh * Ww * Ch


def test_reverse_loss_node_matches_torch():
    """ops.reverse_loss (tmg_reverse_loss_fwd / _bwd: one reduction launch, one gradient launch) against the torch statement of the
    benchmark loss, tests/common.py::loss_reverse, values and gradients; a size that is not a multiple of 4 included."""
    import tmg_ops as ops
    g = torch.Generator().manual_seed(5)
    for shape in ((4, 3, 10, 14), (3, 3, 5, 7), (8, 4, 32, 32)):
        y = torch.randn(shape, generator=g)
        ld = 100.0 * torch.randn(shape[0], generator=g)
        yr, lr = y.double().requires_grad_(True), ld.double().requires_grad_(True)
        ref = C.loss_reverse(yr, lr)
        (ref * 1.7).backward()
        yd = y.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ldd = ld.to(DEV).requires_grad_(True)
        got = ops.reverse_loss(yd, ldd)
        assert got.shape == ()
        _close(got, ref, tol=2e-6, what="loss")
        (got * 1.7).backward()
        _close(yd.grad, yr.grad, tol=2e-6, what="dy")
        _close(ldd.grad, lr.grad, tol=2e-6, what="dlogdet")
        yc = y.to(DEV).requires_grad_(True)                       # an NCHW-contiguous input takes the converting path
        _close(ops.reverse_loss(yc, ldd.detach()), ref, tol=2e-6, what="loss, NCHW input")


def test_sum_logdet_one_launch():
    """ops.sum_logdet: [B] vectors, broadcast scalars (0-d and 1-element tensors), python numbers and None, more terms than one launch
    takes; gradients are the upstream vector / its sum."""
    import tmg_ops as ops
    B = 7
    g = torch.Generator().manual_seed(3)
    vec = [torch.randn(B, generator=g) for _ in range(11)]
    sc = [torch.randn((), generator=g), torch.randn(1, generator=g)]
    terms_r = [v.double().requires_grad_(True) for v in vec] + [s_.double().requires_grad_(True) for s_ in sc]
    ref = sum(terms_r[:11]) + terms_r[11] + terms_r[12] + 2.5
    gl = torch.randn(B, generator=g)
    (ref * gl.double()).sum().backward()
    terms_d = [t.float().to(DEV).requires_grad_(True) for t in vec + sc]
    got = ops.sum_logdet(terms_d[:5] + [None, 2.5, 0.0] + terms_d[5:], B, torch.device(DEV))
    _close(got, ref.detach(), what="sum")
    (got * gl.to(DEV)).sum().backward()
    for td, tr in zip(terms_d, terms_r):
        _close(td.grad, tr.grad, what="grad")
    assert ops.sum_logdet([None, 0.0], B, torch.device(DEV)) == 0.
    one = terms_d[0].detach()
    assert ops.sum_logdet([one], B, torch.device(DEV)) is one


def test_level_pack_matches_torch_stacks(monkeypatch):
    """tmg_level_pack (the stacked / sliced parameter operands of a level node from the modules' own tensors, one launch) against the
    torch stack / slice / cat statement it replaces (kept behind TMG_NO_LEVEL_PACK), for a level with one padding layer (NL = 15) and
    one with three (NL = 5)."""
    import tmg_ops as ops
    g = torch.Generator().manual_seed(8)
    for NL, C, Cc in ((15, 16, 32), (5, 8, 12), (3, 64, 32)):
        ch = C // 2
        cin = ch + Cc
        NLp = (NL + 3) // 4 * 4
        wts = []
        for _ in range(NL):
            wts += [torch.randn(1, cin, 3, 3, generator=g).to(DEV), torch.randn(1, cin + 1, 3, 3, generator=g).to(DEV),
                    torch.randn(C, cin + 2, 3, 3, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV), torch.randn(1, 1, 1, 1, generator=g).to(DEV)]
        got = ops.LevelCouplingFn._level_operands(wts, NL, NLp, C, ch, Cc, torch.device(DEV))
        monkeypatch.setenv("TMG_NO_LEVEL_PACK", "1")
        ref = ops.LevelCouplingFn._level_operands(wts, NL, NLp, C, ch, Cc, torch.device(DEV))
        monkeypatch.delenv("TMG_NO_LEVEL_PACK")
        for a, b, what in zip(got, ref, ("Wz", "Wcat", "Bz", "Kp")):
            assert torch.equal(a, b), (NL, C, what)
        assert float(got[1][NL * C + 2 * NL:].abs().max()) == 0.0 if NLp > NL else True


def test_checker_and_upsample():
    import tmg_ops as ops
    from oracle import tmglow_oracle as O
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 3, 8, 12, generator=g)
    xd = _nhwc(x).requires_grad_(True)
    y = ops.CheckerFn.apply(xd, True)
    ref = O.checker_squeeze(x)
    assert torch.equal(_back(y), ref)
    gy = torch.randn(ref.shape, generator=g)
    (y * _nhwc(gy)).sum().backward()
    assert torch.equal(_back(xd.grad), O.checker_unsqueeze(gy))
    back = ops.CheckerFn.apply(y.detach(), False)
    assert torch.equal(_back(back), x)
    # (channels % 4 == 0: the float4 kernels; scale 4 and one-row maps exceed their candidate window and take the scalar kernels, as do
    # the odd channel counts)
    for (h, w, sc, ch) in [(5, 7, 2, 4), (8, 8, 2, 4), (3, 4, 4, 4), (1, 6, 2, 4), (6, 5, 2, 3), (16, 24, 2, 32), (2, 2, 2, 8), (9, 4, 3, 8)]:
        x = torch.randn(2, ch, h, w, generator=g)
        xr = x.double().requires_grad_(True)
        yr = F.interpolate(xr, scale_factor=sc, mode="bilinear", align_corners=True)
        gy = torch.randn(yr.shape, generator=g).double()
        (yr * gy).sum().backward()
        xd = _nhwc(x).requires_grad_(True)
        y = ops.UpsampleFn.apply(xd, sc)
        _close(_back(y), yr, what="upsample")
        (y * _nhwc(gy.float())).sum().backward()
        _close(_back(xd.grad), xr.grad, what="upsample bwd")


@pytest.mark.parametrize("ch,pad", [(6, 2), (3, 1), (5, 3), (2, 2)])
def test_pad_halves_is_the_concatenation_with_zeros_and_its_own_adjoint(ch, pad):
    """tmg_pad_halves (the zero-padded channel layout of 3-channel fields, flowLSTMBlock._pad_x): bit-identical to
    cat(x1, 0, x2, 0) / its inverse, gradients are the opposite direction, slices of wider tensors are addressed in place."""
    import tmg_ops as ops
    g = torch.Generator().manual_seed(ch * 10 + pad)
    wide = torch.randn(3, 5, 7, 2 * ch + 3, generator=g).to(DEV)
    x = wide[..., 1:1 + 2 * ch]                                   # a channel-slice view: pixel stride != channel count
    z = torch.zeros(3, 5, 7, pad, device=DEV)
    want = torch.cat([x[..., :ch], z, x[..., ch:], z], 3)
    xi = x.detach().clone().requires_grad_(True)
    y = ops.PadHalvesFn.apply(xi, ch, pad, True)
    y2 = ops.PadHalvesFn.apply(x, ch, pad, True)
    assert torch.equal(y, want) and torch.equal(y2, want)
    gy = torch.randn(want.shape, generator=g).to(DEV)
    (y * gy).sum().backward()
    assert torch.equal(xi.grad, torch.cat([gy[..., :ch], gy[..., ch + pad:2 * ch + pad]], 3))
    yp = want.clone().requires_grad_(True)
    back = ops.PadHalvesFn.apply(yp, ch, pad, False)
    assert torch.equal(back, x)
    gb = torch.randn(back.shape, generator=g).to(DEV)
    (back * gb).sum().backward()
    assert torch.equal(yp.grad, torch.cat([gb[..., :ch], z, gb[..., ch:], z], 3))


@pytest.mark.parametrize("shape", [(2, 16, 16, [8, 32]), (1, 7, 9, [6, 5]), (2, 32, 32, [8, 32, 4]), (2, 12, 12, [40])])
def test_c1_forward_and_backward(shape):
    import tmg_hip as Hh_
    B, Hh, Ww, segs = shape
    g = torch.Generator().manual_seed(17)
    xs = [torch.randn(B, c, Hh, Ww, generator=g) for c in segs]
    cin = sum(segs)
    w = 0.2 * torch.randn(1, cin, 3, 3, generator=g)
    xr = [t.double().requires_grad_(True) for t in xs]
    wr = w.double().requires_grad_(True)
    act = F.relu(torch.cat(xr, 1))
    act.retain_grad()
    yr = F.conv2d(act, wr, padding=1)
    dref = torch.randn(yr.shape, generator=g).double()       # pre-activation of the produced channel
    dd = torch.randn(yr.shape, generator=g).double()
    (yr * (dd * (dref > 0))).sum().backward()
    xd = [_nhwc(t) for t in xs]
    wd = w.to(DEV).reshape(cin, 9).contiguous()
    out = torch.zeros(B, Hh, Ww, 4, device=DEV)
    Hh_.c1_fwd(xd, wd, out[..., 1:2], relu_in=True)
    _close(out[..., 1].cpu(), yr[:, 0], what="c1 fwd")
    assert float(out[..., 0].abs().max()) == 0 and float(out[..., 2:].abs().max()) == 0
    dW = torch.zeros(cin, 9, device=DEV)
    gs = [torch.zeros_like(t) for t in xd]
    ddn, drn = _nhwc(dd.float()), _nhwc(dref.float())
    Hh_.c1_bwd(xd, wd, dW, ddn, drn, gs, relu_in=True)
    _close(dW.cpu().reshape(1, cin, 3, 3), wr.grad, tol=5e-5, what="c1 dW")
    got = torch.cat([_back(t) for t in gs], 1)
    _close(got, act.grad, tol=5e-5, what="c1 raw input grad")


@pytest.mark.parametrize("tag", ["small", "clamped"])
def test_physics_loss_matches_reference_fixture(tag):
    """Row F1: fused physics-constrained loss kernels against the fixture recorded from the reference's TMGLowLoss."""
    from types import SimpleNamespace
    from nn.trainFlowParallel import TMGLowLoss
    from pc.physicsConstrained import PhysConstrainedLES
    d = C.load_npz("phys_loss.npz")
    t = lambda k: torch.from_numpy(d[tag + "." + k])  # noqa: E731
    beta, dx, dy = (float(v) for v in d[tag + ".cfg"])
    model = SimpleNamespace(out_std=t("std"), out_mu=t("mu"))
    crit = TMGLowLoss(SimpleNamespace(beta=beta, dx=dx, dy=dy), model).to(DEV)
    y = t("y").to(DEV).requires_grad_(True)
    logp = t("logp").to(DEV).requires_grad_(True)
    loss = crit(y, logp, t("target").to(DEV), t("tmean").to(DEV), t("trms").to(DEV))
    ref = float(d[tag + ".loss"])
    assert abs(loss.item() - ref) <= 2e-5 * abs(ref), (loss.item(), ref)
    loss.backward()
    C.assert_grads({"y": y.grad, "logp": logp.grad}, {"y": d[tag + ".dy"], "logp": d[tag + ".dlogp"]}, "loss grads",
                   global_tol=2e-5, tensor_tol=2e-4)
    hat = (t("std").view(1, 3, 1, 1) * t("y").reshape(-1, 3, y.shape[-2], y.shape[-1]) + t("mu").view(1, 3, 1, 1)).to(DEV)
    phys = PhysConstrainedLES(dx, dy)
    _close(phys.calcPressurePoisson(hat[:, :2], hat[:, 2:]), t("pstar"), what="pstar")
    _close(phys.calcDivergence(hat[:, :2]), t("ustar"), what="ustar")


@pytest.mark.parametrize("shape", [(8, 10, 256, 256), (3, 4, 100, 70), (2, 10, 128, 256)])
def test_physics_loss_at_field_sizes_of_the_trainer(shape):
    """The fused loss kernels at the sizes the trainer runs them at (a 10-step window of 256x256x3 fields, several samples: thousands
    of tiles; and a ragged field) against the physics oracle evaluated in FP64 - the reference fixtures pin the arithmetic on small
    fields only.  Loss value and both gradients."""
    import os
    import sys
    from types import SimpleNamespace
    sys.path.insert(0, os.path.join(C.ROOT, "oracle"))
    import physics_oracle as PO
    from nn.trainFlowParallel import TMGLowLoss
    B, T, Hh, Ww = shape
    g = torch.Generator().manual_seed(Hh + T)
    y = 0.6 * torch.randn(B, T, 3, Hh, Ww, generator=g)
    tgt = 0.6 * torch.randn(B, T, 3, Hh, Ww, generator=g)
    logp = torch.randn(B, T, generator=g) * 50.0
    std, mu = torch.tensor([1.3, 0.7, 2.1]), torch.tensor([0.2, -0.1, 0.4])
    beta, dx, dy = 200.0, 2.0 / 64, 2.0 / 64
    tmean = tgt.mean(1)
    trms = torch.sqrt(((tgt - tmean.unsqueeze(1)) ** 2).mean(1))
    yr, lr = y.double().requires_grad_(True), logp.double().requires_grad_(True)
    ref = PO.tmglow_loss(yr, lr, tgt.double(), tmean.double(), trms.double(), std.double(), mu.double(), beta, dx, dy)
    ref.backward()
    crit = TMGLowLoss(SimpleNamespace(beta=beta, dx=dx, dy=dy), SimpleNamespace(out_std=std, out_mu=mu)).to(DEV)
    yd, ld = y.to(DEV).requires_grad_(True), logp.to(DEV).requires_grad_(True)
    loss = crit(yd, ld, tgt.to(DEV), tmean.to(DEV), trms.to(DEV))
    assert abs(loss.item() - ref.item()) <= 2e-5 * abs(ref.item()), (loss.item(), ref.item())
    loss.backward()
    C.assert_grads({"y": yd.grad, "logp": ld.grad}, {"y": yr.grad, "logp": lr.grad}, "loss grads", global_tol=2e-5, tensor_tol=2e-4)


@pytest.mark.parametrize("shape", [(4, 2, 16, 16, 8, 16, 32), (3, 1, 20, 12, 16, 32, 5), (5, 2, 32, 32, 32, 64, 32), (3, 4, 16, 16, 64, 128, 32)])
def test_grouped_weight_gradient_matches_per_group_launches(shape):
    """tmg_conv_wgrad_grouped (one launch, device segment table) against G separate tmg_conv_wgrad launches and against
    fp64 autograd; the destination mapping is the level node's (x1 | growth buffer rows of a wider weight tensor)."""
    import tmg_hip as H
    G, B, Hh, Ww, ch, C, Cc = shape
    g = torch.Generator().manual_seed(G * 100 + ch)
    cin = ch + Cc
    xs = [torch.randn(B, Hh, Ww, 2 * ch, generator=g).to(DEV) for _ in range(G)]   # x1 is a channel-slice view of these
    Ds = [torch.randn(B, Hh, Ww, 4, generator=g).to(DEV) for _ in range(G)]
    DH = torch.randn(B, Hh, Ww, G * C, generator=g).to(DEV)
    groups = [[x[..., :ch], d] for x, d in zip(xs, Ds)]
    dW = torch.zeros(G, C, cin + 2, 3, 3, device=DEV)
    dB = torch.zeros(G, C, device=DEV)
    ok = H.conv_wgrad_grouped(groups, DH, C, dW, dB, 3, 1, relu_in=True, pad_rep=True, cin_dst=cin + 2, cin_valid=ch + 2, ci_split=ch,
                              ci_off0=0, ci_off1=Cc)
    dW1 = torch.zeros_like(dW)
    dB1 = torch.zeros_like(dB)
    for k in range(G):
        H.conv_wgrad(groups[k], DH[..., k * C:(k + 1) * C], dW1[k], dB1[k], 3, 1, relu_in=True, pad_rep=True, cin_dst=cin + 2,
                     cin_valid=ch + 2, ci_split=ch, ci_off0=0, ci_off1=Cc)
    if not ok:  # the library may decline a shape; the per-group path is then the product path and is checked below
        dW, dB = dW1, dB1
    _close(dW, dW1, tol=2e-5, what="grouped vs per-group dW")
    _close(dB, dB1, tol=2e-5, what="grouped vs per-group dbias")
    for k in range(G):
        x1 = xs[k][..., :ch].permute(0, 3, 1, 2).double().cpu()
        d2 = Ds[k][..., :2].permute(0, 3, 1, 2).double().cpu()
        t = F.pad(F.relu(torch.cat([x1, d2], 1)), (1, 1, 1, 1), mode="replicate")
        w = torch.zeros(C, ch + 2, 3, 3, dtype=torch.float64, requires_grad=True)
        bb = torch.zeros(C, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(t, w, bb)
        (y * DH[..., k * C:(k + 1) * C].permute(0, 3, 1, 2).double().cpu()).sum().backward()
        _close(dW[k][:, :ch], w.grad[:, :ch], tol=3e-5, what="dW x1 rows")
        _close(dW[k][:, ch + Cc:], w.grad[:, ch:], tol=3e-5, what="dW growth rows")
        assert float(dW[k][:, ch:ch + Cc].abs().max()) == 0.0  # the conditioning rows belong to the level-wide launch
        _close(dB[k], bb.grad, tol=3e-5, what="dbias")


@pytest.mark.parametrize("shape", [(2, 16, 16, 8, 32), (1, 9, 13, 4, 5), (3, 32, 8, 16, 32), (2, 40, 24, 64, 32)])
def test_fused_growth_layers_match_two_launches(shape):
    """tmg_c1x2_fwd (both growth-1 layers, one launch) against two tmg_c1_fwd launches and fp64 torch."""
    import tmg_hip as H
    B, Hh, Ww, ch, Cc = shape
    g = torch.Generator().manual_seed(ch * 31 + Hh)
    x = torch.randn(B, Hh, Ww, 2 * ch, generator=g).to(DEV)
    x1 = x[..., :ch]
    cin = ch + Cc
    w1 = (0.3 * torch.randn(1, cin, 3, 3, generator=g)).to(DEV)
    w2 = (0.3 * torch.randn(1, cin + 1, 3, 3, generator=g)).to(DEV)
    Dc = torch.randn(B, Hh, Ww, 8, generator=g).to(DEV)
    D = torch.full((B, Hh, Ww, 4), 7.0, device=DEV)
    H.c1x2_fwd([x1], w1, w2, D, w_rows=ch, w2_d1_row=ch + Cc, add1=Dc[..., 1:2], add2=Dc[..., 5:6])
    D2 = torch.full((B, Hh, Ww, 4), 7.0, device=DEV)
    H.c1_fwd([x1], w1, D2[..., 0:1], relu_in=True, w_rows=ch, fill4=True, add=Dc[..., 1:2])
    H.c1_fwd([x1, D2], w2, D2[..., 1:2], relu_in=True, w_rows=ch + 1, w_split=ch, w_gap=Cc, add=Dc[..., 5:6])
    _close(D, D2, tol=2e-5, what="fused vs two launches")
    xr = x1.permute(0, 3, 1, 2).double().cpu()
    d1 = F.conv2d(F.relu(xr), w1[:, :ch].double().cpu(), padding=1) + Dc[..., 1:2].permute(0, 3, 1, 2).double().cpu()
    w2x = torch.cat([w2[:, :ch], w2[:, ch + Cc:]], 1).double().cpu()
    d2 = F.conv2d(F.relu(torch.cat([xr, d1], 1)), w2x, padding=1) + Dc[..., 5:6].permute(0, 3, 1, 2).double().cpu()
    _close(D[..., 0:1].permute(0, 3, 1, 2), d1, tol=3e-5, what="d1")
    _close(D[..., 1:2].permute(0, 3, 1, 2), d2, tol=3e-5, what="d2")
    assert float(D[..., 2:].abs().max()) == 0.0


@pytest.mark.parametrize("C_,npix,slice_of", [(8, 37, 0), (12, 100, 0), (16, 4096, 0), (24, 333, 0), (32, 1000, 48), (48, 65, 0),
                                               (64, 2048, 0), (80, 129, 0), (96, 257, 0), (128, 640, 160), (192, 33, 0), (256, 300, 0)])
def test_mix_f16_kernel(C_, npix, slice_of):
    """tmg_mix_f16 (fp16 operands, fp32 accumulation) against the same arithmetic spelled out in torch: operands rounded to
    fp16, products and sums exact (fp64).  Every product of two fp16 numbers is exact in fp32, so only the fp32 summation
    order separates the kernel from this reference: tolerance 2e-6 relative to max|y|.  Also W^T (the input-gradient form),
    channel-slice inputs (pixel stride > C) and ragged pixel counts."""
    import tmg_hip as H
    g = torch.Generator().manual_seed(C_ * 1000 + npix)
    W = (torch.randn(C_, C_, generator=g) / math.sqrt(C_)).to(DEV)
    b = torch.randn(C_, generator=g).to(DEV)
    wide = slice_of or C_
    buf = torch.randn(1, 1, npix, wide, generator=g).to(DEV)
    x = buf[..., wide - C_:] if slice_of else buf          # channel-slice view: pixel stride `wide`
    obuf = torch.full((1, 1, npix, wide), 7.0, device=DEV)
    y = obuf[..., :C_] if slice_of else obuf
    x16, W16 = x.half().double(), W.half().double()
    for transposed in (False, True):
        obuf.fill_(7.0)
        H.mix_f16(x, W, None if transposed else b, y, transposed=transposed)
        ref = x16 @ (W16 if transposed else W16.t()) + (0 if transposed else b.double())
        _close(y, ref, tol=2e-6, what="mix_f16 C=%d transposed=%d" % (C_, transposed))
        if slice_of:
            assert bool((obuf[..., C_:] == 7.0).all()), "wrote outside its channel slice"
    # and it must differ from the fp32 product by about fp16's rounding (2^-11 relative per operand), not by more
    full = x.double() @ W.double().t() + b.double()
    H.mix_f16(x, W, b, y)
    dev = float((y.double() - full).abs().max()) / float(full.abs().max())
    assert 1e-5 < dev < 5e-3, dev


@pytest.mark.parametrize("case", [
    (2, 16, 16, [16, 32, 64], 256, False, False, True),    # the gate-conv pattern: three segments, bias, zero padding
    (2, 16, 12, [32], 240, True, True, False),             # the conditioning contraction: ReLU on the way in, replicate padding
    (1, 9, 21, [8, 32, 64], 256, False, False, True),      # 104 input channels (last chunk half full), partial tiles on both axes
    (2, 8, 8, [32], 1920, True, True, False),              # several output-channel blocks
    (3, 5, 3, [16], 64, False, True, True),                # image smaller than a tile
    (1, 32, 48, [64, 64], 128, True, False, True),         # several tiles per image, two full chunks per tile
    (2, 16, 24, [40], 104, False, False, False),           # one output tile per wave (<= 128 channels), 7 of 8 waves live: the out-conv input gradient
    (2, 12, 16, [48], 112, False, False, True),            # the same at the second level's widths
    (1, 16, 16, [96], 144, False, True, True),             # 9 tiles: back on the two-tiles-per-wave instance
])
@pytest.mark.parametrize("arith", ["f32", "bf16x3"])
def test_winograd_conv_matches_fp64(case, arith):
    """tmg_conv_wino_fwd (Winograd F(2x2,3x3) on the fp32 matrix cores) against fp64 F.conv2d, at the direct kernel's tolerance,
    and against the direct kernel itself.  arith "bf16x3": the opt-in tmg_conv_wino_fwd3 (three-way bf16 split of both operands, six
    part products on the bf16 matrix pipe) - held to the SAME tolerances, and its error against fp64 to at most 1.5x the fp32 kernel's."""
    import tmg_hip as H
    B, Hh, Ww, segs, Cout, relu_in, pad_rep, has_b = case
    g = torch.Generator().manual_seed(sum(segs) + Cout + Hh)
    xs = [torch.randn(B, Hh, Ww, c, generator=g) for c in segs]
    w = 0.2 * torch.randn(Cout, sum(segs), 3, 3, generator=g)
    b = 0.3 * torch.randn(Cout, generator=g) if has_b else None
    t = torch.cat(xs, 3).permute(0, 3, 1, 2).double()
    if relu_in:
        t = F.relu(t)
    t = F.pad(t, (1, 1, 1, 1), mode="replicate" if pad_rep else "constant")
    ref = F.conv2d(t, w.double(), b.double() if has_b else None).permute(0, 2, 3, 1)
    xd = [x.to(DEV) for x in xs]
    wd, bd = w.to(DEV), (b.to(DEV) if has_b else None)
    out = torch.full((B, Hh, Ww, Cout), float("nan"), device=DEV)
    assert H.wino_eligible(sum(segs), Cout, 3, 1)
    assert H.conv_wino_fwd(xd, H.conv_wino_pack(wd), Cout, [out[..., :32], out[..., 32:]], bias=bd, relu_in=relu_in, pad_rep=pad_rep)
    if arith == "bf16x3":
        out3 = torch.full((B, Hh, Ww, Cout), float("nan"), device=DEV)
        assert H.conv_wino_fwd3(xd, H.conv_wino_pack3(wd), Cout, [out3[..., :32], out3[..., 32:]], bias=bd, relu_in=relu_in, pad_rep=pad_rep)
        e32 = float((out.double().cpu() - ref).abs().max())
        e3 = float((out3.double().cpu() - ref).abs().max())
        assert e3 <= 1.5 * e32 + 1e-7 * float(ref.abs().max()), (e3, e32)
        out = out3
    _close(out, ref, what="winograd conv")
    direct = torch.empty_like(out)
    H.conv_fwd(xd, H.conv_pack(wd, 0), Cout, 3, 1, [direct], bias=bd, relu_in=relu_in, pad_rep=pad_rep)
    _close(out, direct.double(), what="winograd vs direct")


@pytest.mark.parametrize("kind", ["wide_exponents", "cancellation", "tiny_beside_unit", "all_tiny"])
def test_winograd_bf16x3_adversarial_operands(kind):
    """The bf16x3 arithmetic (three-way bf16 split of both transformed operands, six part products) on operands chosen against it,
    with the fp32-MFMA Winograd kernel's own error against fp64 as the yardstick (same data, same transforms):
      wide_exponents    per-channel scales 2^-30 .. 2^+30 on the activations, the inverse on the weights: every product is O(1) but the
                        operands of one contraction span 60 binades;
      cancellation      channel pairs with equal activations and negated weights (their exact sum is zero) beside one small live
                        channel: the result is 1e-4 of the terms that cancel;
      tiny_beside_unit  activations of 2^-112 (their third bf16 part is an fp32 denormal) beside O(1) channels;
      all_tiny          every activation at 2^-112 .. 2^-108: the third parts are denormal and the matrix pipe flushes them - the
                        documented envelope of the switch: the result keeps >= 15 bits (the first two parts), not 24.
    """
    import tmg_hip as H
    g = torch.Generator().manual_seed(31)
    B, Hh, Ww, Cin, Cout = 2, 16, 16, 64, 256
    x = torch.randn(B, Hh, Ww, Cin, generator=g)
    w = 0.2 * torch.randn(Cout, Cin, 3, 3, generator=g)
    if kind == "wide_exponents":
        e = torch.linspace(-30, 30, Cin).round()
        x = x * torch.exp2(e)
        w = w * torch.exp2(-e).view(1, Cin, 1, 1)
    elif kind == "cancellation":
        x[..., 1:Cin - 1:2] = x[..., 0:Cin - 2:2]
        w[:, 1:Cin - 1:2] = -w[:, 0:Cin - 2:2]
        x[..., Cin - 2:] *= 1e-4
    elif kind == "tiny_beside_unit":
        x[..., ::2] *= 2.0 ** -112
    else:
        x = x * torch.exp2(torch.randint(-112, -107, (Cin,), generator=g).float())
    ref = F.conv2d(F.pad(x.permute(0, 3, 1, 2).double(), (1, 1, 1, 1)), w.double()).permute(0, 2, 3, 1)
    xd, wd = x.to(DEV), w.to(DEV)
    out = torch.empty((B, Hh, Ww, Cout), device=DEV)
    out3 = torch.empty((B, Hh, Ww, Cout), device=DEV)
    assert H.conv_wino_fwd([xd], H.conv_wino_pack(wd), Cout, [out])
    assert H.conv_wino_fwd3([xd], H.conv_wino_pack3(wd), Cout, [out3])
    scale = float(ref.abs().max())
    e32 = float((out.double().cpu() - ref).abs().max()) / scale
    e3 = float((out3.double().cpu() - ref).abs().max()) / scale
    print("\nbf16x3 adversarial %-18s scale %.3e  fp32-MFMA err %.3e  bf16x3 err %.3e (relative to the largest output)" % (kind, scale, e32, e3))
    assert torch.isfinite(out3).all()
    if kind == "all_tiny":
        assert e3 <= 2.0 ** -15, (e3, e32)
    else:
        assert e3 <= 1.5 * e32 + 1e-7, (e3, e32)


def test_winograd_bf16x3_input_gradient_operand():
    """tmg_conv_wino_pack3 mode 1 (the input-gradient operand: transposed weight, flipped taps, channel prefix) through
    conv3x3_auto with the bf16x3 switch on, against fp64 conv_transpose2d - the ConvLSTM out-conv's input gradient 40 -> 104."""
    import tmg_hip as H
    g = torch.Generator().manual_seed(404)
    B, Hh, Ww, K, N = 2, 16, 24, 40, 104
    x = torch.randn(B, Hh, Ww, K, generator=g)
    w = 0.2 * torch.randn(K, N + 8, 3, 3, generator=g)          # forward conv [Cout = K][Cin = N + 8]; gradient w.r.t. its first N inputs
    ref = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1)[:, :N].permute(0, 2, 3, 1)
    out = torch.full((B, Hh, Ww, N), float("nan"), device=DEV)
    H.set_winograd_precision("bf16x3")
    try:
        assert H.conv3x3_auto([x.to(DEV)], w.to(DEV), N, [out], dgrad=True, nvalid=N) is None      # (None: a Winograd kernel took it)
    finally:
        H.set_winograd_precision("f32")
    _close(out, ref, what="bf16x3 winograd input gradient")


@pytest.mark.parametrize("case", [
    (2, 16, 16, 256, [8, 32], True, False, False),        # gate input gradient: 4R dy channels -> the first 40 input channels, two outputs
    (2, 16, 12, 240, [32], True, False, False),           # conditioning gradient: 15 C -> Cc (last chunk half full)
    (1, 9, 21, 104, [40], False, True, True),             # forward operand, bias + ReLU out, partial tiles
    (2, 8, 8, 1920, [32], True, False, False),            # long contraction
    (3, 5, 3, 64, [16], False, True, False),              # image smaller than a tile, one channel tile
])
def test_winograd_narrow_matches_fp64(case):
    """tmg_conv_wino_narrow (few output channels: the waves split the 16 Winograd positions) with the forward operand and with the
    input-gradient operand (mode 1: transposed weight, flipped taps, channel prefix), against fp64 torch."""
    import tmg_hip as H
    B, Hh, Ww, K, outs_c, dgrad, has_b, relu_out = case
    N = sum(outs_c)
    g = torch.Generator().manual_seed(K + N + Hh)
    x = torch.randn(B, Hh, Ww, K, generator=g)
    if dgrad:
        # weight of the forward conv: [Cout = K][Cin >= N]; the gradient w.r.t. its first N input channels
        cin_full = N + 8
        w = 0.2 * torch.randn(K, cin_full, 3, 3, generator=g)
        ref = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1)[:, :N].permute(0, 2, 3, 1)
        U = H.conv_wino_pack(w.to(DEV), 1, N)
    else:
        w = 0.2 * torch.randn(N, K, 3, 3, generator=g)
        b = 0.3 * torch.randn(N, generator=g) if has_b else None
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double() if has_b else None, padding=1)
        if relu_out:
            ref = F.relu(ref)
        ref = ref.permute(0, 2, 3, 1)
        U = H.conv_wino_pack(w.to(DEV))
    bd = b.to(DEV) if (not dgrad and has_b) else None
    outs = [torch.full((B, Hh, Ww, c), float("nan"), device=DEV) for c in outs_c]
    assert H.wino_narrow_eligible(K, N)
    assert H.conv_wino_narrow([x.to(DEV)], U, N, outs, bias=bd, relu_out=relu_out)
    _close(torch.cat(outs, 3), ref, what="narrow winograd")


@pytest.mark.parametrize("case", [
    (2, 16, 16, [8, 32, 64], 256, False, False, True),     # gate weight gradient: 104 input channels (two channel groups), bias
    (2, 16, 12, [32], 240, True, True, False),             # conditioning weight gradient into a wider destination
    (1, 9, 21, [8, 32, 64], 40, False, False, True),       # three output-channel tiles, partial tiles on both axes
    (3, 8, 8, [64], 32, True, False, False),
    (2, 24, 40, [48], 96, False, True, True),              # several tiles per block share
])
def test_winograd_weight_gradient_matches_fp64(case):
    """tmg_conv_wino_wgrad (Winograd F(3x3, 2x2) on the fp32 matrix cores) against fp64 autograd, at the direct
    kernel's tolerance, including the destination-column mapping of the level node."""
    import tmg_hip as H
    B, Hh, Ww, segs, Cout, relu_in, pad_rep, has_b = case
    Cin = sum(segs)
    g = torch.Generator().manual_seed(Cin * 3 + Cout + Ww)
    xs = [torch.randn(B, Hh, Ww, c, generator=g) for c in segs]
    dy = torch.randn(B, Hh, Ww, Cout, generator=g)
    t = torch.cat(xs, 3).permute(0, 3, 1, 2).double()
    if relu_in:
        t = F.relu(t)
    t = F.pad(t, (1, 1, 1, 1), mode="replicate" if pad_rep else "constant")
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    (F.conv2d(t, w, b) * dy.permute(0, 3, 1, 2).double()).sum().backward()
    off = 5 if len(segs) == 1 else 0                          # single-segment cases land in columns [off, off + Cin) of a wider tensor
    dW = torch.zeros(Cout, Cin + 2 * off, 3, 3, device=DEV)
    dB = torch.zeros(Cout, device=DEV) if has_b else None
    assert H.conv_wino_wgrad([x.to(DEV) for x in xs], dy.to(DEV), dW, dB, relu_in=relu_in, pad_rep=pad_rep,
                             cin_dst=Cin + 2 * off, cin_valid=Cin, ci_off0=off)
    _close(dW[:, off:off + Cin], w.grad, tol=5e-5, what="winograd dW")
    if off:
        assert float(dW[:, :off].abs().max()) == 0.0 and float(dW[:, off + Cin:].abs().max()) == 0.0
    if has_b:
        _close(dB, b.grad, tol=5e-5, what="winograd dbias")


@pytest.mark.parametrize("C_,npix,slice_of", [(16, 1000, 0), (32, 4096, 0), (64, 777, 96), (128, 2048, 0), (24, 513, 40), (96, 300, 0)])
def test_mix_f32_kernel(C_, npix, slice_of):
    """tmg_mix_f32 (the stand-alone 1x1 mix in fp32 on the matrix cores) against fp64: W and W^T, bias, channel-slice views
    (pixel stride > C), ragged pixel counts; nothing written outside the channel slice."""
    import tmg_hip as H
    g = torch.Generator().manual_seed(C_ * 77 + npix)
    W = (torch.randn(C_, C_, generator=g) / math.sqrt(C_)).to(DEV)
    b = torch.randn(C_, generator=g).to(DEV)
    wide = slice_of or C_
    buf = torch.randn(1, 1, npix, wide, generator=g).to(DEV)
    x = buf[..., wide - C_:] if slice_of else buf
    obuf = torch.full((1, 1, npix, wide), 7.0, device=DEV)
    y = obuf[..., :C_] if slice_of else obuf
    for transposed in (False, True):
        obuf.fill_(7.0)
        assert H.mix_f32(x, W, None if transposed else b, y, transposed=transposed)
        ref = x.double() @ (W.double() if transposed else W.double().t()) + (0 if transposed else b.double())
        _close(y, ref, tol=2e-6, what="mix_f32 C=%d transposed=%d" % (C_, transposed))
        if slice_of:
            assert bool((obuf[..., C_:] == 7.0).all()), "wrote outside its channel slice"


@pytest.mark.parametrize("amsgrad,wd", [(True, 1e-8), (False, 0.0), (True, 0.05)])
def test_hip_adam_matches_torch_adam(amsgrad, wd):
    """tmg_optim.HipAdam (one launch for all parameters) against torch.optim.Adam over four steps on tensors of assorted sizes
    (scalar, odd lengths, several 4096-element chunks): parameters and all three state tensors; the state dict loads across."""
    from tmg_optim import HipAdam
    g = torch.Generator().manual_seed(17)
    shapes = [(1,), (7,), (33, 5), (4096,), (3, 4097), (16, 16, 3, 3), (1, 1, 1, 1), (20000,)]
    pa = [torch.randn(s_, generator=g).to(DEV).requires_grad_(True) for s_ in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = HipAdam(pa, lr=3e-3, weight_decay=wd, amsgrad=amsgrad)
    ob = torch.optim.Adam(pb, lr=3e-3, weight_decay=wd, amsgrad=amsgrad, foreach=False)
    for it in range(4):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).to(DEV) * (10.0 ** (it - 2))
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        _close(a, b.detach().double(), tol=2e-6, what="parameter")
        sa, sb = oa.state[a], ob.state[b]
        assert float(sa["step"]) == float(sb["step"]) == 4
        for k in ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if amsgrad else ()):
            _close(sa[k], sb[k].double(), tol=2e-6, what=k)
    ob.load_state_dict(oa.state_dict())     # same schema


@pytest.mark.parametrize("NL,C_,Cc,with_x", [(3, 16, 8, True), (15, 32, 32, True), (2, 128, 32, False), (1, 12, 4, True)])
def test_level_finish_kernel(NL, C_, Cc, with_x):
    """tmg_level_finish (parameter-gradient epilogue of a level's fused coupling node, one launch) against the torch formulation
    it replaces: d(kappa) = (<Wz, dWz> + <bz, dBz>) inside the clamp range in fp64, and the scatter-add of the grouped
    weight-gradient rows into dW1 / dW2."""
    import tmg_hip as H
    g = torch.Generator().manual_seed(5 + NL)
    ch = C_ // 2
    cin = ch + Cc
    rnd = lambda *s_: torch.randn(s_, generator=g).to(DEV)  # noqa: E731
    Wz, dWz, Bz, dBz = rnd(NL, C_, cin + 2, 3, 3), rnd(NL, C_, cin + 2, 3, 3), rnd(NL, C_), rnd(NL, C_)
    Kp = torch.tensor([(-4.5, -4.0, 0.3, 1.38, 1.3862944, 1.5)[k % 6] for k in range(NL)], device=DEV)
    tmpX = rnd(NL, 4, ch + 4, 3, 3) if with_x else None
    tmpC = rnd(NL, 4, Cc, 3, 3)
    dW1, dW2 = rnd(NL, 1, cin, 3, 3), rnd(NL, 1, cin + 1, 3, 3)
    r1, r2 = dW1.clone(), dW2.clone()
    if with_x:
        r1[:, 0, :ch] += tmpX[:, 0, :ch]
        r2[:, 0, :ch] += tmpX[:, 1, :ch]
        r2[:, 0, cin] += tmpX[:, 1, ch]
    r1[:, 0, ch:cin] += tmpC[:, 0]
    r2[:, 0, ch:cin] += tmpC[:, 1]
    rK = ((Wz.double() * dWz.double()).flatten(1).sum(1) + (Bz.double() * dBz.double()).sum(1)).float() \
        * ((Kp >= -4.0) & (Kp <= math.log(4.0))).to(torch.float32)
    dK = torch.full((NL,), 7.0, device=DEV)
    # a misaligned dWz view exercises the scalar path of the inner product
    dWz_mis = torch.cat([torch.zeros(1, device=DEV), dWz.flatten()])[1:].view(dWz.shape)
    for gz in (dWz, dWz_mis):
        a1, a2 = dW1.clone(), dW2.clone()
        H.level_finish(Wz, gz, Bz, dBz, Kp, tmpX, tmpC, a1, a2, dK, torch.zeros(4 * NL, device=DEV), ch, Cc)
        assert torch.equal(a1, r1) and torch.equal(a2, r2)
        assert float((dK - rK).abs().max()) <= 1e-6 * max(float(rK.abs().max()), 1.0), (dK, rK)
        assert bool(((dK == 0) == (rK == 0)).all())


def test_zero_pool_hands_out_disjoint_zeroed_buffers():
    """tmg_ops.zeros: small buffers are carved from a shared zero-filled chunk (one fill launch for many buffers) - zero content,
    256-byte aligned, pairwise disjoint, unaffected by writes to their neighbours; large requests get a tensor of their own."""
    import tmg_ops
    bufs = [tmg_ops.zeros(s_, DEV) for s_ in [(3,), (5, 7), (1,), (64, 64, 3, 3), (2, 2)] * 8]
    for b in bufs:
        assert b.dtype == torch.float32 and b.is_contiguous() and b.data_ptr() % 256 == 0 and not bool(b.any())
    for i, b in enumerate(bufs):
        b.fill_(float(i + 1))
    for i, b in enumerate(bufs):
        assert bool((b == float(i + 1)).all())
    spans = sorted((b.data_ptr(), b.data_ptr() + 4 * b.numel()) for b in bufs)
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
    big = tmg_ops.zeros((tmg_ops._ZeroPool.LIMIT + 1,), DEV)
    assert big.untyped_storage().nbytes() == 4 * big.numel() and not bool(big.any())
    assert bufs[0].untyped_storage().nbytes() > 4 * bufs[0].numel() and bufs[0]._base is None      # pooled, but not an autograd view
    assert tmg_ops.zeros_like(bufs[1]).shape == bufs[1].shape


def test_adopted_torch_adam_steps_like_torch_adam():
    """tmg_optim.adopt: the optimizer main.py constructs (torch.optim.Adam, weight decay 1e-8, amsgrad; main.py:78-79 wraps it in an
    ExponentialLR right away) turned into the one-launch HipAdam IN PLACE: same object, same state schema, the scheduler that was
    constructed before the adoption still sees the steps and moves the learning rate the kernel uses."""
    import warnings
    import tmg_optim
    from torch.optim.lr_scheduler import ExponentialLR
    g = torch.Generator().manual_seed(29)
    shapes = [(3,), (4100,), (16, 8, 3, 3), (1, 1, 1, 1)]
    pa = [torch.randn(s_, generator=g).to(DEV).requires_grad_(True) for s_ in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = torch.optim.Adam(pa, lr=1e-3, weight_decay=1e-8, amsgrad=True)
    ob = torch.optim.Adam(pb, lr=1e-3, weight_decay=1e-8, amsgrad=True, foreach=False)
    sa, sb = ExponentialLR(oa, gamma=0.5), ExponentialLR(ob, gamma=0.5)
    ident = id(oa)
    assert tmg_optim.adopt(oa) and isinstance(oa, tmg_optim.HipAdam) and id(oa) == ident and tmg_optim.adopt(oa)
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # "lr_scheduler.step() before optimizer.step()" would mean the wrapper lost the steps
        for it in range(5):
            for a, b in zip(pa, pb):
                gr = torch.randn(a.shape, generator=g).to(DEV)
                a.grad, b.grad = gr.clone(), gr.clone()
            oa.step()
            ob.step()
            if it % 2 == 1:
                sa.step()
                sb.step()
    assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"] == 1e-3 * 0.25
    for a, b in zip(pa, pb):
        _close(a, b.detach().double(), tol=3e-6, what="parameter")
        for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            _close(oa.state[a][k], ob.state[b][k].double(), tol=3e-6, what=k)
    ob.load_state_dict(oa.state_dict())
    # not adopted: another class, fused groups
    assert not tmg_optim.adopt(torch.optim.SGD(pb, lr=0.1))
    assert not tmg_optim.adopt(torch.optim.Adam(pb, lr=1e-3, fused=True))


def test_hip_adam_cached_path_follows_changes():
    """HipAdam's cached steady-state path (parameter / state pointers kept between steps) against torch.optim.Adam when things change
    under it: a parameter without a gradient in one step, a changed learning rate, and a state dict loaded half-way."""
    import copy
    from tmg_optim import HipAdam
    g = torch.Generator().manual_seed(23)
    shapes = [(5,), (4100,), (8, 3, 3, 3), (1,)]
    pa = [torch.randn(s_, generator=g).to(DEV).requires_grad_(True) for s_ in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = HipAdam(pa, lr=2e-3, weight_decay=1e-8, amsgrad=True)
    ob = torch.optim.Adam(pb, lr=2e-3, weight_decay=1e-8, amsgrad=True, foreach=False)
    for it in range(11):
        skip = 2 if it == 3 else -1                   # parameter 2 takes no part in step 3
        for i, (a, b) in enumerate(zip(pa, pb)):
            gr = torch.randn(a.shape, generator=g).to(DEV)
            a.grad, b.grad = (None, None) if i == skip else (gr.clone(), gr.clone())
        if it == 5:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 5e-4
        if it == 7:                                   # round trip through the state dict (a workspace reload)
            oa.load_state_dict(copy.deepcopy(ob.state_dict()))     # (deep copy: load_state_dict keeps the tensors it is given)
        if it == 8:
            # storage re-materialised under the SAME Parameter / state-dict objects (model.to(...), `p.data = ...`, a state tensor
            # swapped by hand for a middle parameter): the cached pointer table must not be used
            pa[1].data = pa[1].data.clone()
            oa.state[pa[2]]["exp_avg"] = oa.state[pa[2]]["exp_avg"].clone()
        oa.step()
        ob.step()
        for a, b in zip(pa, pb):
            _close(a, b.detach().double(), tol=3e-6, what="parameter after step %d" % it)
    for a, b in zip(pa, pb):
        assert float(oa.state[a]["step"]) == float(ob.state[b]["step"])
        for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            _close(oa.state[a][k], ob.state[b][k].double(), tol=3e-6, what=k)


@pytest.mark.parametrize("shape", [(3, 2, 16, 16, 8), (15, 1, 20, 35, 8), (4, 2, 33, 17, 16), (5, 1, 16, 32, 32), (3, 2, 9, 16, 64)])
def test_thin_grouped_weight_gradient(shape):
    """tmg_conv_wgrad_thin_grouped (growth-layer weight gradients, four output channels per group, 4x4x1 MFMA blocks) against the
    general grouped kernel it replaces and against fp64 autograd: ragged tiles, every supported channel half, ReLU'd inputs, zero
    padding, channel-slice views as inputs, a shared dy tensor with the groups side by side."""
    import os
    import tmg_hip as H
    G, B, Hh, Ww, ch = shape
    g = torch.Generator().manual_seed(G * 10 + ch)
    xs = [torch.randn(B, Hh, Ww, 2 * ch, generator=g).to(DEV) for _ in range(G)]
    Ds = [torch.randn(B, Hh, Ww, 4, generator=g).to(DEV) for _ in range(G)]
    DD = torch.randn(B, Hh, Ww, 4 * G, generator=g).to(DEV)
    groups = [[x[..., :ch], d] for x, d in zip(xs, Ds)]
    res = {}
    for mode in ("thin", "general"):
        os.environ.pop("TMG_NO_THIN_WGRAD", None)
        if mode == "general":
            os.environ["TMG_NO_THIN_WGRAD"] = "1"
        dW = torch.zeros(G, 4, ch + 4, 3, 3, device=DEV)
        try:
            assert H.conv_wgrad_grouped(groups, DD, 4, dW, None, 3, 1, relu_in=True)
        finally:
            os.environ.pop("TMG_NO_THIN_WGRAD", None)
        res[mode] = dW
    _close(res["thin"], res["general"], tol=2e-5, what="thin vs general grouped kernel")
    for k in range(G):
        xin = torch.cat([xs[k][..., :ch], Ds[k]], 3).permute(0, 3, 1, 2).double().relu()
        w = torch.zeros(4, ch + 4, 3, 3, dtype=torch.float64, device=DEV, requires_grad=True)
        y = F.conv2d(xin, w, padding=1)
        (y * DD[..., 4 * k:4 * k + 4].permute(0, 3, 1, 2).double()).sum().backward()
        _close(res["thin"][k], w.grad, tol=2e-5, what="group %d vs fp64" % k)


def test_pack_plan_repacks_every_parameter_operand_once_per_update():
    """tmg_hip._PackPlan (round 5): the packed operands of PARAMETERS are re-packed by one tmg_conv_pack_many launch per parameter
    update instead of one launch per request - values identical to the per-call pack, stale after an in-place update (torch version
    counter), after a write behind torch's back that bumps PARAM_GENERATION, after the storage moved; temporaries and hipGraph
    recording are not cached; dead parameters are dropped."""
    import gc
    import tmg_hip as H
    plan = H._PACK_PLAN
    plan.jobs.clear()
    g = torch.Generator().manual_seed(5)
    ws = [torch.nn.Parameter(torch.randn(co, ci, k, k, generator=g).to(DEV)) for co, ci, k in ((16, 8, 3), (40, 20, 3), (32, 32, 1), (7, 10, 3))]
    specs = [(0, 0, None), (1, 0, None), (0, 24, None), (1, 12, (6, 4, 4))]      # (the last: 4 leading + 2 trailing of 10 source channels)
    calls = {"many": 0}
    real_many = H.conv_pack_many

    def counting_many(jobs):
        calls["many"] += 1
        return real_many(jobs)
    H.conv_pack_many = counting_many
    try:
        def request_all():
            return [H.conv_pack(w, m, ce, cm) for w, (m, ce, cm) in zip(ws, specs)]

        def plain_all():
            return [H.conv_pack(w.detach().clone(), m, ce, cm) for w, (m, ce, cm) in zip(ws, specs)]     # (temporaries: never cached)
        first = request_all()
        n_learn = calls["many"]
        assert n_learn == len(ws) and len(plan.jobs) == len(ws)          # learning pass: one launch per new operand
        for a, b in zip(first, plain_all()):
            assert torch.equal(a, b)
        again = request_all()
        assert calls["many"] == n_learn and all(a is b for a, b in zip(first, again))       # unchanged parameters: look-ups
        with torch.no_grad():
            for w in ws:
                w.mul_(1.5)                                              # what torch.optim does: version counters move
        upd = request_all()
        assert calls["many"] == n_learn + 1, "one launch re-packs every stale operand"
        for a, b, old in zip(upd, plain_all(), first):
            assert torch.equal(a, b) and not torch.equal(a, old)
        ws[1].data.mul_(2.0)                                             # behind torch's back: needs the generation bump
        assert H.conv_pack(ws[1], *specs[1]) is upd[1]
        H.PARAM_GENERATION[0] += 1
        upd2 = request_all()
        assert calls["many"] == n_learn + 2
        for a, b in zip(upd2, plain_all()):
            assert torch.equal(a, b)
        ws[0].data = ws[0].data.clone()                                  # storage moved
        assert torch.equal(H.conv_pack(ws[0], *specs[0]), upd2[0]) and calls["many"] == n_learn + 3
        # a hipGraph recording bypasses the plan
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            rec = H.conv_pack(ws[2], *specs[2])
        assert rec is not upd2[2]
        gph.replay()
        torch.cuda.synchronize()
        assert torch.equal(rec, upd2[2])
        # dead parameters leave the plan at the next re-pack
        del ws[3], first, again, upd, upd2, w
        gc.collect()
        with torch.no_grad():
            ws[0].add_(1.0)
        H.conv_pack(ws[0], *specs[0])
        assert len(plan.jobs) == 3
    finally:
        H.conv_pack_many = real_many
        plan.jobs.clear()


@pytest.mark.parametrize("shape", [(3, 2, 16, 16, 16), (15, 1, 20, 35, 16), (4, 3, 17, 9, 32), (2, 1, 64, 64, 32)])
def test_mix_weight_gradient_grouped_kernel(shape):
    """tmg_mix_wgrad_grouped (weight / bias gradients of the 1x1 mixes of all layers of a level, streaming GEMM over the pixels)
    against the general grouped kernel and fp64: two-segment mix inputs (x1 as a channel-slice view | y2), every group with its own
    upstream gradient, pixel counts that are not a multiple of the unroll."""
    import os
    import tmg_hip as H
    G, B, Hh, Ww, C_ = shape
    ch = C_ // 2
    g = torch.Generator().manual_seed(G + C_)
    tins = [torch.randn(B, Hh, Ww, C_, generator=g).to(DEV) for _ in range(G)]
    y2s = [torch.randn(B, Hh, Ww, ch, generator=g).to(DEV) for _ in range(G)]
    douts = [torch.randn(B, Hh, Ww, C_, generator=g).to(DEV) for _ in range(G)]
    ins = [[t[..., :ch], y2] for t, y2 in zip(tins, y2s)]
    res = {}
    for mode in ("stream", "general"):
        os.environ.pop("TMG_NO_MIX_WGRAD_KERNEL", None)
        if mode == "general":
            os.environ["TMG_NO_MIX_WGRAD_KERNEL"] = "1"
        dW = torch.zeros(G, C_, C_, device=DEV)
        db = torch.zeros(G, C_, device=DEV)
        try:
            assert H.conv_wgrad_grouped(ins, None, C_, dW.view(G, C_, C_, 1, 1), db, 1, 1, group_dy=douts)
        finally:
            os.environ.pop("TMG_NO_MIX_WGRAD_KERNEL", None)
        res[mode] = (dW, db)
    _close(res["stream"][0], res["general"][0], tol=2e-5, what="dW: streaming vs general grouped kernel")
    _close(res["stream"][1], res["general"][1], tol=2e-5, what="db: streaming vs general grouped kernel")
    for k in range(G):
        y = torch.cat([tins[k][..., :ch], y2s[k]], 3).reshape(-1, C_).double()
        d = douts[k].reshape(-1, C_).double()
        _close(res["stream"][0][k], d.t() @ y, tol=2e-5, what="dW group %d vs fp64" % k)
        _close(res["stream"][1][k], d.sum(0), tol=2e-5, what="db group %d vs fp64" % k)


@pytest.mark.parametrize("shape", [(2, 5, 7, 8), (1, 16, 16, 32), (3, 9, 33, 4)])
def test_layer_planes_kernel(shape):
    """tmg_layer_planes: [B,H,W,2K] -> [K,B,H,W,2] (pixel counts that are not a multiple of the 64-pixel block)."""
    import tmg_hip as H
    g = torch.Generator().manual_seed(sum(shape))
    src = torch.randn(*shape, generator=g).to(DEV)
    B, Hh, Ww, CP = shape
    assert torch.equal(H.layer_planes(src), src.view(B, Hh, Ww, CP // 2, 2).permute(3, 0, 1, 2, 4).contiguous())


@pytest.mark.parametrize("shape", [(64, 8, 16, 128, 1), (64, 16, 16, 128, 1), (128, 8, 16, 128, 1), (16, 16, 16, 256, 1), (64, 16, 32, 64, 1),
                                   (64, 8, 16, 96, 3), (64, 16, 16, 128, 3)])
def test_weight_gradient_plans_at_stated_batch_sizes(shape):
    """The weight-gradient launch plan depends on the pixel count (how many channel tiles a block stages, how the pairs are dealt
    to the waves), so the plans the BENCHMARKED batch sizes select must be tested at those sizes: with >= 8 192 pixels the 1x1
    contraction of the 128-channel level staged 8 channel tiles per block, four more than the lean staging path addressed - the mix
    weight gradients of that level were garbage at batch 64 while every batch-1 and batch-2 parity test was green (round 3).
    Single launch, two-segment input, and the grouped launch with per-group upstream gradients, against fp64."""
    import tmg_hip as H
    B, Hh, Ww, C_, k = shape
    g = torch.Generator().manual_seed(B + C_ + k)
    x = torch.randn(B, Hh, Ww, C_, generator=g).to(DEV)
    dy = torch.randn(B, Hh, Ww, C_, generator=g).to(DEV)
    xn, dn = x.permute(0, 3, 1, 2).double(), dy.permute(0, 3, 1, 2).double()
    w = torch.zeros(C_, C_, k, k, dtype=torch.float64, device=DEV, requires_grad=True)
    (F.conv2d(xn, w, padding=k // 2) * dn).sum().backward()
    tol = 3e-6 * (B * Hh * Ww) ** 0.5
    dW, db = torch.zeros(C_, C_, k, k, device=DEV), torch.zeros(C_, device=DEV)
    H.conv_wgrad([x], dy, dW, db, k, 1)
    _close(dW, w.grad, tol=tol, what="single launch")
    _close(db, dn.sum((0, 2, 3)), tol=tol, what="bias gradient")
    dW2 = torch.zeros(C_, C_, k, k, device=DEV)
    H.conv_wgrad([x[..., :C_ // 2], x[..., C_ // 2:]], dy, dW2, None, k, 1)
    _close(dW2, w.grad, tol=tol, what="two input segments")
    if k == 1:
        G = 3
        xs = [torch.randn(B, Hh, Ww, C_, generator=g).to(DEV) for _ in range(G)]
        ds = [torch.randn(B, Hh, Ww, C_, generator=g).to(DEV) for _ in range(G)]
        dWg, dbg = torch.zeros(G, C_, C_, 1, 1, device=DEV), torch.zeros(G, C_, device=DEV)
        if H.conv_wgrad_grouped([[t] for t in xs], None, C_, dWg, dbg, 1, 1, group_dy=ds):
            for i in range(G):
                ref = torch.einsum("bhwo,bhwi->oi", ds[i].double(), xs[i].double())
                _close(dWg[i].view(C_, C_), ref, tol=tol, what="grouped launch, group %d" % i)


@pytest.mark.parametrize("case", [(2, 16, 16, [8, 32], True), (1, 9, 20, [16, 32], True), (2, 8, 8, [32, 32], False), (2, 16, 8, [64, 32], True),
                                  (1, 12, 12, [4, 32], True)])
def test_conv_lstm_cell_node_matches_fp64(case):
    """ConvLSTMCellFn (reference convLSTM.py:72-85: gates = conv3x3(cat(t0, h)) + b, i / f / o / g activations, c' = f c + i g,
    h' = o tanh(c')) at the model's widths (64 recurrent features, 40 / 48 / 64 / 96 / 36 input channels): outputs and every gradient
    against fp64 autograd.  Covers the Winograd gate conv, its narrow input gradient (only the channels that need one), and the
    weight gradient as 48 + 64 input-channel blocks on the first two levels."""
    import tmg_ops as ops
    B, Hh, Ww, segs, h_grad = case
    R = 64
    g = torch.Generator().manual_seed(sum(segs) + Hh)
    xs = [torch.randn(B, Hh, Ww, c, generator=g) for c in segs]
    h0 = torch.randn(B, Hh, Ww, R, generator=g)
    c0 = torch.randn(B, Hh, Ww, R, generator=g)
    cin = sum(segs) + R
    w = 0.15 * torch.randn(4 * R, cin, 3, 3, generator=g)
    b = 0.2 * torch.randn(4 * R, generator=g)
    gh, gc = torch.randn(B, Hh, Ww, R, generator=g), torch.randn(B, Hh, Ww, R, generator=g)
    # fp64 reference
    xr = [t.double().requires_grad_(True) for t in xs]
    hr, cr = h0.double().requires_grad_(h_grad), c0.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    t = torch.cat(xr + [hr], 3).permute(0, 3, 1, 2)
    gates = F.conv2d(t, wr, br, padding=1).permute(0, 2, 3, 1)
    i_, f_, o_, g_ = torch.split(gates, R, 3)
    cn = torch.sigmoid(f_) * cr + torch.sigmoid(i_) * torch.tanh(g_)
    hn = torch.sigmoid(o_) * torch.tanh(cn)
    ((hn * gh.double()).sum() + (cn * gc.double()).sum()).backward()
    # HIP
    xd = [t.to(DEV).requires_grad_(True) for t in xs]
    hd, cd = h0.to(DEV).requires_grad_(h_grad), c0.to(DEV).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    h2, c2 = ops.ConvLSTMCellFn.apply(wd, bd, hd, cd, *xd)
    _close(h2, hn, what="h'")
    _close(c2, cn, what="c'")
    ((h2 * gh.to(DEV)).sum() + (c2 * gc.to(DEV)).sum()).backward()
    tol = 3e-5
    for a, r, what in zip(xd, xr, ["dx%d" % k for k in range(len(xs))]):
        _close(a.grad, r.grad, tol=tol, what=what)
    _close(cd.grad, cr.grad, tol=tol, what="dc")
    if h_grad:
        _close(hd.grad, hr.grad, tol=tol, what="dh")
    else:
        assert hd.grad is None
    _close(wd.grad, wr.grad, tol=tol, what="dW")
    _close(bd.grad, br.grad, tol=tol, what="db")


@pytest.mark.parametrize("training", [True, False])
def test_dense_block_node_matches_per_layer_path(training):
    """tmg_ops.DenseBlockFn (all layers of an encoder dense block on one pre-sized buffer, gradient accumulated in place) against
    the per-layer path it replaces (BNReLUConvFn + torch.cat, reference denseBlock.py:49-67): outputs, input gradient, every
    parameter gradient and the BatchNorm running statistics."""
    import os
    from nn.modules.denseBlock import DenseBlock
    C.seed_all(31)
    blk = DenseBlock(num_layers=4, in_features=16, growth_rate=4, drop_rate=0.)
    g = torch.Generator().manual_seed(32)
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.add_(0.2 * torch.randn(p_.shape, generator=g))
    blk.to(DEV).train(training)
    sd0 = {k: v.clone() for k, v in blk.state_dict().items()}
    x = torch.randn(3, 10, 14, 16, generator=g).to(DEV)
    gy = torch.randn(3, 10, 14, 32, generator=g).to(DEV)
    res = {}
    for tag, env in (("node", None), ("layers", "1")):
        if env:
            os.environ["TMG_NO_DENSE_BLOCK_NODE"] = env
        try:
            blk.load_state_dict(sd0)
            blk.zero_grad()
            xi = x.clone().requires_grad_(True)
            y = blk.run(xi)
            (y * gy).sum().backward()
            res[tag] = (y.detach().clone(), xi.grad.clone(), {k: p_.grad.clone() for k, p_ in blk.named_parameters()},
                        {k: v.clone() for k, v in blk.state_dict().items() if "running" in k or "num_batches" in k})
        finally:
            os.environ.pop("TMG_NO_DENSE_BLOCK_NODE", None)
    a, b = res["node"], res["layers"]
    assert a[0].shape == (3, 10, 14, 32)
    _close(a[0], b[0].double(), tol=1e-6, what="block output")
    _close(a[1], b[1].double(), tol=2e-5, what="input gradient")
    for k in b[2]:
        _close(a[2][k], b[2][k].double(), tol=2e-5, what=k)
    for k in b[3]:
        assert torch.allclose(a[3][k].float(), b[3][k].float(), rtol=1e-6, atol=1e-7), k


def test_conv_pack_many_matches_single_packs():
    """tmg_conv_pack_many (<= 16 jobs per launch through the kernel arguments; more jobs = more launches) against tmg_conv_pack /
    tmg_conv_pack_map job by job: mixed shapes, both modes, a zero-extended and a re-mapped input-channel dimension."""
    import tmg_hip as H
    g = torch.Generator().manual_seed(21)
    ws = [torch.randn(co, ci, k, k, generator=g).to(DEV) for co, ci, k in
          [(8, 20, 3), (4, 36, 3), (64, 64, 1), (17, 5, 3), (32, 44, 3), (6, 12, 3), (16, 16, 1), (40, 104, 3), (12, 8, 3)]]
    jobs = []
    for w in ws:
        jobs += [(w, 0), (w, 1)]
    jobs += [(ws[0], 0, 24), (ws[4], 0, 16, (12, 8, 30)), (ws[4], 1, 16, (12, 8, 30)), (ws[7], 1, 44, (40, 8, 60))]
    assert len(jobs) > 16                        # two launches
    got = H.conv_pack_many(jobs)
    assert len(got) == len(jobs)
    for j, pk in zip(jobs, got):
        ref = H.conv_pack(*j)
        assert pk.shape == ref.shape and torch.equal(pk, ref), j[1:]


def test_conv_node_with_premasked_output_gradient():
    """ops.conv(relu_out=True, _grad_premasked=True): with an upstream gradient that is already zero wherever the output is, the
    backward pass without its own mask launch returns the gradients of the masked form bit for bit."""
    import tmg_ops as ops
    g = torch.Generator().manual_seed(5)
    x = _nhwc(torch.randn(2, 8, 9, 11, generator=g))
    w = (0.3 * torch.randn(12, 8, 3, 3, generator=g)).to(DEV)
    b = torch.randn(12, generator=g).to(DEV)
    res = []
    for pre in (False, True):
        xi, wi, bi = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = ops.conv([xi], wi, bi, relu_out=True, _grad_premasked=pre)
        up = torch.randn(y.shape, generator=torch.Generator().manual_seed(6)).to(DEV) * (y.detach() > 0)     # the consumer's mask
        y.backward(up)
        res.append((y.detach(), xi.grad, wi.grad, bi.grad))
    for a, b_ in zip(*res):
        assert torch.equal(a, b_)
    assert float((res[0][0] == 0).float().mean()) > 0.2          # the ReLU really clipped something
    # the contract check (TMG_CHECK_PREMASK=1): an un-masked upstream gradient - a second consumer of `out` - is reported
    import os
    os.environ["TMG_CHECK_PREMASK"] = "1"
    try:
        xi, wi, bi = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = ops.conv([xi], wi, bi, relu_out=True, _grad_premasked=True)
        with pytest.raises(RuntimeError, match="contract violated"):
            y.backward(torch.ones_like(y))
    finally:
        os.environ.pop("TMG_CHECK_PREMASK", None)


def test_lstm_pointwise_backward_without_cell_state_gradient():
    """tmg_lstm_pointwise_bwd with dc_prev = NULL (the previous cell state carries no gradient): the gate gradients are those of the
    call that also writes dc_prev."""
    import tmg_hip as H
    g = torch.Generator().manual_seed(8)
    B, Hh, Ww, R = 2, 5, 6, 8
    gates = torch.randn(B, Hh, Ww, 4 * R, generator=g).to(DEV)
    c_prev = torch.randn(B, Hh, Ww, R, generator=g).to(DEV)
    dh = torch.randn(B, Hh, Ww, R, generator=g).to(DEV)
    dc = torch.randn(B, Hh, Ww, R, generator=g).to(DEV)
    c_next, h_next = torch.empty_like(c_prev), torch.empty_like(c_prev)
    H.lstm_pointwise_fwd(gates, c_prev, c_next, h_next)
    a1, a2 = gates.clone(), gates.clone()
    dcp = torch.empty_like(c_prev)
    H.lstm_pointwise_bwd(a1, c_prev, c_next, dh, dc, dcp)
    H.lstm_pointwise_bwd(a2, c_prev, c_next, dh, dc, None)
    assert torch.equal(a1, a2) and bool(torch.isfinite(dcp).all())


def test_fill_i64_writes_host_values_through_kernel_arguments():
    """tmg_fill_i64 (the grouped launches' pointer tables while a hipGraph is being recorded: values travel as kernel arguments, 256 per
    launch): 0, 1, 256, 257 and 700 values incl. negative and > 2^53 ones arrive bit-exactly; a captured fill replays its values."""
    import tmg_hip as H
    for n in (0, 1, 256, 257, 700):
        vals = [(-1) ** i * (i * 0x1234567 + (1 << 60) * (i % 3)) for i in range(n)]
        dst = torch.full((max(n, 1),), -7, dtype=torch.int64, device=DEV)
        rc = H.lib().tmg_fill_i64(H._ptr(dst), H._i64(*vals) if n else None, H.c_i64(n), H._stream())
        assert rc == 0
        if n:
            assert dst.cpu().tolist() == vals
        else:
            assert dst.cpu().tolist() == [-7]
    # inside a capture: the table of a grouped launch is built this way (tmg_hip._segment_table)
    rows = [[3 * i + j for j in range(16)] for i in range(5)]
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        H._segment_table(rows, torch.device(DEV, torch.cuda.current_device()))       # (warm-up: the eager, cached form)
    torch.cuda.current_stream().wait_stream(st)
    with torch.cuda.graph(g, stream=st):
        tab = H._segment_table(rows, torch.device(DEV, torch.cuda.current_device()))
        out = tab.clone()
    tab.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert out.cpu().tolist() == rows


@pytest.mark.parametrize("k1", [3, 5])
@pytest.mark.parametrize("k2", [3, 5])
@pytest.mark.parametrize("scale", [True, False])
def test_residual_fields_for_every_stencil_and_scaling(k1, k2, scale):
    """PhysConstrainedLES.calcDivergence / calcPressurePoisson on the HIP path with the reference's 3x3 and 5x5 stencils (every
    combination) and scale = True / False against the fields recorded from the reference (tests/golden/phys_fields.npz, partly clamped);
    stencil sizes the reference rejects raise ValueError here too."""
    from pc.physicsConstrained import PhysConstrainedLES
    d = C.load_npz("phys_fields.npz")
    dx, dy, rho = (float(v) for v in d["cfg"])
    tag = "k%d%d.%s" % (k1, k2, "scaled" if scale else "raw")
    au, ap = (float(v) for v in d[tag + ".amp"])
    u, p = torch.from_numpy(d["u"]).to(DEV), torch.from_numpy(d["p"]).to(DEV)
    phys = PhysConstrainedLES(dx, dy, rho=rho, grad_kernels=[k1, k2])
    C.assert_field(phys.calcDivergence(au * u, scale=scale), d[tag + ".ustar"], tag + " ustar", atol=2e-5)
    C.assert_field(phys.calcPressurePoisson(ap * u, ap * p, scale=scale), d[tag + ".pstar"], tag + " pstar", atol=2e-5)
    with pytest.raises(ValueError):
        PhysConstrainedLES(dx, dy, grad_kernels=[7, 3])
