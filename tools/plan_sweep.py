"""Sweep the launch plans of the contraction kernels over pixel counts for the model's channel configurations (all levels of
configs M / cfg5): forward, input gradient and weight gradient through the autograd node against fp64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
import tmg_ops as ops
dev = "cuda"
torch.manual_seed(0)
bad = 0
# (Cin segments, Cout, k, relu_in, pad_rep)
SHAPES = [([8, 4], 16, 3, True, True), ([16, 4], 32, 3, True, True), ([32, 4], 64, 3, True, True), ([64, 4], 128, 3, True, True), ([128, 4], 256, 3, True, True),
          ([16], 16, 1, False, False), ([32], 32, 1, False, False), ([64], 64, 1, False, False), ([128], 128, 1, False, False), ([256], 256, 1, False, False),
          ([8, 32, 64], 256, 3, False, False), ([16, 32, 64], 256, 3, False, False), ([32, 32, 64], 256, 3, False, False), ([64, 32, 64], 256, 3, False, False),
          ([8, 32, 64], 40, 3, False, False), ([64, 32, 64], 96, 3, False, False),
          ([32], 240, 3, True, True), ([32], 960, 3, True, True), ([32], 1920, 3, True, True), ([32], 32, 3, True, False),
          ([8], 16, 3, False, True), ([64], 128, 3, False, True)]
for segs, Cout, k, relu, rep in SHAPES:
    for (B, Hh, Ww) in [(1, 16, 16), (8, 16, 16), (32, 16, 16), (64, 16, 16), (128, 16, 16), (64, 8, 16), (64, 32, 32), (16, 64, 64)]:
        cin = sum(segs)
        if B * Hh * Ww * max(cin, Cout) > 3e8:
            continue
        xs = [torch.randn(B, Hh, Ww, c, device=dev, requires_grad=True) for c in segs]
        w = (0.2 * torch.randn(Cout, cin, k, k, device=dev)).requires_grad_(True)
        b = torch.randn(Cout, device=dev, requires_grad=True)
        y = ops.conv(xs, w, b, None, ksize=k, stride=1, relu_in=relu, pad_rep=rep)
        gy = torch.randn_like(y)
        gr = torch.autograd.grad(y, xs + [w, b], gy)
        xr = torch.cat([t.detach() for t in xs], 3).permute(0, 3, 1, 2).double().requires_grad_(True)
        wr = w.detach().double().requires_grad_(True)
        br = b.detach().double().requires_grad_(True)
        t = F.relu(xr) if relu else xr
        if k == 3:
            t = F.pad(t, (1, 1, 1, 1), mode="replicate" if rep else "constant")
        yr = F.conv2d(t, wr, br)
        grr = torch.autograd.grad(yr, [xr, wr, br], gy.permute(0, 3, 1, 2).double())
        def rel(a, r):
            return float((a.double() - r).abs().max() / r.abs().max().clamp_min(1e-30))
        e_y = rel(y.permute(0, 3, 1, 2), yr.detach())
        e_x = rel(torch.cat(gr[:len(segs)], 3).permute(0, 3, 1, 2), grr[0])
        e_w = rel(gr[-2], grr[1])
        e_b = rel(gr[-1], grr[2])
        tol = 2e-6 * (B * Hh * Ww) ** 0.5 + 1e-5
        flag = "" if max(e_y, e_x) < 2e-5 and max(e_w, e_b) < tol else "   <<<<<<<< BAD"
        if flag:
            bad += 1
        if flag or os.environ.get("VERBOSE"):
            print("segs %-14s -> %4d k%d relu%d rep%d  B%3d %2dx%-2d  y %.1e dx %.1e dW %.1e db %.1e%s" % (segs, Cout, k, relu, rep, B, Hh, Ww, e_y, e_x, e_w, e_b, flag))
print("bad cases:", bad)
