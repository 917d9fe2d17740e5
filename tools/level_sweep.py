"""Sweep one flow level (LSTMFLowBlock, generative direction + backward) over channel widths, field sizes and batch sizes against the
per-layer / per-op HIP path (TMG_NO_LEVEL_FUSION=1: every contraction through the general kernels): the fused / grouped kernels'
launch plans depend on the pixel count.  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import common as C
from nn.modules.flowLSTMBlock import LSTMFLowBlock
DEV = "cuda"
bad = 0
for cin, (hs, ws), B in [(4, (128, 128), 8), (4, (64, 64), 64), (4, (20, 36), 64), (8, (64, 64), 16), (8, (32, 32), 64), (8, (17, 33), 32),
                         (16, (32, 32), 16), (16, (16, 16), 64), (16, (16, 16), 128), (16, (9, 23), 64), (32, (16, 16), 64), (32, (8, 16), 64), (32, (16, 16), 128),
                         (3, (64, 64), 32), (3, (30, 30), 64), (64, (8, 8), 64)]:
    C.seed_all(cin * 7 + B)
    blk = LSTMFLowBlock(cin, 32, 64, 6, LUdecompose=True, train_sampling=True, do_split=True, squeeze_type=0)
    C.perturb_(blk, 5, 0.02, 0.05, 0.02)
    blk.to(DEV)
    g = torch.Generator().manual_seed(9)
    z = torch.randn(B, 2 * cin, hs, ws, generator=g).to(DEV)
    cond = torch.randn(B, 32, hs, ws, generator=g).to(DEV)
    hst = torch.randn(B, 64, hs, ws, generator=g).to(DEV)
    cst = torch.randn(B, 64, hs, ws, generator=g).to(DEV)
    eps = torch.randn(B, 2 * cin, hs, ws, generator=g).to(DEV)
    res = {}
    for tag, env in (("fused", None), ("plain", "1")):
        if env:
            os.environ["TMG_NO_LEVEL_FUSION"] = env
        else:
            os.environ.pop("TMG_NO_LEVEL_FUSION", None)
        blk.zero_grad()
        zi, ci, hi, cc = (t.clone().requires_grad_(True) for t in (z, cond, hst, cst))
        xr, ldr, st = blk.reverse(zi, ci, (hi, cc), eps=eps)
        ((xr[:2] ** 2).sum() * 0.5 + ldr[:2].sum() * 0.02 + (st[0][:2] ** 2).sum() * 0.1).backward()
        res[tag] = (xr.detach(), ldr.detach(), {**{k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None},
                                                 "@dz": zi.grad.clone(), "@dcond": ci.grad.clone(), "@dh": hi.grad.clone(), "@dc": cc.grad.clone()})
    os.environ.pop("TMG_NO_LEVEL_FUSION", None)
    a, b = res["fused"], res["plain"]
    ex = float((a[0] - b[0]).abs().max() / b[0].abs().max())
    worst, wk = 0.0, None
    num = den = 0.0
    for k in b[2]:
        d = (a[2][k].double() - b[2][k].double())
        sc = float(b[2][k].abs().max())
        num += float((d ** 2).sum()); den += float((b[2][k].double() ** 2).sum())
        if sc > 0 and float(d.abs().max()) / sc > worst:
            worst, wk = float(d.abs().max()) / sc, k
    glob = (num / max(den, 1e-300)) ** 0.5
    flag = "" if ex < 1e-4 and glob < 2e-4 and worst < 5e-3 else "   <<<<<< BAD"
    bad += bool(flag)
    print("C=%3d %3dx%-3d B=%3d  x %.1e  grads global %.1e worst %.1e (%s)%s" % (4 * cin, hs, ws, B, ex, glob, worst, wk, flag), flush=True)
print("bad:", bad)
