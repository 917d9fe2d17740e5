import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd")):
    sys.path.insert(0, p)
import torch
import tmg_hip as H
dev = "cuda"
torch.manual_seed(0)
for (B, Hh, Ww, C) in [(32, 8, 16, 128), (64, 8, 16, 128), (64, 16, 16, 128), (64, 16, 32, 64), (48, 8, 16, 128), (33, 8, 16, 128), (64, 8, 16, 64), (128, 8, 16, 128)]:
    x = torch.randn(B, Hh, Ww, C, device=dev)
    dy = torch.randn(B, Hh, Ww, C, device=dev)
    dy[2:] = 0            # as in the stated-batch test: upstream gradient on two samples only
    ref = torch.einsum("bhwo,bhwi->oi", dy.double(), x.double())
    refb = dy.double().sum((0, 1, 2))
    for zero_tail in (True, False):
        d = dy if zero_tail else torch.randn(B, Hh, Ww, C, device=dev)
        r = torch.einsum("bhwo,bhwi->oi", d.double(), x.double())
        rb = d.double().sum((0, 1, 2))
        dW = torch.zeros(C, C, 1, 1, device=dev); db = torch.zeros(C, device=dev)
        H.conv_wgrad([x], d, dW, db, 1, 1)
        e1 = float((dW.view(C, C).double() - r).abs().max() / r.abs().max())
        eb = float((db.double() - rb).abs().max() / rb.abs().max())
        # two-segment input as the level node passes it
        dW2 = torch.zeros(C, C, 1, 1, device=dev); db2 = torch.zeros(C, device=dev)
        H.conv_wgrad([x[..., :C // 2], x[..., C // 2:]], d, dW2, db2, 1, 1)
        e2 = float((dW2.view(C, C).double() - r).abs().max() / r.abs().max())
        # grouped (3 groups, each its own dy)
        G = 3
        xs = [torch.randn(B, Hh, Ww, C, device=dev) for _ in range(G)]
        ds = [d.clone() for _ in range(G)]
        dWg = torch.zeros(G, C, C, 1, 1, device=dev); dbg = torch.zeros(G, C, device=dev)
        ok = H.conv_wgrad_grouped([[t] for t in xs], None, C, dWg, dbg, 1, 1, group_dy=ds)
        eg = -1.0
        if ok:
            eg = max(float((dWg[k].view(C, C).double() - torch.einsum("bhwo,bhwi->oi", ds[k].double(), xs[k].double())).abs().max() / r.abs().max()) for k in range(G))
        print("B%3d %2dx%-2d C%3d zero_tail=%d  single %.2e (bias %.2e)  two-seg %.2e  grouped %s %.2e" % (B, Hh, Ww, C, zero_tail, e1, eb, e2, ok, eg))
