"""The 256-channel level (BASELINE configs[4]'s deepest) at batch 64: fused and per-layer HIP paths against the CPU oracle in fp64
and fp32 (is the fused-vs-per-layer difference of tools/scratch/level_sweep.py conditioning or a fault?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import common as C
import tmglow_oracle as O
from nn.modules.flowLSTMBlock import LSTMFLowBlock
DEV = "cuda"
for scales in ((0.02, 0.05, 0.02), (0.002, 0.01, 0.002)):
    cin, (hs, ws), B, K = 64, (8, 8), 64, 6
    C.seed_all(cin * 7 + B)
    blk = LSTMFLowBlock(cin, 32, 64, K, LUdecompose=True, train_sampling=True, do_split=True, squeeze_type=0)
    C.perturb_(blk, 5, *scales)
    sd = {"glow.flow_blocks.0." + k: v.clone() for k, v in blk.state_dict().items()}
    blk.to(DEV)
    g = torch.Generator().manual_seed(9)
    z = torch.randn(B, 2 * cin, hs, ws, generator=g)
    cond = torch.randn(B, 32, hs, ws, generator=g)
    hst = torch.randn(B, 64, hs, ws, generator=g)
    cst = torch.randn(B, 64, hs, ws, generator=g)
    eps = torch.randn(B, 2 * cin, hs, ws, generator=g)
    n = 2
    ref = {}
    for dt in (torch.float64, torch.float32):
        P = O.params_from_state_dict(sd, dtype=dt)
        zi = z[:n].to(dt).requires_grad_(True)
        xr, ld, st = O.flow_level_reverse(P, 0, K, zi, cond[:n].to(dt), (hst[:n].to(dt), cst[:n].to(dt)), eps[:n].to(dt))
        ((xr ** 2).sum() * 0.5 + ld.sum() * 0.02 + (st[0] ** 2).sum() * 0.1).backward()
        ref[dt] = (xr.detach(), {k[len("glow.flow_blocks.0."):]: v.grad.clone() for k, v in O.trainable(P).items() if v.grad is not None})
    res = {}
    for tag, env in (("fused", None), ("plain", "1")):
        if env:
            os.environ["TMG_NO_LEVEL_FUSION"] = env
        else:
            os.environ.pop("TMG_NO_LEVEL_FUSION", None)
        blk.zero_grad()
        xr, ldr, st = blk.reverse(z.to(DEV), cond.to(DEV), (hst.to(DEV), cst.to(DEV)), eps=eps.to(DEV))
        ((xr[:n] ** 2).sum() * 0.5 + ldr[:n].sum() * 0.02 + (st[0][:n] ** 2).sum() * 0.1).backward()
        res[tag] = (xr.detach()[:n].cpu(), {k: p.grad.detach().cpu().clone() for k, p in blk.named_parameters() if p.grad is not None})
    os.environ.pop("TMG_NO_LEVEL_FUSION", None)

    def gerr(got, r):
        num = den = worst = 0.0
        for k in r:
            d = got[k].double() - r[k].double()
            num += float((d ** 2).sum()); den += float((r[k].double() ** 2).sum())
            sc = float(r[k].abs().max())
            if sc > 0:
                worst = max(worst, float(d.abs().max()) / sc)
        return (num / den) ** 0.5, worst
    r64, r32 = ref[torch.float64], ref[torch.float32]
    xs = float(r64[0].abs().max())
    print("perturbation %s: |x|max %.2f" % (scales, xs))
    print("   fp32 oracle vs fp64: x %.1e  grads %.1e / %.1e" % ((float((r32[0].double() - r64[0]).abs().max()) / xs,) + gerr(r32[1], r64[1])))
    for tag in ("fused", "plain"):
        print("   HIP %-6s vs fp64: x %.1e  grads %.1e / %.1e" % ((tag, float((res[tag][0].double() - r64[0]).abs().max()) / xs) + gerr(res[tag][1], r64[1])))
