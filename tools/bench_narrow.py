#!/usr/bin/env python3
"""Micro-benchmark of the narrow levels' per-layer kernels (c1x2_fwd, dense2_bwd lean, cpl_fwd, cpl_bwd) at the metric
configuration's first two levels (B 64: 128^2 x 16 channels, 64^2 x 32 channels) - GPU only.

  LAYOUT=slice (default): x1 = the leading half of a [.., C] tensor (the layout of round 3)
  LAYOUT=split:           x1 / x2 in tensors of their own ([.., C/2] each)
  TMG_NO_XCD_MAP=1:       plain tile order (read once per process by the library)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import tmg_hip as H  # noqa: E402


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = "cuda"
    split = os.environ.get("LAYOUT", "slice") == "split"
    B, Cc, NL = 64, 32, 15
    tag = "%s, xcd map %s" % ("split halves" if split else "slices", "off" if os.environ.get("TMG_NO_XCD_MAP") else "on")
    for lvl, (hw, C) in enumerate([(128, 16), (64, 32)], 1):
        ch = C // 2
        cin = ch + Cc
        npx = B * hw * hw
        mk = lambda c: torch.randn(B, hw, hw, c, device=dev)  # noqa: E731
        full, dfull = mk(C), mk(C)
        if split:
            x1, x2, dt1 = mk(ch), mk(ch), mk(ch)
        else:
            x1, x2, dt1 = full[..., :ch], full[..., ch:], dfull[..., :ch]
        D, GD, G0 = mk(4), mk(4), mk(ch)
        DD = torch.empty(B, hw, hw, 4 * NL, device=dev)
        Dc = torch.randn(NL, B, hw, hw, 2, device=dev)
        w1 = 0.1 * torch.randn(1, cin, 3, 3, device=dev)
        w2 = 0.1 * torch.randn(1, cin + 1, 3, 3, device=dev)
        Dout = torch.empty(B, hw, hw, 4, device=dev)
        k = 3
        t = timeit(lambda: H.c1x2_fwd([x1], w1, w2, Dout, w_rows=ch, w2_d1_row=ch + Cc, add1=Dc[k][..., 0:1], add2=Dc[k][..., 1:2]))
        byts = npx * 4.0 * (ch + 2 + 4)
        print("[%s] L%d c1x2_fwd        %7.1f us  alg %.2f TB/s" % (tag, lvl, t, byts / t / 1e6), flush=True)
        t = timeit(lambda: H.dense2_bwd([x1, D], w1, w2, None, None, GD, D, [G0], [dt1], ch, add0=dt1, rows1=ch, rows2=ch + 1,
                                        dd1=DD[..., 4 * k:4 * k + 1], dd2=DD[..., 4 * k + 1:4 * k + 2], split2=ch, gap2=Cc, dd_quad=True))
        byts = npx * 4.0 * (ch + 4 + 4 + ch + ch + ch + 4)
        print("[%s] L%d dense2_bwd lean %7.1f us  alg %.2f TB/s" % (tag, lvl, t, byts / t / 1e6), flush=True)
        # fused coupling kernels
        Hc = torch.randn(B, hw, hw, NL * C, device=dev)
        wz = 0.02 * torch.randn(C, cin + 2, 3, 3, device=dev)
        bz = torch.zeros(C, device=dev)
        kap = torch.zeros(1, 1, 1, 1, device=dev)
        Wm = torch.randn(C, C, device=dev) / C ** 0.5
        bm = torch.randn(C, device=dev)
        xin = (x1, x2) if split else full
        out = (mk(ch), mk(ch)) if split else torch.empty_like(full)
        dout = (mk(ch), mk(ch)) if split else dfull
        dtin = (dt1, mk(ch)) if split else torch.empty_like(full)
        r, y2 = mk(ch), mk(ch)
        ld = torch.zeros(B, device=dev)
        for hname, hc in (("Hc slice", Hc[..., k * C:(k + 1) * C]), ("Hc dense", Hc[..., k * C:(k + 1) * C].contiguous())):
            t = timeit(lambda: H.coupling_fwd(xin, out, r, y2, D, hc, wz, bz, kap, Wm, bm, ld, True, ch + Cc))
            byts = npx * 4.0 * (C + 4 + C + C + ch + ch)
            print("[%s] L%d cpl_fwd (%s) %7.1f us  alg %.2f TB/s" % (tag, lvl, hname, t, byts / t / 1e6), flush=True)
        DH = torch.empty(B, hw, hw, NL * C, device=dev)
        g = torch.randn(B, device=dev)
        t = timeit(lambda: H.coupling_bwd(dout, x2, r, g, Wm, wz, kap, DH[..., k * C:(k + 1) * C], dtin, G0, GD, ch + Cc))
        byts = npx * 4.0 * (C + ch + ch + C + C + ch + 4)
        print("[%s] L%d cpl_bwd          %7.1f us  alg %.2f TB/s" % (tag, lvl, t, byts / t / 1e6), flush=True)

if __name__ == "__main__":
    main()
