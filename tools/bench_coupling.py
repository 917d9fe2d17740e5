#!/usr/bin/env python3
"""Micro-benchmark of the fused coupling-layer kernel (tmg_coupling_fwd) at the metric configuration's narrow levels (GPU only)."""
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
    B, Cc, NL = 64, 32, 15
    for lvl, (hw, C) in enumerate([(128, 16), (64, 32)], 1):
        ch = C // 2
        cin = ch + Cc
        x = torch.randn(B, hw, hw, C, device=dev)
        Hc = torch.randn(B, hw, hw, NL * C, device=dev)
        D = torch.randn(B, hw, hw, 4, device=dev)
        wz = 0.02 * torch.randn(C, cin + 2, 3, 3, device=dev)
        bz = torch.zeros(C, device=dev)
        kap = torch.zeros(1, 1, 1, 1, device=dev)
        Wm = torch.randn(C, C, device=dev) / C ** 0.5
        bm = torch.randn(C, device=dev)
        out = torch.empty_like(x)
        r = torch.empty(B, hw, hw, ch, device=dev)
        y2 = torch.empty(B, hw, hw, ch, device=dev)
        ld = torch.zeros(B, device=dev)
        hc = Hc[..., 3 * C:4 * C]
        if os.environ.get("DENSE_HC"):
            hc = hc.contiguous()
        t = timeit(lambda: H.coupling_fwd(x, out, r, y2, D, hc, wz, bz, kap, Wm, bm, ld, True, ch + Cc))
        npx = B * hw * hw
        byts = npx * 4 * (C * 1.27 / 2 + C / 2 + 4 * 1.27 + C + C + C)   # x1 (+halo), x2, D (+halo), hc, out, r + y2
        print("L%d C=%d: %.1f us  %.2f TB/s (%.0f MB)" % (lvl, C, t, byts / t / 1e6, byts / 1e6), flush=True)


if __name__ == "__main__":
    main()
