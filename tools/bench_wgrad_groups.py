#!/usr/bin/env python3
"""Micro-benchmark of the narrow levels' grouped weight-gradient launches (one launch per level for all 15 coupling layers) at the
metric configuration's first two levels (B 64: 128^2 x 16 channels, 64^2 x 32 channels) - GPU only: the zero convs' (x1 | D rows,
replicate padding), the growth layers' (4 output channels per group) and the 1x1 mixes'.  Algorithmic bytes = every operand read once.

  TMG_WG_PLAN=MPIXMAX,GXMUL   plan override of conv_wgrad_kernel (largest pixel tile, multiplier of the pixel-share count)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import tmg_hip as H  # noqa: E402
import tmg_ops as ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(2):
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
    tag = os.environ.get("TMG_WG_PLAN", "default plan")
    for lvl, (hw, C) in enumerate([(128, 16), (64, 32)], 1):
        ch = C // 2
        cin = ch + Cc
        npx = B * hw * hw
        mk = lambda c: torch.randn(B, hw, hw, c, device=dev)  # noqa: E731
        wg_in = [[mk(ch), mk(4)] for _ in range(NL)]
        DH, DD = mk(NL * C), mk(4 * NL)
        dWz = ops.zeros((NL, C, cin + 2, 3, 3), dev)
        dBz = ops.zeros((NL, C), dev)
        t = timeit(lambda: H.conv_wgrad_grouped(wg_in, DH, C, dWz, dBz, 3, 1, relu_in=True, pad_rep=True, cin_dst=cin + 2, cin_valid=ch + 2,
                                                ci_split=ch, ci_off0=0, ci_off1=Cc))
        byts = npx * 4.0 * NL * (ch + 4 + C)
        print("[%s] L%d zero-conv weight gradients (15 groups)   %7.1f us  alg %.2f TB/s  %5.1f TF" % (tag, lvl, t, byts / t / 1e6, 2.0 * npx * NL * (ch + 4) * C * 9 / t / 1e6), flush=True)
        tmpX = ops.zeros((NL, 4, ch + 4, 3, 3), dev)
        t = timeit(lambda: H.conv_wgrad_grouped(wg_in, DD, 4, tmpX, None, 3, 1, relu_in=True))
        byts = npx * 4.0 * NL * (ch + 4 + 4)
        print("[%s] L%d growth-layer weight gradients (15 groups) %7.1f us  alg %.2f TB/s" % (tag, lvl, t, byts / t / 1e6), flush=True)
        ys = [(mk(ch), mk(ch)) for _ in range(NL)]
        gs = [(mk(ch), mk(ch)) for _ in range(NL)]
        dWm = ops.zeros((NL, C, C, 1, 1), dev)
        dbm = ops.zeros((NL, C), dev)
        t = timeit(lambda: H.conv_wgrad_grouped([list(y) for y in ys], None, C, dWm, dbm, 1, 1, group_dy=gs))
        byts = npx * 4.0 * NL * 2 * C
        print("[%s] L%d 1x1 mix weight gradients (15 groups)      %7.1f us  alg %.2f TB/s" % (tag, lvl, t, byts / t / 1e6), flush=True)


if __name__ == "__main__":
    main()
