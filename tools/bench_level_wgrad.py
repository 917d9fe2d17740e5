#!/usr/bin/env python3
"""The per-layer weight gradients of a narrow level: tmg_level_wgrad_merged (one launch) against the three grouped launches it
replaces, at the metric configuration's first two levels (GPU only).  TMG_LW_DBG=1|2 (ablation builds: skip the MFMA phase | the
staging) give wrong results."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-turbulence_amd"))
import torch  # noqa: E402
import tmg_hip as H  # noqa: E402

dev = torch.device("cuda")


def run(B, Hh, Ww, ch, G=15, Cc=32, reps=10):
    C = 2 * ch
    cin = ch + Cc
    rnd = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    x1s = [rnd(B, Hh, Ww, ch) for _ in range(G)]
    Ds = [rnd(B, Hh, Ww, 4) for _ in range(G)]
    y2s = [rnd(B, Hh, Ww, ch) for _ in range(G)]
    douts = [(rnd(B, Hh, Ww, ch), rnd(B, Hh, Ww, ch)) for _ in range(G)]
    DH = rnd(B, Hh, Ww, G * C)
    DD = rnd(B, Hh, Ww, 2 * ((G + 3) // 4 * 4))
    wg_in = [[a, d] for a, d in zip(x1s, Ds)]
    mix_wg = [([a, y2], d) for a, y2, d in zip(x1s, y2s, douts)]
    bufs = (torch.zeros(G, C, cin + 2, 3, 3, device=dev), torch.zeros(G, C, device=dev), torch.zeros(G, 4, ch + 4, 3, 3, device=dev),
            torch.zeros(G, C, C, device=dev), torch.zeros(G, C, device=dev))

    def merged():
        if not H.level_wgrad_merged(wg_in, mix_wg, DH, DD, C, *bufs, Cc):
            raise SystemExit("outside the kernel envelope")

    def three():
        assert H.conv_wgrad_grouped(wg_in, DH, C, bufs[0], bufs[1], 3, 1, relu_in=True, pad_rep=True, cin_dst=cin + 2, cin_valid=ch + 2,
                                    ci_split=ch, ci_off0=0, ci_off1=Cc)
        assert H.conv_wgrad_grouped(wg_in, DD, 2, bufs[2], None, 3, 1, relu_in=True)
        assert H.conv_wgrad_grouped([a for a, _ in mix_wg], None, C, bufs[3].view(G, C, C, 1, 1), bufs[4], 1, 1, group_dy=douts)

    npix = B * Hh * Ww
    alg = npix * G * (ch + 4 + ch + 2 * C + 2 + C) * 4
    for name, fn in (("one launch", merged), ("three launches", three)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%dx%dx%d ch %2d  %-15s %8.3f ms  (%.1f us / layer, %.2f TB/s of the one-read bytes)" % (B, Hh, Ww, ch, name, ms, ms * 1e3 / G,
                                                                                                  alg / ms / 1e9))


if __name__ == "__main__":
    run(64, 128, 128, 8)
    if len(sys.argv) > 1:
        run(64, 64, 64, 16)
