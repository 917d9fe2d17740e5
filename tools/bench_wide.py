#!/usr/bin/env python3
"""Micro-benchmark of the wide levels' per-layer launches (64 channels at 32^2, 128 channels at 16^2, 64 samples) - GPU only.

Generative direction, one coupling layer: c1x2_fwd -> conv_fwd (zero conv with the conditioning addend, bias, exp(kappa)) ->
mix_affine_fwd; backward: mix_affine_bwd -> conv_fwd (zero-conv input gradient) -> border fold -> dense2_bwd.
The matrix-pipe share of the two convs is printed against the fp32 MFMA peak (157.3 TF).

  TMG_FWD_PLAN=MT,WM,WN,NTW,GMUL,KCHMAX   launch-plan override of conv_fwd_kernel (read once per process by the library)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import tmg_hip as H  # noqa: E402


def timeit(fn, n=50):
    for _ in range(3):
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
    tag = os.environ.get("TMG_FWD_PLAN", "default plan")
    tot = 0.0
    for lvl, (hw, C) in ((3, (32, 64)), (4, (16, 128))):
        ch = C // 2
        cin = ch + Cc
        npx = B * hw * hw
        mk = lambda c: torch.randn(B, hw, hw, c, device=dev)  # noqa: E731
        tin, dcur = mk(C), mk(C)
        x1 = tin[..., :ch]
        D, GD, G0 = mk(4), mk(4), mk(ch)
        DD = torch.empty(B, hw, hw, 4 * NL, device=dev)
        Dc = torch.randn(B, hw, hw, 2 * 16, device=dev)
        Hc = torch.randn(B, hw, hw, NL * C, device=dev)
        DH = torch.empty(B, hw, hw, NL * C, device=dev)
        w1 = 0.1 * torch.randn(1, cin, 3, 3, device=dev)
        w2 = 0.1 * torch.randn(1, cin + 1, 3, 3, device=dev)
        Wz = 0.02 * torch.randn(NL, C, cin + 2, 3, 3, device=dev)
        bz = torch.zeros(C, device=dev)
        kap = torch.zeros(1, 1, 1, 1, device=dev)
        Wm = torch.randn(C, C, device=dev) / C ** 0.5
        bm = torch.randn(C, device=dev)
        PZ = H.conv_pack_batched(Wz, 0, ch + 4, (ch + 2, ch, Cc))
        PZt = H.conv_pack_batched(Wz, 1, ch + 4, (ch + 2, ch, Cc))
        k = 3
        hh = torch.empty(B, hw, hw, C, device=dev)
        out, r, y2 = torch.empty(B, hw, hw, C, device=dev), mk(ch), mk(ch)
        ld = torch.zeros(B, device=dev)
        g = torch.randn(B, device=dev)
        dto1, dtin = torch.empty(B, hw, hw, ch, device=dev), torch.empty(B, hw, hw, C, device=dev)
        dhh = DH[..., k * C:(k + 1) * C]
        rows = []
        t = timeit(lambda: H.c1x2_fwd([x1], w1, w2, D, w_rows=ch, w2_d1_row=ch + Cc, add1=Dc[..., 2 * k:2 * k + 1], add2=Dc[..., 2 * k + 1:2 * k + 2]))
        rows.append(("c1x2_fwd", t, None))
        t = timeit(lambda: H.conv_fwd([x1, D], PZ[k], C, 3, 1, [hh], bias=bz, kappa=kap, relu_in=True, pad_rep=True, add=Hc[..., k * C:(k + 1) * C]))
        rows.append(("conv_fwd zero conv %d+4 -> %d" % (ch, C), t, 2.0 * npx * (ch + 4) * C * 9))
        t = timeit(lambda: H.mix_affine_fwd(tin, hh, Wm, bm, out, r, y2, ld))
        rows.append(("mix_affine_fwd", t, 2.0 * npx * C * C))
        t = timeit(lambda: H.mix_affine_bwd(dcur, Wm, r, tin[..., ch:], g, kap, dto1, dtin[..., ch:], dhh))
        rows.append(("mix_affine_bwd", t, 2.0 * npx * C * C))
        t = timeit(lambda: H.conv_fwd([dhh], PZt[k], ch + 4, 3, 1, [G0, GD]))
        rows.append(("conv_fwd input gradient %d -> %d+4" % (C, ch), t, 2.0 * npx * (ch + 4) * C * 9))
        t = timeit(lambda: H.conv_rep_border_fix(dhh, PZt[k], [G0, GD]))
        rows.append(("border fold", t, None))
        t = timeit(lambda: H.dense2_bwd([x1, D], w1, w2, None, None, GD, D, [G0], [dtin[..., :ch]], ch, add0=dto1, rows1=ch, rows2=ch + 1,
                                        dd1=DD[..., 4 * k:4 * k + 1], dd2=DD[..., 4 * k + 1:4 * k + 2], split2=ch, gap2=Cc, dd_quad=True))
        rows.append(("dense2_bwd", t, None))
        if os.environ.get("WINO_PROBE"):
            # the same contractions through the Winograd kernels (no addend / exp(kappa): a probe of the kernels' speed on these shapes)
            wz_eff = torch.cat([Wz[k][:, :ch], Wz[k][:, cin:cin + 2], torch.zeros(C, 2, 3, 3, device=dev)], 1).contiguous()   # x1 | D rows
            t = timeit(lambda: H.conv3x3_auto([x1, D], wz_eff, C, [hh], bias=bz, relu_in=True, pad_rep=True))
            rows.append(("WINOGRAD probe %d+4 -> %d" % (ch, C), t, 2.0 * npx * (ch + 4) * C * 9))
            t = timeit(lambda: H.conv3x3_auto([dhh], wz_eff, ch + 4, [G0, GD], dgrad=True))
            rows.append(("WINOGRAD probe input gradient %d -> %d+4" % (C, ch), t, 2.0 * npx * (ch + 4) * C * 9))
        for name, t, fl in rows:
            print("[%s] L%d %-36s %7.1f us%s" % (tag, lvl, name, t, "" if fl is None else "  %5.1f TF = %.2f of the fp32 MFMA peak" % (fl / t / 1e6, fl / t / 1e6 / 157.3)),
                  flush=True)
        s = sum(t for _, t, _ in rows)
        tot += s
        print("[%s] L%d layer chain %.1f us (x %d layers = %.2f ms)" % (tag, lvl, s, NL, s * NL / 1e3), flush=True)
    print("[%s] both levels, per-layer launches: %.2f ms per step" % (tag, tot * NL / 1e3))


if __name__ == "__main__":
    main()
