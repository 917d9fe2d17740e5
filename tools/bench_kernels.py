#!/usr/bin/env python3
"""Micro-benchmark of the contraction kernels at the metric configuration's layer shapes (GPU only).
Prints achieved algorithmic TFLOP/s for forward / input-gradient / weight-gradient of each shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import tmg_hip as H  # noqa: E402

SHAPES = {
    # name: (B, H, W, seg channels, Cout, ksize, relu_in, pad_rep)
    "gate_L1": (64, 128, 128, [8, 32, 64], 256, 3, False, False),
    "gate_L2": (64, 64, 64, [16, 32, 64], 256, 3, False, False),
    "outc_L1": (64, 128, 128, [8, 32, 64], 40, 3, False, False),
    "zero_L1": (64, 128, 128, [8, 32, 4], 16, 3, True, True),
    "zero_L2": (64, 64, 64, [16, 32, 4], 32, 3, True, True),
    "zero_L3": (64, 32, 32, [32, 32, 4], 64, 3, True, True),
    "zero_L4": (64, 16, 16, [64, 32, 4], 128, 3, True, True),
    # per-layer coupling zero-conv of the level-fused path: x1 half + the 4-channel growth buffer (cond enters as `add`)
    "zl_L1": (64, 128, 128, [8, 4], 16, 3, True, True),
    "zl_L2": (64, 64, 64, [16, 4], 32, 3, True, True),
    "zl_L3": (64, 32, 32, [32, 4], 64, 3, True, True),
    "zl_L4": (64, 16, 16, [64, 4], 128, 3, True, True),
    "mix_L1": (64, 128, 128, [16], 16, 1, False, False),
    "mix_L4": (64, 16, 16, [128], 128, 1, False, False),
}


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    names = sys.argv[1:] or list(SHAPES)
    dev = "cuda"
    for name in names:
        B, Hh, Ww, segs, Cout, k, relu_in, pad_rep = SHAPES[name]
        xs = [torch.randn(B, Hh, Ww, c, device=dev) for c in segs]
        cin = sum(segs)
        w = 0.1 * torch.randn(Cout, cin, k, k, device=dev)
        out = torch.empty(B, Hh, Ww, Cout, device=dev)
        dy = torch.randn(B, Hh, Ww, Cout, device=dev)
        dxs = [torch.empty_like(t) for t in xs]
        dW = torch.zeros_like(w)
        wpk, wpk_t = H.conv_pack(w, 0), H.conv_pack(w, 1)
        fl = 2.0 * B * Hh * Ww * Cout * cin * k * k
        t_f = timeit(lambda: H.conv_fwd(xs, wpk, Cout, k, 1, [out], relu_in=relu_in, pad_rep=pad_rep))
        t_d = timeit(lambda: H.conv_fwd([dy], wpk_t, cin, k, 1, dxs))
        t_w = timeit(lambda: H.conv_wgrad(xs, dy, dW, None, k, 1, relu_in=relu_in, pad_rep=pad_rep))
        print("%-8s fwd %8.3f ms %6.1f TF | dgrad %8.3f ms %6.1f TF | wgrad %8.3f ms %6.1f TF" % (
            name, t_f, fl / t_f / 1e9, t_d, fl / t_d / 1e9, t_w, fl / t_w / 1e9), flush=True)


if __name__ == "__main__" and "tail" not in sys.argv and "d2" not in sys.argv and "d2f" not in sys.argv:
    main()


def bench_tail():
    """Coupling tail (dense2 + zero-conv + affine) forward / backward per level of the metric config."""
    import tmg_ops as ops
    dev = "cuda"
    only = [int(a[1:]) for a in sys.argv if a.startswith("L") and a[1:].isdigit()]
    for lvl, (hw, C) in enumerate([(128, 16), (64, 32), (32, 64), (16, 128)], 1):
        if only and lvl not in only:
            continue
        B, Cc = 64, 32
        ch = C // 2
        cin = ch + Cc
        x = torch.randn(B, hw, hw, C, device=dev, requires_grad=True)
        cond = torch.randn(B, hw, hw, Cc, device=dev, requires_grad=True)
        w1 = (0.1 * torch.randn(1, cin, 3, 3, device=dev)).requires_grad_(True)
        w2 = (0.1 * torch.randn(1, cin + 1, 3, 3, device=dev)).requires_grad_(True)
        wz = (0.02 * torch.randn(C, cin + 2, 3, 3, device=dev)).requires_grad_(True)
        bz = torch.zeros(C, device=dev, requires_grad=True)
        kap = torch.zeros(1, 1, 1, 1, device=dev, requires_grad=True)
        gy = torch.randn(B, hw, hw, C, device=dev)
        gl = torch.randn(B, device=dev)
        H.prof_enable(False)

        def fwd():
            return ops.CouplingTailFn.apply(x, cond, w1, w2, wz, bz, kap, True, 0)

        t_f = timeit(fwd)
        y, ld = fwd()

        def bwd():
            torch.autograd.grad([y, ld], [x, cond, w1, w2, wz, bz, kap], [gy, gl], retain_graph=True)

        t_b = timeit(bwd)
        print("tail L%d: fwd %7.3f ms  bwd %7.3f ms" % (lvl, t_f, t_b), flush=True)


if __name__ == "__main__" and "tail" in sys.argv:
    bench_tail()


def bench_d2():
    """dense2 backward alone at level-1 shape; TMG_D2_DBG=1 skips the input-gradient part, =2 the weight-gradient part."""
    dev = "cuda"
    for lvl, (hw, C) in enumerate([(128, 16), (64, 32), (32, 64), (16, 128)], 1):
        B, Cc = 64, 32
        ch, cin = C // 2, C // 2 + 32
        x = torch.randn(B, hw, hw, C, device=dev)
        cond = torch.randn(B, hw, hw, Cc, device=dev)
        D = torch.randn(B, hw, hw, 4, device=dev)
        GD = torch.randn(B, hw, hw, 4, device=dev)
        G = [torch.randn(B, hw, hw, ch, device=dev), torch.randn(B, hw, hw, Cc, device=dev)]
        dx = torch.empty(B, hw, hw, C, device=dev)
        dy = torch.randn(B, hw, hw, C, device=dev)
        w1, w2 = torch.randn(cin, 9, device=dev), torch.randn(cin + 1, 9, device=dev)
        dw1, dw2 = torch.zeros_like(w1), torch.zeros_like(w2)
        nn_in = [x[..., :ch], cond]
        for dbg in ("0", "1", "2", "3"):
            os.environ["TMG_D2_DBG"] = dbg
            t = timeit(lambda: H.dense2_bwd(nn_in + [D], w1, w2, dw1, dw2, GD, D, G, [dx[..., :ch], G[1]], cin, add0=dy[..., :ch], rows1=cin, rows2=cin + 1))
            print("dense2_bwd L%d dbg=%s: %7.3f ms" % (lvl, dbg, t), flush=True)
        os.environ["TMG_D2_DBG"] = "0"
        D0 = torch.zeros(B, hw, hw, 4, device=dev)
        t = timeit(lambda: H.c1_fwd(nn_in, w1, D0[..., 0:1], relu_in=True))
        print("c1_fwd L%d: %7.3f ms" % (lvl, t), flush=True)


if __name__ == "__main__" and "d2" in sys.argv:
    bench_d2()


def bench_d2_fused():
    """dense2 backward in the level-fused configuration (only the x1 half as input; conditioning handled level-wide)."""
    dev = "cuda"
    for lvl, (hw, C) in enumerate([(128, 16), (64, 32), (32, 64), (16, 128)], 1):
        B, Cc = 64, 32
        ch = C // 2
        cin = ch + Cc
        tin = torch.randn(B, hw, hw, C, device=dev)
        D = torch.randn(B, hw, hw, 4, device=dev)
        GD = torch.randn(B, hw, hw, 4, device=dev)
        G0 = torch.randn(B, hw, hw, ch, device=dev)
        dtin = torch.empty(B, hw, hw, C, device=dev)
        dto = torch.randn(B, hw, hw, C, device=dev)
        DD = torch.zeros(B, hw, hw, 32, device=dev)
        w1, w2 = torch.randn(1, cin, 3, 3, device=dev), torch.randn(1, cin + 1, 3, 3, device=dev)
        dW1, dW2 = torch.zeros_like(w1), torch.zeros_like(w2)
        x1 = tin[..., :ch]
        t = timeit(lambda: H.dense2_bwd([x1, D], w1, w2, dW1, dW2, GD, D, [G0], [dtin[..., :ch]], ch, add0=dto[..., :ch], rows1=ch,
                                        rows2=ch + 1, dd1=DD[..., 3:4], dd2=DD[..., 19:20], split2=ch, gap2=Cc))
        print("dense2_bwd (fused cfg) L%d: %7.3f ms" % (lvl, t), flush=True)


if __name__ == "__main__" and "d2f" in sys.argv:
    bench_d2_fused()
