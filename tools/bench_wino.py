#!/usr/bin/env python3
"""Winograd vs direct 3x3 contraction on the path's widest shapes (GPU only): milliseconds and direct-algorithm TFLOP/s."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import tmg_hip as H  # noqa: E402

dev = torch.device("cuda")
CASES = [(64, 128, 128, [8, 32, 64], 256, False, False, True), (64, 128, 128, [32], 240, True, True, False),
         (64, 64, 64, [16, 32, 64], 256, False, False, True), (64, 64, 64, [32], 480, True, True, False),
         (64, 32, 32, [32], 960, True, True, False), (64, 16, 16, [32], 1920, True, True, False),
         # 64..128 output channels (one tile per wave): the ConvLSTM block's out-conv input gradients
         (64, 128, 128, [40], 104, False, False, False), (64, 64, 64, [48], 112, False, False, False), (64, 32, 32, [64], 128, False, False, False)]
for B, Hh, Ww, segs, Cout, relu, rep, hb in CASES:
    xs = [torch.randn(B, Hh, Ww, c, device=dev) for c in segs]
    w = 0.1 * torch.randn(Cout, sum(segs), 3, 3, device=dev)
    b = torch.randn(Cout, device=dev) if hb else None
    out = torch.empty(B, Hh, Ww, Cout, device=dev)
    U, Wp = H.conv_wino_pack(w), H.conv_pack(w, 0)
    fl = 2.0 * B * Hh * Ww * Cout * sum(segs) * 9

    def t(fn, n=10):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n
    U3 = H.conv_wino_pack3(w)
    # alternating rounds in one process (fp32 MFMA Winograd, the bf16x3 variant, the direct kernel)
    tws, t3s, tds = [], [], []
    for _ in range(3):
        tws.append(t(lambda: H.conv_wino_fwd(xs, U, Cout, [out], bias=b, relu_in=relu, pad_rep=rep)))
        t3s.append(t(lambda: H.conv_wino_fwd3(xs, U3, Cout, [out], bias=b, relu_in=relu, pad_rep=rep)))
        tds.append(t(lambda: H.conv_fwd(xs, Wp, Cout, 3, 1, [out], bias=b, relu_in=relu, pad_rep=rep)))
    tw, t3, td = sorted(tws)[1], sorted(t3s)[1], sorted(tds)[1]
    print("%4dx%-4d %4d -> %4d   winograd fp32 %7.3f ms (%6.1f TF)   bf16x3 %7.3f ms (%6.1f TF, %.2fx)   direct %7.3f ms (%6.1f TF)" % (
        Hh, Ww, sum(segs), Cout, tw, fl / tw / 1e9, t3, fl / t3 / 1e9, tw / t3, td, fl / td / 1e9))
    if os.environ.get("TMG_BENCH_WINO_WIDE_ONLY") and (B, Hh, Ww, segs, Cout, relu, rep, hb) == CASES[-1]:
        sys.exit(0)

print("few output channels (input gradients):")
for B, Hh, Ww, K, N in [(64, 128, 128, 256, 40), (64, 128, 128, 240, 32), (64, 64, 64, 256, 48), (64, 64, 64, 480, 32), (64, 32, 32, 960, 32),
                        (64, 16, 16, 1920, 32), (64, 128, 128, 104, 40)]:
    x = torch.randn(B, Hh, Ww, K, device=dev)
    w = 0.1 * torch.randn(K, N, 3, 3, device=dev)      # forward weight [Cout = K][Cin = N]: dgrad contracts K -> N
    out = torch.empty(B, Hh, Ww, N, device=dev)
    U, Wp = H.conv_wino_pack(w, 1), H.conv_pack(w, 1)
    fl = 2.0 * B * Hh * Ww * K * N * 9

    def t(fn, n=10):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n
    U3 = H.conv_wino_pack3(w, 1)
    tws, t3s = [], []
    for _ in range(3):
        tws.append(t(lambda: H.conv_wino_narrow([x], U, N, [out])))
        t3s.append(t(lambda: H.conv_wino_fwd3([x], U3, N, [out])) if N >= 64 else float("nan"))
    tw, t3 = sorted(tws)[1], sorted(t3s)[1]
    td = t(lambda: H.conv_fwd([x], Wp, N, 3, 1, [out]))
    print("%4dx%-4d %4d -> %4d   winograd (few-output kernel, fp32) %7.3f ms (%6.1f TF)   bf16x3 wide kernel %7.3f ms (%6.1f TF, %.2fx)   direct %7.3f ms (%6.1f TF)" % (
        Hh, Ww, K, N, tw, fl / tw / 1e9, t3, fl / t3 / 1e9, tw / t3, td, fl / td / 1e9))

print("weight gradients:")
for B, Hh, Ww, segs, Cout in [(64, 128, 128, [8, 32, 64], 256), (64, 128, 128, [32], 240), (64, 64, 64, [16, 32, 64], 256), (64, 64, 64, [32], 480),
                              (64, 32, 32, [32], 960), (64, 16, 16, [32], 1920), (64, 128, 128, [8, 32, 64], 40)]:
    xs = [torch.randn(B, Hh, Ww, c, device=dev) for c in segs]
    dy = torch.randn(B, Hh, Ww, Cout, device=dev)
    dW = torch.zeros(Cout, sum(segs), 3, 3, device=dev)
    fl = 2.0 * B * Hh * Ww * Cout * sum(segs) * 9

    def t(fn, n=10):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n
    tw = t(lambda: H.conv_wino_wgrad(xs, dy, dW, None))
    os.environ["TMG_NO_WINOGRAD"] = "1"
    td = t(lambda: H.conv_wgrad(xs, dy, dW, None, 3, 1))
    del os.environ["TMG_NO_WINOGRAD"]
    print("%4dx%-4d %4d -> %4d   winograd %7.3f ms (%6.1f TF)   direct %7.3f ms (%6.1f TF)" % (Hh, Ww, sum(segs), Cout, tw, fl / tw / 1e9, td, fl / td / 1e9))
