#!/usr/bin/env python3
"""The stand-alone 1x1 channel mix with fp32 operands (mix32_kernel / the general 1x1 conv above 128 channels) and with fp16 operands
(mix16_kernel) at cfg5's level shapes, batch 64 (GPU only): microseconds per launch and GB/s of the 8 C bytes per pixel both kernels
move.  Under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` the same script gives the HBM bytes per launch (tools/pmc_summary.py)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import tmg_hip as H  # noqa: E402

dev = torch.device("cuda")
B = 64
for hw, C in ((256, 16), (128, 32), (64, 64), (32, 128), (16, 256)):
    x = torch.randn(B, hw, hw, C, device=dev)
    y = torch.empty_like(x)
    W = torch.randn(C, C, device=dev) / C ** 0.5
    b = torch.randn(C, device=dev)
    pk = H.conv_pack(W.reshape(C, C, 1, 1), 0)

    def f32():
        if not H.mix_f32(x, W, b, y):
            H.conv_fwd([x], pk, C, 1, 1, [y], bias=b)

    def f16():
        H.mix_f16(x, W, b, y)

    def t(fn, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    a = sorted(t(f32) for _ in range(3))[1]
    c = sorted(t(f16) for _ in range(3))[1]
    nbytes = 8.0 * C * B * hw * hw
    print("%3dx%-3d C = %3d   fp32 operands %7.1f us (%5.0f GB/s)   fp16 operands %7.1f us (%5.0f GB/s)   fp16 / fp32 time %.2f" % (
        hw, hw, C, a, nbytes / a / 1e3, c, nbytes / c / 1e3, c / a))
