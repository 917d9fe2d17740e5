"""Diagnostic build (TMG_EXTRA_DEFS=-DTMG_WINO_STAMP): cycles per phase of wino_fwd_kernel, summed over waves."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd")):
    sys.path.insert(0, p)
import torch
import tmg_hip as H
lib = H.lib()
dev = torch.device("cuda")
names = ["commit", "issue", "transform", "barrier1", "mfma loop", "epilogue", "barrier2", "loop head"]
for B, Hh, Ww, segs, Cout, relu, rep, hb in [(64, 128, 128, [8, 32, 64], 256, False, False, True), (64, 128, 128, [32], 240, True, True, False), (64, 64, 64, [16, 32, 64], 256, False, False, True)]:
    xs = [torch.randn(B, Hh, Ww, c, device=dev) for c in segs]
    w = 0.1 * torch.randn(Cout, sum(segs), 3, 3, device=dev)
    b = torch.randn(Cout, device=dev) if hb else None
    out = torch.empty(B, Hh, Ww, Cout, device=dev)
    U = H.conv_wino_pack(w)
    fn = lambda: H.conv_wino_fwd(xs, U, Cout, [out], bias=b, relu_in=relu, pad_rep=rep)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    lib.tmg_wino_stamps(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 5
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    lib.tmg_wino_stamps(buf, 0)
    v = [buf[i] for i in range(9)]
    tot = sum(v[:8])
    nblk = v[8]
    print("%dx%d %d->%d  %.3f ms/launch (stamped build); blocks %d; mean cycles per wave per launch %.0f" % (Hh, Ww, sum(segs), Cout, e0.elapsed_time(e1) / n, nblk // n, tot / (8.0 * nblk)))
    for nm, x in zip(names, v[:8]):
        print("     %-10s %5.1f %%" % (nm, 100.0 * x / tot))
