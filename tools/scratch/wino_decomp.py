import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd")):
    sys.path.insert(0, p)
if len(sys.argv) > 1:
    import torch
    import tmg_hip as H
    dev = torch.device("cuda")
    for B, Hh, Ww, segs, Cout, relu, rep, hb in [(64, 128, 128, [8, 32, 64], 256, False, False, True), (64, 128, 128, [32], 240, True, True, False), (64, 64, 64, [16, 32, 64], 256, False, False, True)]:
        xs = [torch.randn(B, Hh, Ww, c, device=dev) for c in segs]
        w = 0.1 * torch.randn(Cout, sum(segs), 3, 3, device=dev)
        b = torch.randn(Cout, device=dev) if hb else None
        out = torch.empty(B, Hh, Ww, Cout, device=dev)
        U = H.conv_wino_pack(w)
        fn = lambda: H.conv_wino_fwd(xs, U, Cout, [out], bias=b, relu_in=relu, pad_rep=rep)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); e1.synchronize()
        print("dbg=%s  %dx%d %d->%d  %.3f ms" % (os.environ.get("TMG_WINO_DBG", "0"), Hh, Ww, sum(segs), Cout, e0.elapsed_time(e1) / 10))
else:
    for d in ("0", "1", "2", "3"):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, TMG_WINO_DBG=d))
