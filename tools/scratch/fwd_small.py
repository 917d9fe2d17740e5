import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd")):
    sys.path.insert(0, p)
import torch
import tmg_hip as H
dev = torch.device("cuda")
# (B, H, W, segs, Cout, dgrad-like?) : the per-layer zero convs of the wide levels, forward and input gradient
for name, B, Hh, Ww, segs, Cout in [("L4 fwd 68->128", 64, 16, 16, [64, 4], 128), ("L4 dgrad 128->68", 64, 16, 16, [128], 68), ("L3 fwd 36->64", 64, 32, 32, [32, 4], 64), ("L3 dgrad 64->36", 64, 32, 32, [64], 36),
                                  ("L4 lstm zero 98->128", 64, 16, 16, [96, 4], 128)]:
    xs = [torch.randn(B, Hh, Ww, c, device=dev) for c in segs]
    w = 0.1 * torch.randn(Cout, sum(segs), 3, 3, device=dev)
    out = torch.empty(B, Hh, Ww, Cout, device=dev)
    Wp = H.conv_pack(w, 0)
    fl = 2.0 * B * Hh * Ww * Cout * sum(segs) * 9
    fn = lambda: H.conv_fwd(xs, Wp, Cout, 3, 1, [out], relu_in=True, pad_rep=True)
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / 50
    print("MINBLK=%s  %-22s %.1f us  %.1f TF" % (os.environ.get("TMG_FWD_MINBLK", "256"), name, 1e3 * t, fl / t / 1e9))
