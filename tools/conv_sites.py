#!/usr/bin/env python3
"""Shapes, call sites and event-timed durations of every conv / weight-gradient launch of one config-M training step (GPU only)."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
argv = sys.argv[1:]
sys.argv = [sys.argv[0]]
import torch  # noqa: E402
import bench  # noqa: E402

for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import common as C  # noqa: E402
import tmg_hip as H  # noqa: E402

B = int(argv[0]) if argv else 64
cfg = bench.CONFIGS["M"]
dev = torch.device("cuda")
model = bench.build_model(cfg, dev)
h, w = cfg["_in_hw"]
x = torch.randn(B, cfg["in_features"], h, w, device=dev)
states = model.initLSTMStates(torch.arange(B), [h * 2, w * 2])


def step():
    model.zero_grad(set_to_none=True)
    y, ld, _ = model.sample(x, states)
    C.loss_reverse(y, ld).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
rec = collections.OrderedDict()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "tmg_hip" not in fr.filename and "conv_sites" not in fr.filename and "torch/" not in fr.filename:
            return "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
    return "?"


def wrap(name, describe):
    orig = getattr(H, name)

    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(*a, **k)
        e1.record()
        e1.synchronize()
        key = (name, describe(*a, **k), site())
        v = rec.setdefault(key, [0, 0.0])
        v[0] += 1
        v[1] += e0.elapsed_time(e1)
        return r
    setattr(H, name, f)


def shp(t):
    if t is None:
        return "-"
    if isinstance(t, (list, tuple)):
        return "(" + ",".join(shp(u) for u in t) + ")"
    return "x".join(str(s) for s in t.shape)


wrap("conv_fwd", lambda inputs, wpk, Cout, ksize, stride, outs, **k: "in[%s] -> %d k%d s%d %s" % (
    ",".join(shp(t) for t in inputs), Cout, ksize, stride, " ".join(sorted(kk for kk, vv in k.items() if vv is not None and vv is not False))))
wrap("conv_wgrad", lambda inputs, dy, dW, dbias, ksize, stride, **k: "in[%s] dy %s k%d" % (",".join(shp(t) for t in inputs), shp(dy), ksize))
wrap("conv_wgrad_grouped", lambda gi, dy, gc, dW, dbias, ksize, stride, **k: "groups %d in[%s] dy %s gc %s k%d" % (
    len(gi), ",".join(shp(t) for t in gi[0]), shp(dy), gc, ksize))
wrap("conv_rep_border_fix", lambda dy, w, outs, **k: "dy %s" % shp(dy))
wrap("dense2_bwd", lambda inputs, *a, **k: "in[%s]" % ",".join(shp(t) for t in inputs))
wrap("c1x2_fwd", lambda inputs, *a, **k: "in[%s]" % ",".join(shp(t) for t in inputs))
wrap("coupling_fwd", lambda x, *a, **k: shp(x))
wrap("coupling_bwd", lambda x, *a, **k: shp(x))
wrap("conv_wino_fwd", lambda inputs, U, Cout, outs, **k: "in[%s] -> %d %s" % (",".join(shp(t) for t in inputs), Cout, " ".join(sorted(kk for kk, vv in k.items() if vv is not None and vv is not False))))
wrap("conv_wino_narrow", lambda inputs, U, Cout, outs, **k: "in[%s] -> %d %s" % (",".join(shp(t) for t in inputs), Cout, " ".join(sorted(kk for kk, vv in k.items() if vv is not None and vv is not False))))


def first_shape(*a, **k):
    for v in list(a) + list(k.values()):
        if isinstance(v, torch.Tensor):
            return shp(v)
        if isinstance(v, (list, tuple)) and v and isinstance(v[0], torch.Tensor):
            return shp(v[0])
    return "-"


for nm in ("affine_apply", "affine_bwd", "lstm_pointwise_fwd", "lstm_pointwise_bwd", "gauss_fwd", "gauss_bwd", "checker", "upsample_fwd",
           "upsample_bwd", "chan_reduce", "bn_bwd_apply", "masked_add", "c1_fwd", "c1_bwd", "dkappa", "mix_f16", "conv_dgrad_direct",
           "conv_pack", "conv_pack_batched", "mix_affine_fwd", "mix_affine_bwd", "mix_f32", "layer_planes", "pad_halves", "lu_fold_fwd", "lu_fold_bwd",
           "conv_wino_pack", "conv_pack_many", "chan_moments"):
    wrap(nm, first_shape)
step()
torch.cuda.synchronize()
lev = collections.OrderedDict()
for (name, d, s_), (n, t) in rec.items():
    import re
    m = re.search(r"%dx(\d+)x(\d+)x" % B, d)
    key = (m.group(1) if m else "-", name)
    v = lev.setdefault(key, [0, 0.0])
    v[0] += n
    v[1] += t
bylev = collections.defaultdict(float)
for (hh, name), (n, t) in lev.items():
    bylev[hh] += t
print("by spatial size:", {k: round(v, 2) for k, v in bylev.items()})
for (hh, name), (n, t) in sorted(lev.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
    print("   H=%-4s %-22s %4d launches %8.3f ms" % (hh, name, n, t))
tot = sum(v[1] for v in rec.values())
print("total %.2f ms in %d launches" % (tot, sum(v[0] for v in rec.values())))
for (name, d, s), (n, t) in sorted(rec.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOP", 70))]:
    print("%7.3f ms %3d x %8.1f us  %-20s %-34s %s" % (t, n, 1e3 * t / n, name, s, d[:150]))
