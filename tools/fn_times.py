#!/usr/bin/env python3
"""GPU time of every custom autograd node (forward and backward) of one config-M training step, by node class and spatial size.
Events are recorded on the stream without host synchronisation, so the step runs as it does in bench.py (GPU only)."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
argv = sys.argv[1:]
sys.argv = [sys.argv[0]]
import torch  # noqa: E402
import bench  # noqa: E402

for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import common as C  # noqa: E402
import tmg_ops as O  # noqa: E402

B = int(argv[0]) if argv else 64
cfg = bench.CONFIGS["M"]
dev = torch.device("cuda")
model = bench.build_model(cfg, dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
h, w = cfg["_in_hw"]
x = torch.randn(B, cfg["in_features"], h, w, device=dev)
states = model.initLSTMStates(torch.arange(B), [h * 2, w * 2])
log = []
ON = [False]


def patch(cls):
    for which in ("forward", "backward"):
        orig = getattr(cls, which)

        def f(ctx, *a, _orig=orig, _which=which, _cls=cls):
            if not ON[0]:
                return _orig(ctx, *a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            hw = "-"
            for v in a:
                if isinstance(v, torch.Tensor) and v.dim() == 4:
                    hw = "%dx%d" % (tuple(v.shape[1:3]) if v.shape[1] == v.shape[2] or v.shape[3] < v.shape[1] else tuple(v.shape[2:4]))
                    break
            e0.record()
            r = _orig(ctx, *a)
            e1.record()
            log.append((_cls.__name__, _which, hw, e0, e1))
            return r
        setattr(cls, which, staticmethod(f))


for name in dir(O):
    c = getattr(O, name)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        patch(c)


def step():
    opt.zero_grad(set_to_none=True)
    y, ld, _ = model.sample(x, states)
    C.loss_reverse(y, ld).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
ON[0] = True
s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s0.record()
step()
s1.record()
torch.cuda.synchronize()
print("step %.2f ms" % s0.elapsed_time(s1))
agg = collections.OrderedDict()
for cls, which, hw, e0, e1 in log:
    v = agg.setdefault((cls, which, hw), [0, 0.0])
    v[0] += 1
    v[1] += e0.elapsed_time(e1)
print("inside custom nodes: %.2f ms" % sum(v[1] for v in agg.values()))
for (cls, which, hw), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%8.3f ms %4d x  %-22s %-9s %s" % (t, n, cls, which, hw))
