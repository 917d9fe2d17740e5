#!/usr/bin/env python3
"""Per-level forward (sample direction) time and forward/backward split at the metric configuration."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import common as C  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
cfg = C.CFG_M
model = bench.build_model(cfg, dev)
B = 64
x = torch.randn(B, 4, 128, 128, device=dev)
states = model.initLSTMStates(torch.arange(B), [256, 256])
marks = []
glow = model.glow
orig = [blk.reverse for blk in glow.flow_blocks]
for i, blk in enumerate(glow.flow_blocks):
    def wrap(f, i=i):
        def g(*a, **k):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); out = f(*a, **k); e1.record(); marks.append((i, e0, e1)); return out
        return g
    blk.reverse = wrap(blk.reverse)
for it in range(3):
    marks.clear()
    ea, eb, ec = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    ea.record()
    y, ld, _ = model.sample(x, states)
    loss = C.loss_reverse(y, ld)
    eb.record()
    loss.backward()
    ec.record()
    torch.cuda.synchronize()
    model.zero_grad(set_to_none=True)
print("forward %.2f ms  backward %.2f ms" % (ea.elapsed_time(eb), eb.elapsed_time(ec)))
for i, e0, e1 in marks:
    print("  level %d reverse (forward pass): %.2f ms" % (i + 1, e0.elapsed_time(e1)))
