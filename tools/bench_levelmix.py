#!/usr/bin/env python3
"""GPU time and launch count of the parameter-side folding (ActNorm + PLU -> [K,C,C] mix matrices) of all levels, forward + backward."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]]
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

dev = torch.device("cuda")
model = bench.build_model(bench.CONFIGS["M"], dev)
blocks = list(model.glow.flow_blocks)


def run():
    tot = 0
    for i, b in enumerate(blocks):
        Wm, bm, ld = b._level_mix(True, 100)[:3]
        tot = tot + (Wm * Wm).sum() + (bm * bm).sum() + ld
    tot.backward()


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
e1.synchronize()
print("level-mix folding, 4 levels fwd+bwd: %.3f ms per step (wall on stream)" % (e0.elapsed_time(e1) / 10))
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    run()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_time_total > 0]
print("kernel launches:", len(ev), " kernel time %.3f ms" % (sum(e.device_time_total for e in ev) / 1e3))
import collections
c = collections.Counter()
for e in ev:
    c[e.name[:90]] += 1
for k, v in c.most_common(14):
    print("%5d  %s" % (v, k))
