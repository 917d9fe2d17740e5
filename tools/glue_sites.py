#!/usr/bin/env python3
"""Python source lines behind the small torch-native launches of one training step (fill / copy / add / cat ...), GPU only."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import torch  # noqa: E402
import bench  # noqa: E402

for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import common as C  # noqa: E402

cfg = bench.CONFIGS["M"]
dev = torch.device("cuda")
model = bench.build_model(cfg, dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
B = 8
h, w = cfg["_in_hw"]
x = torch.randn(B, cfg["in_features"], h, w, device=dev)
states = model.initLSTMStates(torch.arange(B), [h * 2, w * 2])


def step():
    opt.zero_grad(set_to_none=True)
    y, ld, _ = model.sample(x, states)
    C.loss_reverse(y, ld).backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::copy_", "aten::add_", "aten::add", "aten::cat", "aten::mul", "aten::zero_", "aten::index_put_", "aten::sum")
sites = collections.Counter()
for e in prof.events():
    if e.name in want and e.device_time_total > 0:
        st = [s for s in (e.stack or []) if "deep-turbulence_amd" in s or "bench" in s or "common.py" in s or "optim" in s]
        key = (e.name, st[0].split("/")[-1] if st else ("autograd/backward" if not e.stack else e.stack[0].split("/")[-1]))
        sites[key] += 1
for (name, where), n in sites.most_common(45):
    print("%5d  %-16s %s" % (n, name, where[:110]))

print("---- autograd nodes / aten ops of the step by call count")
ka = prof.key_averages()
for e in sorted(ka, key=lambda e: -e.count)[:70]:
    print("%5d  %-60s cpu %8.1f us  dev %8.1f us" % (e.count, e.key[:60], e.cpu_time_total, e.device_time_total))

print("---- parents of the fill / copy / cat launches")
chains = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::copy_", "aten::cat", "aten::add", "aten::mul", "aten::add_") and e.device_time_total > 0:
        names = []
        q = e.cpu_parent
        while q is not None and len(names) < 4:
            names.append(q.name[:48])
            q = q.cpu_parent
        chains[(e.name, " < ".join(names))] += 1
for (n, c), k in chains.most_common(40):
    print("%5d  %-12s %s" % (k, n, c))
