#!/usr/bin/env python3
"""Which torch-native operators launch the small kernels of a step (count by aten op), to find fusable host-side glue."""
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
B = 16
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and (e.device_time_total > 0 or e.count > 20)]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    print("%-40s count %5d  device %8.1f us  cpu %8.1f us" % (e.key, e.count, e.device_time_total, e.cpu_time_total))
