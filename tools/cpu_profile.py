#!/usr/bin/env python3
"""Host-side cost of one training step at a small batch (where the step is launch-bound): cProfile of the enqueueing thread, GPU only."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
argv = sys.argv[1:]
sys.argv = [sys.argv[0]]
import torch  # noqa: E402
import bench  # noqa: E402

for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import common as C  # noqa: E402
from tmg_optim import HipAdam  # noqa: E402

B = int(argv[0]) if argv else 8
cfg = bench.CONFIGS["M"]
dev = torch.device("cuda")
model = bench.build_model(cfg, dev)
opt = HipAdam(model.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
h, w = cfg["_in_hw"]
x = torch.randn(B, cfg["in_features"], h, w, device=dev)
states = model.initLSTMStates(torch.arange(B), [h * 2, w * 2])
states = [(a.contiguous(memory_format=torch.channels_last), b.contiguous(memory_format=torch.channels_last)) for a, b in states]


def step():
    opt.zero_grad(set_to_none=True)
    y, ld, _ = model.sample(x, states)
    C.loss_reverse(y, ld).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue time per step %.2f ms; wall per step incl. drain %.2f ms" % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
