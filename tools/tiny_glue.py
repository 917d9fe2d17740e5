#!/usr/bin/env python3
"""Census of the SMALL torch-native launches of one training step at config M (device time < 12 us): count and time per (aten op,
autograd node / Python source line).  GPU only."""
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
from tmg_optim import HipAdam  # noqa: E402
import tmg_ops  # noqa: E402
cfg = bench.CONFIGS["M"]
dev = torch.device("cuda")
model = bench.build_model(cfg, dev)
opt = HipAdam(model.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
B = 64
h, w = cfg["_in_hw"]
x = torch.randn(B, cfg["in_features"], h, w, device=dev)
states = model.initLSTMStates(torch.arange(B), [h * 2, w * 2])
states = [(a.contiguous(memory_format=torch.channels_last), b.contiguous(memory_format=torch.channels_last)) for a, b in states]


def step():
    opt.zero_grad(set_to_none=True)
    y, ld, _ = model.sample(x, states)
    tmg_ops.reverse_loss(y, ld).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt, tim = collections.Counter(), collections.Counter()
for e in prof.events():
    if not e.name.startswith("aten::") or e.device_time_total <= 0 or e.device_time_total >= float(os.environ.get("TMG_GLUE_MAX_US", "12")):
        continue
    if any(c.device_time_total > 0 and c.name.startswith("aten::") for c in (e.cpu_children or [])):
        continue            # count the leaf op that launched
    st = [s for s in (e.stack or []) if "deep-turbulence_amd" in s or "common.py" in s or "bench" in s]
    par, q = [], e.cpu_parent
    while q is not None and len(par) < 3:
        par.append(q.name[:44])
        q = q.cpu_parent
    key = (e.name, st[0].split("/")[-1][:70] if st else " < ".join(par))
    cnt[key] += 1
    tim[key] += e.device_time_total
print("small torch-native launches: %d, %.1f us" % (sum(cnt.values()), sum(tim.values())))
for k, n in cnt.most_common(60):
    print("%4d %7.1f us  %-18s %s" % (n, tim[k], k[0], k[1]))

# ---- second census: every device-side copy / fill (hipMemcpyAsync -> __amd_rocclr_copyBuffer, Memcpy HtoD / DtoD records) by the op
# and Python line that issued it
ccnt, ctim = collections.Counter(), collections.Counter()
for e in prof.events():
    ks = [k for k in (e.kernels or []) if ("Memcpy" in k.name or "rocclr" in k.name or "Memset" in k.name)]
    if not ks:
        continue
    if any((c.kernels or []) for c in (e.cpu_children or [])):
        continue
    st = [s for s in (e.stack or []) if "deep-turbulence_amd" in s or "common.py" in s or "bench" in s]
    par, q = [], e.cpu_parent
    while q is not None and len(par) < 4:
        par.append(q.name[:40])
        q = q.cpu_parent
    for k in ks:
        key = (k.name[:40], e.name[:30], st[0].split("/")[-1][:70] if st else " < ".join(par))
        ccnt[key] += 1
        ctim[key] += k.duration
print("device copies / fills: %d, %.1f us" % (sum(ccnt.values()), sum(ctim.values())))
for k, n in ccnt.most_common(60):
    print("%4d %7.1f us  %-40s %-30s %s" % (n, ctim[k], k[0], k[1], k[2]))
