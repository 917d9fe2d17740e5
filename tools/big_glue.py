import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import torch, bench
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import common as C
cfg = bench.CONFIGS["M"]; dev = torch.device("cuda")
model = bench.build_model(cfg, dev)
B = 64; h, w = cfg["_in_hw"]
x = torch.randn(B, cfg["in_features"], h, w, device=dev)
states = model.initLSTMStates(torch.arange(B), [h * 2, w * 2])
states = [(a.contiguous(memory_format=torch.channels_last), b.contiguous(memory_format=torch.channels_last)) for a, b in states]
def step():
    model.zero_grad(set_to_none=True)
    y, ld, _ = model.sample(x, states)
    C.loss_reverse(y, ld).backward()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.name in ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::fill_", "aten::mul", "aten::sum") and e.device_time_total > 15:
        st = [s for s in (e.stack or []) if "deep-turbulence_amd" in s or "common.py" in s]
        par = []; q = e.cpu_parent
        while q is not None and len(par) < 4: par.append(q.name[:40]); q = q.cpu_parent
        rows.append((e.device_time_total, e.name, str(e.input_shapes)[:80], (st[0].split("/")[-1][:60] if st else ""), " < ".join(par)))
rows.sort(reverse=True)
print("total us", sum(r[0] for r in rows))
for r in rows[:45]: print("%8.1f us %-12s %-80s %-60s %s" % r)
