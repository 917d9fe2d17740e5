#!/usr/bin/env python3
"""What the gradient exchange costs a rank, measured on ONE GPU (a process group of one rank on RCCL): config M, 64 samples, the
bench step with (a) no bucket, (b) GradBucket on the one-rank group, (c) GradBucket with the collective itself patched out (the
copies into the flat buffers, the control vector, the division and the re-binding remain).  GPU only."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
sys.argv = [sys.argv[0]]
os.environ["TMG_FORCE_DIST"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
import tmg_dist  # noqa: E402
import tmg_ops  # noqa: E402
from tmg_optim import HipAdam  # noqa: E402

tmg_dist.init_from_env("nccl")
dev = torch.device("cuda")
cfg = bench.CONFIGS["M"]
model = bench.build_model(cfg, dev)
opt = HipAdam(model.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
B = 64
h, w = cfg["_in_hw"]
x = torch.randn(B, cfg["in_features"], h, w, device=dev)
states = model.initLSTMStates(torch.arange(B), [h * 2, w * 2])
states = [(a.contiguous(memory_format=torch.channels_last), b.contiguous(memory_format=torch.channels_last)) for a, b in states]


def run(bucket, n=15):
    def step():
        opt.zero_grad(set_to_none=True)
        y, ld, _ = model.sample(x, states)
        tmg_ops.reverse_loss(y, ld).backward()
        if bucket is not None:
            bucket.allreduce_mean()
        opt.step()
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    if bucket is not None:
        bucket.overlap_report()      # (drop the warm-up steps' event pairs: the first collective creates the communicator)
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("no bucket                      %.2f ms per step" % run(None))
bk = tmg_dist.GradBucket(model.parameters(), measure=True, force=True, bucket_mb=float(os.environ.get("TMG_BUCKET_MB", "32")),
                         order=os.environ.get("TMG_BUCKET_ORDER", "arrival"))
t_ = run(bk)
print("bucket, RCCL one-rank group    %.2f ms per step   %d buckets, %d launched from hooks, report of the timed steps: %s" % (t_, len(bk.buckets), bk.launched_during_backward, bk.overlap_report()))
real = dist.all_reduce


class _Done:
    def wait(self):
        return True


dist.all_reduce = lambda *a, **k: _Done()
print("bucket, collective patched out %.2f ms per step" % run(bk))
dist.all_reduce = real
if os.environ.get("TMG_BUCKET_PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    run(bk, n=10)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
dist.barrier()
dist.destroy_process_group()
