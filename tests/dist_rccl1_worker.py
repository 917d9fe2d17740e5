"""Worker of tests/test_dist_gpu.py::test_rccl_world_size_one_*: ONE rank on the real collective backend ("nccl" = RCCL), a process
group of world size 1 on cuda:0.  Everything a multi-GPU rank runs goes through RCCL here - the flat parameter broadcast, the
persistent gradient buckets with their asynchronous handles, the hook-driven launches during backward, the control vector in the
last bucket's tail, the stream ordering between the collective stream and the compute stream - only the peers are missing.  The same
steps are run by an identical model without any bucket; results must agree."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import common as C  # noqa: E402

sys.path.insert(0, C.PKG)
import tmg_dist  # noqa: E402
from nn.tmGlow import TMGlow  # noqa: E402


def main(out_path):
    os.environ["TMG_FORCE_DIST"] = "1"
    rank, world, _ = tmg_dist.init_from_env("nccl")
    assert world == 1 and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    d = C.load_npz("tiny_train.npz")
    cfg = C.CFG_TINY
    L = len(cfg["glow_blocks"])
    sd = {k: torch.from_numpy(v) for k, v in C.sub(d, "sd.").items()}
    res = {}
    models = {}
    for tag in ("rccl", "plain"):
        m = TMGlow(**C.build_kwargs(cfg))
        m.load_state_dict(sd)
        models[tag] = m.to(dev).train()
    m = models["rccl"]
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    tmg_dist.broadcast_parameters(m, force=True)             # flat broadcast through RCCL: a one-rank group returns the values
    torch.cuda.synchronize()
    res["broadcast_exact"] = all(torch.equal(v, before[k]) for k, v in m.state_dict().items())
    bucket = tmg_dist.GradBucket(m.parameters(), bucket_mb=0.02, measure=True, force=True)
    seeds = torch.from_numpy(d["seeds"])
    xs_g = torch.from_numpy(d["xs"])[0]
    xs = [xs_g[t].to(dev) for t in range(xs_g.shape[0])]
    eps = [[torch.from_numpy(d["eps.0.%d.%d" % (t, i)]).to(dev) for i in range(L + 1)] for t in range(len(xs))]
    out = {}
    for tag in ("rccl", "plain"):
        m = models[tag]
        bk = bucket if tag == "rccl" else None
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
        key = m.initLSTMStates(seeds, [16, 16])
        states = [(h.clone(), c.clone()) for h, c in key]
        rec = {"loss": [], "gn": []}
        # (a) the trainer's unit: a BPTT window (fused gradient accumulation: the buckets go after backward)
        for _ in range(2):
            loss, gn, states, _ = tmg_dist.train_window(m, opt, xs, states, key, C.loss_reverse, bucket=bk, max_grad_norm=float(d["max_grad_norm"]),
                                                        sample=lambda mod, x, st, t: mod.reconstruct(x, st, eps[t]))
            rec["loss"].append(float(loss))
            rec["gn"].append(float(gn))
        # (b) the benchmark's unit: single steps with plain backward - the hooks hand complete buckets to RCCL while backward runs
        h0 = bucket.launched_during_backward if bk is not None else 0
        for _ in range(6):
            opt.zero_grad(set_to_none=True)
            y, lp, _ = m.reconstruct(xs[0], states, eps[0])
            loss = C.loss_reverse(y, lp)
            loss.backward()
            if bk is not None:
                bk.allreduce_mean()
            opt.step()
            rec["loss"].append(float(loss))
        if bk is not None:
            rec["hooked_single_step"] = bucket.launched_during_backward - h0
            rec["nbuckets"] = len(bucket.buckets)
            rec["overlap"] = bucket.overlap_report()
            rec["second_passes"] = bucket.second_passes
            rec["deferred_steps"] = bucket.deferred_steps
        rec["params"] = {k: v.detach().cpu().clone() for k, v in m.named_parameters()}
        out[tag] = rec
    res.update(out)
    res["backend"] = torch.distributed.get_backend()
    torch.save(res, out_path)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
