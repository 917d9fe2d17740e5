"""Worker of tests/test_dist_gpu.py: one rank of a 2-rank data-parallel run of the HIP model (both ranks on cuda:0,
TMG_SINGLE_DEVICE=1, gloo backend - RCCL cannot put two ranks on one device).  Launched by torch.distributed.run."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import common as C  # noqa: E402

sys.path.insert(0, C.PKG)
import tmg_dist  # noqa: E402
from nn.tmGlow import TMGlow  # noqa: E402


def main(out_path, n_windows, captured=False):
    rank, world, _ = tmg_dist.init_from_env("gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    d = C.load_npz("tiny_train.npz")
    cfg = C.CFG_TINY
    L = len(cfg["glow_blocks"])
    m = TMGlow(**C.build_kwargs(cfg))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in C.sub(d, "sd.").items()})
    m.to(dev).train()
    if rank == 1:
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.05)          # de-synchronise; the broadcast below must undo it
    tmg_dist.broadcast_parameters(m)
    bucket = tmg_dist.GradBucket(m.parameters(), bucket_mb=0.02)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
    seeds = torch.from_numpy(d["seeds"])
    key = m.initLSTMStates(tmg_dist.shard(seeds, rank, world), [16, 16])
    states = [(h.clone(), c.clone()) for h, c in key]
    res = {"loss": [], "gn": []}
    cw = None
    for a in range(n_windows):
        xs_g = torch.from_numpy(d["xs"])[a]
        xs = [tmg_dist.shard(xs_g[t], rank, world).to(dev) for t in range(xs_g.shape[0])]
        eps = [[tmg_dist.shard(torch.from_numpy(d["eps.%d.%d.%d" % (a, t, i)]), rank, world).to(dev) for i in range(L + 1)]
               for t in range(len(xs))]
        if captured and cw is None:
            # forward passes + loss + backward of a window as one hipGraph replay (the latents are arguments of the recorded body);
            # the bucket's hooks are switched off, the exchange follows the replay
            def body(xs_, states_, eps_):
                loss_, outs_ = 0.0, []
                for t in range(len(xs_)):
                    y, lp, states_ = m.reconstruct(xs_[t], states_, eps_[t])
                    loss_ = loss_ + C.loss_reverse(y, lp)
                    outs_.append((y, lp))
                return loss_, (states_, outs_)
            cw = tmg_dist.CapturedWindow(m, body, (xs, states, eps), bucket=bucket)
        loss, gn, states, _ = tmg_dist.train_window(m, opt, xs, states, key, C.loss_reverse, bucket=bucket,
                                                    max_grad_norm=float(d["max_grad_norm"]),
                                                    sample=lambda mod, x, st, t: mod.reconstruct(x, st, eps[t]),
                                                    captured=cw, captured_extra=(eps,))
        res["loss"].append(float(loss))
        res["gn"].append(float(gn))
    res["log_s"] = dict(m.named_parameters())[str(d["log_s_key"])].detach().cpu()
    res["hooked"], res["nbuckets"] = bucket.launched_during_backward, len(bucket.buckets)
    if not captured:
        # the SINGLE-STEP path (bench.py's: plain loss.backward(), no fused accumulation): the gradients reach p.grad through autograd's
        # AccumulateGrad, the hooks see them arrive and hand complete buckets to the collective while backward is still running
        h0 = bucket.launched_during_backward
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            y, lp, _ = m.reconstruct(xs[0], states, eps[0])
            C.loss_reverse(y, lp).backward()
            bucket.allreduce_mean()
            opt.step()
        res["hooked_single_step"] = bucket.launched_during_backward - h0
        res["log_s_single"] = dict(m.named_parameters())[str(d["log_s_key"])].detach().cpu()
    torch.save(res, "%s.rank%d" % (out_path, rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), len(sys.argv) > 3 and sys.argv[3] == "captured")
