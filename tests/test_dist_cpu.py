"""World-size-2 gloo test of the data-parallel harness (tmg_dist): batch sharding with rank-local LSTM states,
mean gradient all-reduce, and equivalence with the single-process result on the global batch.
The compute stand-in on CPU is the oracle (the HIP product path has no CPU mode); the harness code under
test (GradBucket, shard, broadcast_parameters, train_window) is exactly what bench.py runs on N GPUs."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common as C
from oracle import tmglow_oracle as O


class OracleModel(torch.nn.Module):
    """nn.Module shell around the functional oracle so the harness sees .parameters() / .sample()."""

    def __init__(self, sd, cfg):
        super().__init__()
        self.cfg = cfg
        P = O.params_from_state_dict(sd)
        self.names = list(P.keys())
        self.P = P
        self.plist = torch.nn.ParameterList([torch.nn.Parameter(v.detach().clone()) for v in P.values() if v.requires_grad])
        self.train_names = [k for k, v in P.items() if v.requires_grad]

    def _params(self):
        P = dict(self.P)
        for k, p in zip(self.train_names, self.plist):
            P[k] = p
        return P

    def reconstruct(self, x, h, eps):
        return O.tmglow_reconstruct(self._params(), self.cfg, x, h, eps)


def _run(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, C.PKG)
    import tmg_dist
    torch.set_num_threads(2)
    r, w, _ = tmg_dist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    d = C.load_npz("tiny_train.npz")
    cfg = C.CFG_TINY
    sd = {k: torch.from_numpy(v) for k, v in C.sub(d, "sd.").items()}
    model = OracleModel(sd, cfg)
    if rank == 1:  # deliberately de-synchronise, then let the harness broadcast rank 0's weights
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.1)
    tmg_dist.broadcast_parameters(model)
    bucket = tmg_dist.GradBucket(model.parameters(), bucket_mb=0.05)  # several buckets
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
    xs_g = torch.from_numpy(d["xs"])[0]          # [tback, B, ...] global batch B = 2
    L = len(cfg["glow_blocks"])
    key_g = O.init_lstm_states(cfg, torch.from_numpy(d["seeds"]), [16, 16])
    xs = [tmg_dist.shard(xs_g[t], rank, world) for t in range(xs_g.shape[0])]
    key = [(tmg_dist.shard(h, rank, world), tmg_dist.shard(c, rank, world)) for h, c in key_g]
    eps = [[tmg_dist.shard(torch.from_numpy(d["eps.0.%d.%d" % (t, i)]), rank, world) for i in range(L + 1)] for t in range(len(xs))]

    def sample(m, x, st, t):
        return m.reconstruct(x, st, eps[t])

    loss, gn, states, outs = tmg_dist.train_window(model, opt, xs, [(h.clone(), c.clone()) for h, c in key], key, C.loss_reverse,
                                                   bucket=bucket, max_grad_norm=float(d["max_grad_norm"]), sample=sample)
    ret[rank] = {"loss": float(loss), "gn": float(gn), "log_s": dict(zip(model.train_names, model.plist))[str(d["log_s_key"])].detach().clone(),
                 "y0": outs[0][0].clone()}
    dist.barrier()
    dist.destroy_process_group()


def _expected_two_replicas():
    """Single-process emulation of what two data-parallel replicas must compute: per-shard forward/backward from the
    same weights (BatchNorm statistics stay shard-local, as in the reference's per-GPU replicas, parallel.py:118-150),
    mean of the gradients, clip, one Adam step."""
    d = C.load_npz("tiny_train.npz")
    cfg = C.CFG_TINY
    L = len(cfg["glow_blocks"])
    sd = {k: torch.from_numpy(v) for k, v in C.sub(d, "sd.").items()}
    xs_g = torch.from_numpy(d["xs"])[0]
    key_g = O.init_lstm_states(cfg, torch.from_numpy(d["seeds"]), [16, 16])
    losses, grads = [], []
    for r in range(2):
        P = O.params_from_state_dict(sd)
        st = [(h[r:r + 1].clone(), c[r:r + 1].clone()) for h, c in key_g]
        loss = 0.0
        for t in range(xs_g.shape[0]):
            eps = [torch.from_numpy(d["eps.0.%d.%d" % (t, i)])[r:r + 1] for i in range(L + 1)]
            y, lp, st = O.tmglow_reconstruct(P, cfg, xs_g[t, r:r + 1], st, eps)
            loss = loss + C.loss_reverse(y, lp)
        loss.backward()
        losses.append(float(loss))
        grads.append({k: v.grad.clone() for k, v in P.items() if v.requires_grad and v.grad is not None})
    P = O.params_from_state_dict(sd)
    params = [P[k] for k in grads[0]]
    for k, p_ in zip(grads[0], params):
        p_.grad = 0.5 * (grads[0][k] + grads[1][k])
    gn = torch.nn.utils.clip_grad_norm_(params, float(d["max_grad_norm"]))
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-8, amsgrad=True)
    opt.step()
    return losses, float(gn), P[str(d["log_s_key"])].detach()


def test_two_rank_window_matches_two_replica_emulation():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run, args=(world, port, ret), nprocs=world, join=True)
    losses, gn, log_s = _expected_two_replicas()
    assert torch.equal(ret[0]["log_s"], ret[1]["log_s"])              # replicas stay identical
    C.assert_field(ret[0]["log_s"], log_s, "log_s after step", atol=1e-6)
    for r in range(2):
        assert abs(ret[r]["loss"] - losses[r]) < 1e-5                    # rank-local loss on its own shard
        assert abs(ret[r]["gn"] - gn) < 1e-4 * gn                        # clip norm of the AVERAGED gradient on every rank
