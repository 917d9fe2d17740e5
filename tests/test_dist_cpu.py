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


N_WINDOWS = 2


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
    L = len(cfg["glow_blocks"])
    key_g = O.init_lstm_states(cfg, torch.from_numpy(d["seeds"]), [16, 16])
    key = [(tmg_dist.shard(h, rank, world), tmg_dist.shard(c, rank, world)) for h, c in key_g]
    states = [(h.clone(), c.clone()) for h, c in key]
    losses, gns = [], []
    for a in range(N_WINDOWS):   # window 0: synchronous first pass (learns the live set); window 1: hook-driven overlapped buckets
        xs_g = torch.from_numpy(d["xs"])[a]          # [tback, B, ...] global batch B = 2
        xs = [tmg_dist.shard(xs_g[t], rank, world) for t in range(xs_g.shape[0])]
        eps = [[tmg_dist.shard(torch.from_numpy(d["eps.%d.%d.%d" % (a, t, i)]), rank, world) for i in range(L + 1)] for t in range(len(xs))]

        def sample(m, x, st, t):
            return m.reconstruct(x, st, eps[t])

        loss, gn, states, outs = tmg_dist.train_window(model, opt, xs, states, key, C.loss_reverse, bucket=bucket,
                                                       max_grad_norm=float(d["max_grad_norm"]), sample=sample)
        losses.append(float(loss))
        gns.append(float(gn))
    ret[rank] = {"loss": losses, "gn": gns, "log_s": dict(zip(model.train_names, model.plist))[str(d["log_s_key"])].detach().clone(),
                 "hooked": bucket.launched_during_backward, "nbuckets": len(bucket.buckets)}
    dist.barrier()
    dist.destroy_process_group()


def _expected_two_replicas():
    """Single-process emulation of what two data-parallel replicas must compute over N_WINDOWS windows: per-shard
    forward/backward from the same weights (BatchNorm statistics stay shard-local, as in the reference's per-GPU replicas,
    parallel.py:118-150), mean of the gradients, clip, one Adam step, states re-anchored half-way to their seeds."""
    d = C.load_npz("tiny_train.npz")
    cfg = C.CFG_TINY
    L = len(cfg["glow_blocks"])
    sd = {k: torch.from_numpy(v) for k, v in C.sub(d, "sd.").items()}
    key_g = O.init_lstm_states(cfg, torch.from_numpy(d["seeds"]), [16, 16])
    # two replicas that share the trainable weights (kept identical by the averaged update) but own their BatchNorm buffers
    P0 = O.params_from_state_dict(sd)
    reps = [P0, {k: (v if v.requires_grad else v.clone()) for k, v in P0.items()}]
    names = [k for k, v in P0.items() if v.requires_grad]
    opt = torch.optim.Adam([P0[k] for k in names], lr=1e-3, weight_decay=1e-8, amsgrad=True)
    states = [[(h[r:r + 1].clone(), c[r:r + 1].clone()) for h, c in key_g] for r in range(2)]
    all_losses, all_gn = [], []
    for a in range(N_WINDOWS):
        xs_g = torch.from_numpy(d["xs"])[a]
        grads, losses = [], []
        for r in range(2):
            for k in names:
                P0[k].grad = None
            st = states[r]
            loss = 0.0
            for t in range(xs_g.shape[0]):
                eps = [torch.from_numpy(d["eps.%d.%d.%d" % (a, t, i)])[r:r + 1] for i in range(L + 1)]
                y, lp, st = O.tmglow_reconstruct(reps[r], cfg, xs_g[t, r:r + 1], st, eps)
                loss = loss + C.loss_reverse(y, lp)
            loss.backward()
            losses.append(float(loss))
            grads.append({k: P0[k].grad.clone() for k in names if P0[k].grad is not None})
            states[r] = [(0.5 * h.detach() + 0.5 * hk[r:r + 1], 0.5 * c.detach() + 0.5 * ck[r:r + 1]) for (h, c), (hk, ck) in zip(st, key_g)]
        live = list(grads[0])
        for k in names:
            P0[k].grad = 0.5 * (grads[0][k] + grads[1][k]) if k in grads[0] else None
        gn = torch.nn.utils.clip_grad_norm_([P0[k] for k in live], float(d["max_grad_norm"]))
        opt.step()
        all_losses.append(losses)
        all_gn.append(float(gn))
    return all_losses, all_gn, P0[str(d["log_s_key"])].detach()


def test_two_rank_window_matches_two_replica_emulation():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run, args=(world, port, ret), nprocs=world, join=True)
    losses, gn, log_s = _expected_two_replicas()
    assert torch.equal(ret[0]["log_s"], ret[1]["log_s"])              # replicas stay identical
    C.assert_field(ret[0]["log_s"], log_s, "log_s after the steps", atol=2e-6)
    for r in range(2):
        for a in range(N_WINDOWS):
            assert abs(ret[r]["loss"][a] - losses[a][r]) < 2e-5 * (1 + a)       # rank-local loss on its own shard
            assert abs(ret[r]["gn"][a] - gn[a]) < 2e-4 * gn[a] * (1 + a)        # clip norm of the AVERAGED gradient on every rank
        # the second window's buckets were all-reduced from the gradient hooks, while backward was still running
        assert ret[r]["nbuckets"] > 1 and ret[r]["hooked"] >= ret[r]["nbuckets"] - 1, (ret[r]["hooked"], ret[r]["nbuckets"])


def _run_bucket_edges(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, C.PKG)
    import tmg_dist
    torch.set_num_threads(1)
    tmg_dist.init_from_env("gloo")
    torch.manual_seed(5)
    a, b, c = (torch.nn.Parameter(torch.randn(7, 3)) for _ in range(3))
    x = torch.randn(4, 3) + rank
    bucket = tmg_dist.GradBucket([a, b, c], bucket_mb=1e-4, order="reverse")      # one parameter per bucket
    out = {}

    def step(use):
        for p in (a, b, c):
            p.grad = None
        sum(((p @ x.t()) ** 2).sum() * (i + 1) for i, p in enumerate((a, b, c)) if i in use).backward()
        own = {i: p.grad.clone() for i, p in enumerate((a, b, c)) if p.grad is not None}
        bucket.allreduce_mean()
        return own

    flat_ptrs = None
    for name, use in (("first", (0, 1)), ("hooked", (0, 1)), ("late", (0, 1, 2)), ("missing", (0, 2)), ("steady", (0, 1, 2))):
        own = step(use)
        out[name] = {"own": own, "mean": {i: p.grad.clone() for i, p in enumerate((a, b, c)) if p.grad is not None}}
        if name == "late":
            flat_ptrs = [f.data_ptr() for f in bucket._flat]
    out["persistent"] = flat_ptrs == [f.data_ptr() for f in bucket._flat]      # the flat buffers are allocated once
    out["flag_reads"], out["nbuckets"], out["n_hooked"] = bucket.flag_reads, len(bucket.buckets), bucket.launched_during_backward
    # a second backward before the exchange must raise instead of silently reducing partial gradients
    for p in (a, b, c):
        p.grad = None
    loss = sum(((p @ x.t()) ** 2).sum() for p in (a, b, c))
    loss.backward(retain_graph=True)
    try:
        loss.backward()
        out["second_backward"] = "accepted"
    except RuntimeError as e:
        out["second_backward"] = "raised" if "GradBucket" in str(e) else str(e)
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_live_set_changes_and_misuse():
    """GradBucket hygiene: the bucket layout is static (all parameters that require a gradient), so a parameter that first receives
    a gradient in a later step is reduced without any re-layout, a parameter without a gradient on ANY rank keeps `grad = None` as
    in a single-process run (the optimizer skips it), the flat buffers persist, the reduced has-gradient flags are read on the host
    only when the live set changes, and a second backward before the exchange raises."""
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run_bucket_edges, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    for name in ("first", "hooked", "late", "missing", "steady"):
        keys = set(r0[name]["own"]) | set(r1[name]["own"])
        assert set(r0[name]["mean"]) == keys and set(r1[name]["mean"]) == keys, (name, keys)    # no gradient anywhere -> None
        for i in sorted(keys):
            want = 0.5 * (r0[name]["own"].get(i, 0.0) + r1[name]["own"].get(i, 0.0))
            assert torch.allclose(r0[name]["mean"][i], want, rtol=1e-6, atol=1e-6), (name, i)
            assert torch.equal(r0[name]["mean"][i], r1[name]["mean"][i]), (name, i)
    # live sets: {0,1} {0,1} {0,1,2} {0,2} {0,1,2}: the flags are read in steps 1, 3, 4, 5 - not in the steady second step
    assert r0["flag_reads"] == 4 and r0["nbuckets"] == 3 and r0["persistent"] and r0["n_hooked"] > 0
    assert r0["second_backward"] == "raised" and r1["second_backward"] == "raised"


def _run_divergent_live_sets(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, C.PKG)
    import tmg_dist
    torch.set_num_threads(1)
    tmg_dist.init_from_env("gloo")
    torch.manual_seed(5)
    a, b, c = (torch.nn.Parameter(torch.randn(7, 3)) for _ in range(3))
    x = torch.randn(4, 3) + rank
    bucket = tmg_dist.GradBucket([a, b, c], bucket_mb=1e-4, order="reverse")
    out = []
    # ranks DISAGREE about which parameters have gradients (rank 0: a, b; rank 1: a, c), then swap: the fixed collective sequence
    # must neither hang nor mis-pair buffers; a parameter live on one rank is reduced with zeros from the other
    for use in ((0, 1) if rank == 0 else (0, 2), (0, 2) if rank == 0 else (0, 1), (0, 1, 2)):
        for p in (a, b, c):
            p.grad = None
        sum(((p @ x.t()) ** 2).sum() * (i + 1) for i, p in enumerate((a, b, c)) if i in use).backward()
        own = {i: p.grad.clone() for i, p in enumerate((a, b, c)) if p.grad is not None}
        bucket.allreduce_mean()
        out.append({"own": own, "mean": {i: p.grad.clone() for i, p in enumerate((a, b, c)) if p.grad is not None}})
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_with_different_live_sets_do_not_deadlock():
    world = 2
    port = 33500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run_divergent_live_sets, args=(world, port, ret), nprocs=world, join=True)
    for s0, s1 in zip(ret[0], ret[1]):
        assert set(s0["mean"]) == {0, 1, 2} and set(s1["mean"]) == {0, 1, 2}
        for i in range(3):
            want = 0.5 * (s0["own"].get(i, 0.0) + s1["own"].get(i, 0.0))
            assert torch.allclose(s0["mean"][i], want, rtol=1e-6, atol=1e-6) and torch.equal(s0["mean"][i], s1["mean"][i])


def _run_one_sided_growth(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, C.PKG)
    import tmg_dist
    import tmg_ops
    torch.set_num_threads(1)
    tmg_dist.init_from_env("gloo")

    class Lin(torch.autograd.Function):      # routes its parameter gradient through the gradient sink (bound AFTER backward)
        @staticmethod
        def forward(ctx, p, x):
            ctx.save_for_backward(p, x)
            return p @ x.t()

        @staticmethod
        def backward(ctx, g):
            p, x = ctx.saved_tensors
            return tmg_ops._defer((p,), (g @ x,)) + (None,)

    torch.manual_seed(5)
    prm = [torch.nn.Parameter(torch.randn(7, 3)) for _ in range(4)]      # a, b, c, d -> buckets [d, c], [b, a]
    x = torch.randn(4, 3) + rank
    bucket = tmg_dist.GradBucket(prm, bucket_mb=2.1 * 21 * 4 / 2 ** 20, order="reverse")
    opt = torch.optim.Adam(prm, lr=1e-2, weight_decay=1e-3, amsgrad=True)
    out = []
    # both ranks: b and d live.  From step 2 on rank 1 ALONE also uses c - its gradient is bound after backward (deferred node), i.e.
    # after bucket [d, c] has gone from d's hook; from step 4 on rank 1 alone uses a through plain autograd.  Rank 0's own live set
    # never changes: it must still send the late bucket a second time, re-read the flags and step c / a like rank 1 does.
    for step in range(6):
        use = {1, 3} | ({2} if rank == 1 and step >= 2 else set()) | ({0} if rank == 1 and step >= 4 else set())
        opt.zero_grad(set_to_none=True)
        loss = 0.0
        for i in sorted(use, reverse=True):
            y = Lin.apply(prm[i], x) if i == 2 else prm[i] @ x.t()
            loss = loss + (y ** 2).sum() * (i + 1)
        with tmg_ops.fused_grad_accumulation():
            loss.backward()
        own = {i: p.grad.clone() for i, p in enumerate(prm) if p.grad is not None}
        bucket.allreduce_mean()
        out.append({"own": own, "mean": {i: p.grad.clone() for i, p in enumerate(prm) if p.grad is not None}})
        opt.step()
    ret[rank] = {"steps": out, "params": [p.detach().clone() for p in prm], "second": bucket.second_passes,
                 "reads": bucket.flag_reads, "nbuckets": len(bucket.buckets), "hooked": bucket.launched_during_backward}
    dist.barrier()
    dist.destroy_process_group()


def test_live_set_grows_on_one_rank_only():
    """Round-4 advisor finding: the re-read of the reduced has-gradient flags and the second pass of a bucket whose live set grew
    behind it were decided from RANK-LOCAL state.  Here only rank 1's live set grows (once behind a bucket that has already gone
    from a hook, once not); rank 0 - whose own set never changes - must issue the same collectives, bind the same gradient set and
    keep its replica identical."""
    world = 2
    port = 34500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run_one_sided_growth, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0["nbuckets"] == 2 and r0["hooked"] > 0 and r1["hooked"] > 0
    for k, (s0, s1) in enumerate(zip(r0["steps"], r1["steps"])):
        keys = set(s0["own"]) | set(s1["own"])
        assert set(s0["mean"]) == keys and set(s1["mean"]) == keys, (k, keys, set(s0["mean"]), set(s1["mean"]))
        for i in sorted(keys):
            want = 0.5 * (s0["own"].get(i, 0.0) + s1["own"].get(i, 0.0))
            assert torch.allclose(s0["mean"][i], want, rtol=1e-6, atol=1e-6), (k, i)
            assert torch.equal(s0["mean"][i], s1["mean"][i]), (k, i)
    assert set(r0["steps"][1]["mean"]) == {1, 3} and set(r0["steps"][2]["mean"]) == {1, 2, 3} and set(r0["steps"][5]["mean"]) == {0, 1, 2, 3}
    for p0, p1 in zip(r0["params"], r1["params"]):
        assert torch.equal(p0, p1)                       # replicas stay identical through the optimizer steps
    # the deferred gradient of c is late in EVERY step from 2 on (it never passes a hook): the bucket goes twice on BOTH ranks
    assert r0["second"] == r1["second"] and r0["second"] >= 1
    assert r0["reads"] == r1["reads"]


def _run_change_after_settling(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, C.PKG)
    import warnings
    import tmg_dist
    torch.set_num_threads(1)
    tmg_dist.init_from_env("gloo")
    torch.manual_seed(5)
    prm = [torch.nn.Parameter(torch.randn(7, 3)) for _ in range(4)]      # a, b, c, d -> buckets [d, c], [b, a]
    x = torch.randn(4, 3) + rank
    bucket = tmg_dist.GradBucket(prm, bucket_mb=2.1 * 21 * 4 / 2 ** 20, defer_on_cpu=True, order="reverse")
    opt = torch.optim.SGD(prm, lr=1e-2)
    out, warned = [], []
    for step in range(10):
        use = {1, 3} | ({0} if rank == 1 and step >= 6 else set())          # from step 6 on rank 1 ALONE also uses a
        opt.zero_grad(set_to_none=True)
        loss = 0.0
        for i in sorted(use, reverse=True):
            loss = loss + ((prm[i] @ x.t()) ** 2).sum() * (i + 1)
        loss.backward()
        own = {i: p.grad.clone() for i, p in enumerate(prm) if p.grad is not None}
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            bucket.allreduce_mean()
        warned.append(any("changed after it had settled" in str(m.message) for m in w))
        out.append({"own": own, "mean": {i: p.grad.clone() for i, p in enumerate(prm) if p.grad is not None}, "deferred": bucket.deferred_steps})
        opt.step()
    ret[rank] = {"steps": out, "warned": warned, "params": [p.detach().clone() for p in prm], "second": bucket.second_passes}
    dist.barrier()
    dist.destroy_process_group()


def test_live_set_change_after_settling_is_caught_one_step_late_on_every_rank():
    """Round 6: once the live set has been quiet for SYNC_STEPS steps the reduced control vector is examined ONE STEP LATE (no host
    synchronisation in a steady step).  Here rank 1 alone starts using parameter `a` at step 6, long after settling: step 6 runs without
    the flag re-read (a's gradient is dropped - on BOTH ranks alike: the replicas stay identical), step 7 finds the previous step's
    reduced vector non-zero on both ranks (a warning each), returns to the synchronous protocol and binds a's mean gradient from then
    on.  No rank-local decision, no deadlock, no divergence."""
    world = 2
    port = 36500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run_change_after_settling, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0["steps"][5]["deferred"] >= 2 and r1["steps"][5]["deferred"] >= 2          # steps 4 and 5 ran without the host read
    for k in range(10):
        s0, s1 = r0["steps"][k], r1["steps"][k]
        want_keys = {1, 3} if k <= 6 else {0, 1, 3}
        assert set(s0["mean"]) == want_keys and set(s1["mean"]) == want_keys, (k, set(s0["mean"]), set(s1["mean"]))
        for i in want_keys:
            want = 0.5 * (s0["own"].get(i, 0.0) + s1["own"].get(i, 0.0))
            assert torch.allclose(s0["mean"][i], want, rtol=1e-6, atol=1e-6), (k, i)
            assert torch.equal(s0["mean"][i], s1["mean"][i]), (k, i)
    assert r0["warned"] == r1["warned"] and r0["warned"][7] and sum(r0["warned"]) == 1   # caught at step 7, by both ranks, once
    for p0, p1 in zip(r0["params"], r1["params"]):
        assert torch.equal(p0, p1)
    assert r0["second"] == r1["second"]


def _run_flat_broadcast(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, C.PKG)
    import tmg_dist
    torch.set_num_threads(1)
    tmg_dist.init_from_env("gloo")
    torch.manual_seed(100 + rank)            # every rank starts with DIFFERENT weights and buffers
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.BatchNorm2d(5), torch.nn.Conv2d(5, 2, 1))
    with torch.no_grad():
        m[1].running_mean.add_(rank + 1.0)
        m[1].num_batches_tracked.add_(3 + rank)
    calls = []
    real = dist.broadcast
    dist.broadcast = lambda t, src, *a, **k: (calls.append(t.numel()), real(t, src, *a, **k))[1]
    try:
        tmg_dist.broadcast_parameters(m)
    finally:
        dist.broadcast = real
    ret[rank] = ({k: v.clone() for k, v in m.state_dict().items()}, calls)
    dist.barrier()
    dist.destroy_process_group()


def test_parameter_broadcast_is_one_collective_per_dtype():
    world = 2
    port = 35500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run_flat_broadcast, args=(world, port, ret), nprocs=world, join=True)
    torch.manual_seed(100)
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.BatchNorm2d(5), torch.nn.Conv2d(5, 2, 1))
    (sd0, calls0), (sd1, _) = ret[0], ret[1]
    assert len(calls0) == 2, calls0                          # fp32 tensors, int64 counter
    for k, v in ref.state_dict().items():
        want = v + 1.0 if k == "1.running_mean" else (v + 3 if k == "1.num_batches_tracked" else v)
        assert torch.equal(sd0[k], want) and torch.equal(sd1[k], want), k


def _run_deferred_grads(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, C.PKG)
    import tmg_dist
    import tmg_ops
    torch.set_num_threads(1)
    tmg_dist.init_from_env("gloo")

    class Lin(torch.autograd.Function):      # a node that routes its parameter gradient through the gradient sink, like the HIP nodes
        @staticmethod
        def forward(ctx, p, x):
            ctx.save_for_backward(p, x)
            return p @ x.t()

        @staticmethod
        def backward(ctx, g):
            p, x = ctx.saved_tensors
            return tmg_ops._defer((p,), (g @ x,)) + (None,)

    torch.manual_seed(3)
    a, b = (torch.nn.Parameter(torch.randn(5, 3)) for _ in range(2))
    x = torch.randn(4, 3) + rank
    bucket = tmg_dist.GradBucket([a, b], bucket_mb=1e-4, order="reverse")
    out = []
    for step in range(3):
        for p in (a, b):
            p.grad = None
        loss = sum((Lin.apply(p, x * (t + 1)) ** 2).sum() * (i + 1) for t in range(2) for i, p in enumerate((a, b)))     # two "time-steps"
        with tmg_ops.fused_grad_accumulation():
            loss.backward()
        own = [p.grad.clone() for p in (a, b)]
        bucket.allreduce_mean()
        out.append({"own": own, "mean": [p.grad.clone() for p in (a, b)]})
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_with_gradients_bound_after_backward():
    """Inside tmg_ops.fused_grad_accumulation a node hands autograd no parameter gradient (the window's sum is bound to p.grad when
    backward has finished) - but the parameter's post-accumulate hook still fires.  The bucket must not take that for an arrival:
    round 4's first version reduced zero-filled buckets from the second window on (found by tests/test_dist_gpu.py)."""
    world = 2
    port = 37500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run_deferred_grads, args=(world, port, ret), nprocs=world, join=True)
    for s0, s1 in zip(ret[0], ret[1]):
        for i in range(2):
            want = 0.5 * (s0["own"][i] + s1["own"][i])
            assert float(want.abs().max()) > 0
            assert torch.allclose(s0["mean"][i], want, rtol=1e-6, atol=1e-6) and torch.equal(s0["mean"][i], s1["mean"][i])
