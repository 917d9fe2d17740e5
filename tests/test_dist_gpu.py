"""Two data-parallel ranks of the HIP model on ONE device (TMG_SINGLE_DEVICE=1, gloo) against the same two shards processed one
after the other in a single process with averaged gradients: exercises batch sharding, the parameter broadcast, rank-local LSTM
states and the hook-driven bucketed gradient all-reduce of tmg_dist.GradBucket on device tensors (the RCCL path differs only
in the backend string)."""
import os
import subprocess
import sys

import pytest
import torch

import common as C

pytestmark = pytest.mark.gpu
N_WINDOWS = 2


def _single_process_two_shards():
    from nn.tmGlow import TMGlow
    dev = "cuda"
    d = C.load_npz("tiny_train.npz")
    cfg = C.CFG_TINY
    L = len(cfg["glow_blocks"])
    reps = []
    for r in range(2):   # two replicas: shared trainable values (kept equal by the averaged update), own BatchNorm buffers
        m = TMGlow(**C.build_kwargs(cfg))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in C.sub(d, "sd.").items()})
        reps.append(m.to(dev).train())
    params = [list(m.parameters()) for m in reps]
    opts = [torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-8, amsgrad=True) for ps in params]
    seeds = torch.from_numpy(d["seeds"])
    keys = [reps[r].initLSTMStates(seeds[r:r + 1], [16, 16]) for r in range(2)]
    states = [[(h.clone(), c.clone()) for h, c in keys[r]] for r in range(2)]
    losses, gns = [], []
    for a in range(N_WINDOWS):
        xs_g = torch.from_numpy(d["xs"])[a]
        ls = []
        for r in range(2):
            opts[r].zero_grad(set_to_none=True)
            st, loss = states[r], 0.0
            for t in range(xs_g.shape[0]):
                eps = [torch.from_numpy(d["eps.%d.%d.%d" % (a, t, i)])[r:r + 1].to(dev) for i in range(L + 1)]
                y, lp, st = reps[r].reconstruct(xs_g[t, r:r + 1].to(dev), st, eps)
                loss = loss + C.loss_reverse(y, lp)
            loss.backward()
            ls.append(float(loss))
            states[r] = [(0.5 * h.detach() + 0.5 * hk, 0.5 * c.detach() + 0.5 * ck) for (h, c), (hk, ck) in zip(st, keys[r])]
        for p0, p1 in zip(*params):   # mean of the two shards' gradients on both replicas
            if p0.grad is not None:
                g = 0.5 * (p0.grad + p1.grad)
                p0.grad, p1.grad = g, g.clone()
        gn = torch.nn.utils.clip_grad_norm_([p for p in params[0] if p.grad is not None], float(d["max_grad_norm"]))
        torch.nn.utils.clip_grad_norm_([p for p in params[1] if p.grad is not None], float(d["max_grad_norm"]))
        for o in opts:
            o.step()
        losses.append(ls)
        gns.append(float(gn))
    return losses, gns, dict(reps[0].named_parameters())[str(d["log_s_key"])].detach().cpu()


@pytest.mark.parametrize("mode", ["eager", "captured"])
def test_two_ranks_on_one_device_match_single_process(tmp_path, mode):
    """captured: every rank replays its windows as hipGraphs (tmg_dist.CapturedWindow with the bucket's hooks switched off: the exchange
    follows the replay) - same losses, gradient norms and parameters as the eager ranks and as the single process."""
    out = str(tmp_path / "res")
    env = dict(os.environ, TMG_SINGLE_DEVICE="1", TMG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + (os.getpid() % 1500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(C.ROOT, "tests", "dist_hip_worker.py"), out, str(N_WINDOWS)] + (
        ["captured"] if mode == "captured" else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [torch.load("%s.rank%d" % (out, k)) for k in range(2)]
    losses, gns, log_s = _single_process_two_shards()
    assert torch.equal(res[0]["log_s"], res[1]["log_s"])                       # replicas stay identical
    C.assert_field(res[0]["log_s"], log_s, "log_s after the steps", atol=2e-6)
    for k in range(2):
        for a in range(N_WINDOWS):
            assert abs(res[k]["loss"][a] - losses[a][k]) < 2e-5 * (1 + a)
            assert abs(res[k]["gn"][a] - gns[a]) < 5e-4 * gns[a] * (1 + a)
        # (train_window sums the window's per-time-step parameter gradients in tmg_ops.fused_grad_accumulation: they reach p.grad when
        # backward has finished, so the buckets of the custom nodes' parameters go right after it instead of from the hooks - 0.2 ms
        # of exposed exchange per 0.5 s window against 1.5 ms of tiny adds per time-step; the hook-driven launches are exercised by
        # tests/test_dist_cpu.py and asserted on device tensors below: the worker's two single steps)
        assert res[k]["nbuckets"] > 1
        if mode == "eager":
            # two single steps after the windows (plain backward on device tensors): buckets were handed to the collective from the hooks
            assert res[k]["hooked_single_step"] >= 2, res[k]["hooked_single_step"]
        else:
            assert res[k]["hooked"] == 0
    if mode == "eager":
        assert torch.equal(res[0]["log_s_single"], res[1]["log_s_single"])      # replicas still identical after the hook-driven steps


def test_bench_two_rank_path_on_one_device():
    """`bench.py --gpus 2 --config cfg4` (BASELINE configs[3]: back-step 256x256x4 sharded over the ranks) through the launcher the
    driver uses, both ranks on this one GPU (TMG_SINGLE_DEVICE=1, gloo - RCCL cannot put two ranks on one device): the N>1 bench
    path - init from the environment, parameter broadcast, persistent gradient buckets all-reduced from the hooks, barrier + MAX
    timing, the one JSON line from rank 0 with the event-pair measurement of the exchange - cannot rot unnoticed."""
    import json
    env = dict(os.environ, TMG_SINGLE_DEVICE="1", TMG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 31200 + (os.getpid() % 1500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(C.ROOT, "bench.py"), "--gpus", "2", "--config", "cfg4", "--batch", "4", "--steps", "3",
           "--warmup", "2", "--no-cpu-baseline", "--no-events"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                          # rank 0 prints exactly one JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    cfg = out["config"]
    assert cfg["global_batch"] == 8 and cfg["world_size_observed"] == 2 and cfg["backend"] == "gloo" and cfg["parallelism"] == "dp2"
    assert abs(out["value"] - 8 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-2 * out["value"]          # whole-job samples / max-over-ranks time
    ar = cfg["allreduce"]
    assert ar is not None and ar["buckets"] >= 1 and ar["bytes"] > 20e6 and ar["exchange_ms"] > 0 and ar["exposed_ms"] >= 0
    assert cfg["loss_last"] == cfg["loss_last"]            # not NaN


def test_rccl_world_size_one_buckets_and_broadcast(tmp_path):
    """the REAL collective backend without a second GPU.  tests/dist_rccl1_worker.py joins a process group of one
    rank with backend "nccl" (RCCL) and runs the flat parameter broadcast, two BPTT windows (buckets issued after the fused gradient
    accumulation) and six single steps (buckets handed to RCCL from the post-accumulate hooks while backward is running; the control
    vector in the last bucket's tail) - against the same steps on an identical model with no bucket at all.  A one-rank all-reduce /
    broadcast is the identity, so losses, gradient norms and parameters must agree to the noise of the float atomics; what the test
    adds over the gloo ones is RCCL's asynchronous handles and its stream ordering against the compute stream."""
    out = str(tmp_path / "rccl1.pt")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(30400 + (os.getpid() % 1500)),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("TMG_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(C.ROOT, "tests", "dist_rccl1_worker.py"), out], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = torch.load(out)
    assert res["backend"] == "nccl" and res["broadcast_exact"]
    a, b = res["rccl"], res["plain"]
    assert a["nbuckets"] > 1 and a["hooked_single_step"] >= 6 and a["second_passes"] == 0
    assert a["deferred_steps"] >= 2          # the live set settled: the later steps ran without a host synchronisation
    assert a["overlap"] is not None and a["overlap"]["exchange_ms"] > 0
    for la, lb in zip(a["loss"], b["loss"]):
        assert abs(la - lb) <= 2e-5 * abs(lb) + 2e-5, (a["loss"], b["loss"])
    for ga, gb in zip(a["gn"], b["gn"]):
        assert abs(ga - gb) <= 5e-4 * gb
    for k, v in b["params"].items():
        assert float((a["params"][k] - v).abs().max()) <= 2e-2 * 8e-3, k          # Adam: |update| <= lr per step, eight steps


def test_bench_forced_bucket_reports_rccl():
    """`bench.py --gpus 1 --force-bucket`: the single-GPU bench step with the gradient buckets and the all-reduce of a multi-GPU rank on
    a one-rank RCCL group; the JSON line names the backend and carries the event-pair measurement of the exchange."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(32700 + (os.getpid() % 1500)))
    env.pop("TMG_DIST_BACKEND", None)
    cmd = [sys.executable, os.path.join(C.ROOT, "bench.py"), "--gpus", "1", "--force-bucket", "--config", "cfg4", "--batch", "4", "--steps", "3",
           "--warmup", "2", "--no-cpu-baseline", "--no-events"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == 1 and cfg["backend"] == "nccl" and cfg["world_size_observed"] == 1
    ar = cfg["allreduce"]
    assert ar is not None and ar["buckets"] >= 1 and ar["bytes"] > 20e6 and ar["exchange_ms"] > 0
    assert cfg["loss_last"] == cfg["loss_last"]
