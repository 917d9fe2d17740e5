#!/usr/bin/env python3
"""bench.py -- flow-field samples/s through the full TM-Glow generative step on MI355X.

One *step* = one batch through one time-step of `TMGlow.sample` (the direction main.py trains through)
-> scalar loss mean(y^2) + mean(logdet)/(noc*H*W) -> backward -> [gradient all-reduce] -> Adam(amsgrad).
Workload at N=1: the metric configuration of BASELINE.json (256x256x4 output, 4 flow levels, K=16,
64 samples per GPU).  `value` = global samples / s with inputs resident in HBM.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

The JSON line also carries
  roofline:     the dominant kernel's algorithmic TFLOP/s (HIP events on the launch stream, live in the
                timed region) against the fp32 matrix peak of gfx950 (157.3 TFLOP/s)
  cpu_baseline: the CPU oracle (oracle/tmglow_oracle.py, a port of the reference's torch-CPU path) timed on
                this box's host cores on a bounded sample of the same workload.
"""
import time
_T0 = time.perf_counter()      # (before the heavy imports: on a fresh box the first `import torch` alone can take a minute or two)
import argparse  # noqa: E402
import json  # noqa: E402
import os  # noqa: E402
import sys  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import common as C  # noqa: E402

CONFIGS = {"M": C.CFG_M, "cfg1": C.CFG1, "cfg2": C.CFG2, "cfg3": C.CFG3, "cfg4": C.CFG_M, "cfg5": C.CFG5, "tiny": C.CFG_TINY}
DEFAULT_BATCH = {"M": 64, "cfg1": 8, "cfg2": 32, "cfg3": 64, "cfg4": 32, "cfg5": 64, "tiny": 2}  # cfg4: global 256 over 8 GPUs
GFLOP_PER_SAMPLE = {"M": 63.72, "cfg1": 2.81, "cfg2": 12.95, "cfg3": 31.86, "cfg4": 63.72, "cfg5": 267.45}  # SURVEY 8-D, fwd+bwd
PEAK_FP32_MFMA_TF = 157.3  # MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0      # HBM3E, same guide


def build_model(cfg, device):
    import contextlib
    from nn.tmGlow import TMGlow
    C.seed_all(12345)
    with contextlib.redirect_stdout(sys.stderr):  # the constructor prints its parameter count (as the reference's does);
        model = TMGlow(**C.build_kwargs(cfg))     # stdout carries exactly one JSON line
    C.perturb_(model, 7, *C.perturb_scales(cfg))
    return model.to(device).train()


def host_cores():
    """Host cores THIS process may use: the scheduler affinity mask, cut by the cgroup CPU quota (a container that sees 128 CPUs in
    os.cpu_count() may own 8 of them - 32 OpenMP threads on 8 cores spin against each other), capped at 32 (more threads are slower
    on these small convolutions)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(32, n))


def _cpu_baseline_worker(name, threads, B, max_steps, budget_s, t_spawn):
    """Runs in a child process: the oracle on `threads` host threads.  Every phase of the child is stamped (the driver's box spent
    185 s in the cfg1 child of round 4 for 4.9 s of timed steps and nothing said where)."""
    ph, last = {}, [time.time()]

    def stamp(k):
        now = time.time()
        ph[k] = round(now - last[0], 2)
        last[0] = now
    ph["spawn_to_worker_entry (interpreter start, import torch, import bench)"] = round(time.time() - t_spawn, 2)
    from oracle import tmglow_oracle as O
    from nn.tmGlow import TMGlow
    stamp("import oracle + package")
    cfg = CONFIGS[name]
    torch.set_num_threads(threads)
    C.seed_all(12345)
    m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 7, *C.perturb_scales(cfg))
    P = O.params_from_state_dict(m.state_dict())
    del m
    params = list(O.trainable(P).values())
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-8, amsgrad=True)
    stamp("model construction (QR / LU initialisation on the host)")
    Hin, Win = cfg["_in_hw"]
    up = cfg["_up"]
    g = torch.Generator().manual_seed(12345)
    x = torch.randn(B, cfg["in_features"], Hin, Win, generator=g)
    st = O.init_lstm_states(cfg, torch.arange(B), [Hin * up, Win * up])
    stamp("inputs + seed states")

    def step():
        opt.zero_grad()
        y, ld, _ = O.tmglow_sample(P, cfg, x, st)
        C.loss_reverse(y, ld).backward()
        opt.step()

    tw = time.time()
    step()  # warm-up (allocator, thread pool)
    tw = time.time() - tw
    stamp("warm-up step")
    n, t0 = 0, time.time()
    while n < max_steps and (n == 0 or time.time() - t0 < budget_s):
        step()
        n += 1
    dt = time.time() - t0
    stamp("timed steps")
    print(json.dumps({"value": B * n / dt, "steps": n, "batch": B, "threads": threads, "torch_threads": torch.get_num_threads(),
                      "os_cpu_count": os.cpu_count(), "warmup_step_s": round(tw, 2), "timed_s": round(dt, 2), "phases_s": ph}))


def cpu_baseline(name, hard_timeout_s=300, max_steps=12, budget_s=25.0):
    """The CPU oracle (a port of the reference's torch-CPU path) on this box's host cores, on a bounded
    sample of the same workload: sample() + backward + Adam.  Runs in a child process under a hard timeout;
    thread count = the cores this process may use (host_cores(): affinity mask and cgroup quota), capped at 32."""
    import subprocess
    threads = host_cores()
    B = {"M": 8, "cfg4": 8, "cfg3": 8, "cfg5": 2}.get(name, DEFAULT_BATCH[name])     # the metric shape at batch 8
    code = "import time; t=%r; import bench; bench._cpu_baseline_worker(%r, %d, %d, %d, %f, t)" % (time.time(), name, threads, B, max_steps, budget_s)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=hard_timeout_s)
        res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return {"value": round(res["value"], 4), "unit": "samples/s", "cores": threads, "kind": "port",
                "sample": "config %s at batch %d, %d timed step(s) after 1 warm-up, torch CPU fp32 oracle, sample()+backward+Adam"
                          % (name, res["batch"], res["steps"]), "warmup_step_s": res.get("warmup_step_s"), "timed_s": res.get("timed_s"),
                "os_cpu_count": res.get("os_cpu_count"), "child_wall_s": round(time.time() - t0, 2), "child_phases_s": res.get("phases_s")}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "samples/s", "cores": threads, "kind": "port", "sample": "failed: %s" % type(e).__name__,
                "child_wall_s": round(time.time() - t0, 2)}


def pmc_traffic(kernel):
    """HBM bytes per launch of the event class `kernel` from the committed rocprofv3 PMC passes (profiles/r<N>_traffic_per_launch.json of
    the latest round: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, launch-weighted over the kernels / template instances the class
    groups)."""
    import glob
    import re
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_per_launch.json")),
                   key=lambda f: (int(re.match(r"r(\d+)", os.path.basename(f)).group(1)), len(os.path.basename(f))))
    if not cands:
        return None
    path = [f for f in cands if re.match(r"r\d+_traffic", os.path.basename(f))][-1] if any(re.match(r"r\d+_traffic", os.path.basename(f)) for f in cands) else cands[-1]
    tab = json.load(open(path))
    if kernel.startswith("wino_fwd"):
        pats = [("wino_fwd_kernel", None), ("wino_nn_kernel", None)]
    elif kernel.startswith("wino_wgrad"):
        pats = [("wino_wgrad_kernel", None)]
    else:
        base, _, tail = kernel.partition("<")
        pats = [(base, tail.rstrip(">").split(","))]
    tot = n = 0
    for k, v in tab.items():
        if not isinstance(v, dict):
            continue
        for base, want in pats:
            if not k.startswith(base):
                continue
            if want is not None:
                if not k.startswith(base + "<"):
                    continue
                have = [a.strip() for a in k[len(base) + 1:].rstrip(">").split(",")]
                if not (len(have) == len(want) and all(w == "*" or w == h for w, h in zip(want, have))):
                    continue
            tot += (v["fetch_bytes_corrected"] + v["write_bytes"]) * v["launches"]
            n += v["launches"]
    return round(tot / n) if n else None


_PHASES = []


def _phase(name):
    """Host wall-clock bookkeeping of the whole run (the driver's clock around bench.py covers imports, model construction, seed
    states, warm-up, the event passes and the CPU baselines - not only the timed region): printed to stderr as the run goes and
    reported as `host_phases_s` in the JSON line."""
    now = time.perf_counter()
    last = _PHASES[-1][2] if _PHASES else _T0
    _PHASES.append((name, round(now - last, 2), now))
    print("[bench %7.1f s] %s: %.2f s" % (now - _T0, name, now - last), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)  # >= 20 timed steps after 5 warm-up
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="M", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip per-launch HIP-event timing")
    ap.add_argument("--direction", default="sample", choices=["sample", "forward"],
                    help="sample: the generative direction the reference trains through (the metric); forward: density direction "
                         "forward(x, y) with loss -mean(logp)/(noc*H*W), reported beside it (SURVEY 8-D)")
    ap.add_argument("--adam", default="hip", choices=["fused", "foreach", "hip"],
                    help="implementation of the Adam update: torch foreach (what main.py constructs), torch fused, or tmg_optim.HipAdam (one launch)")
    ap.add_argument("--graph", action="store_true", help="forward + loss + backward of the step as one hipGraph replay (tmg_dist.CapturedWindow; N=1 only); the optimizer step stays eager")
    ap.add_argument("--mix", default=None, choices=["f32", "f16"],
                    help="arithmetic of the 1x1 channel mixes: f32 MFMA (default) or fp16 operands / fp32 accumulation (the variant "
                         "BASELINE.json configs[4] names; not faster here - cfg5 reports it beside the fp32 line)")
    ap.add_argument("--wino", default="f32", choices=["f32", "bf16x3"],
                    help="arithmetic of the wide Winograd contractions in the TIMED region: fp32 MFMA (default, the headline) or the opt-in "
                         "bf16x3 form (exact three-way bf16 split of both fp32 operands, six MFMAs per accumulator tile, fp32 accumulate: "
                         "fp32-grade error, tests/test_model_parity.py::test_metric_configuration_with_bf16x3_winograd_matches_oracle); "
                         "`dtype` names it")
    ap.add_argument("--force-bucket", action="store_true",
                    help="N = 1 only: a process group of ONE rank on the real collective backend (RCCL) with the gradient buckets, their hooks "
                         "and the all-reduce in the step - what a multi-GPU rank runs, minus the peers; config.allreduce / backend report it")
    args = ap.parse_args()
    if args.force_bucket and args.gpus == 1:
        os.environ["TMG_FORCE_DIST"] = "1"

    _phase("imports (torch, tests/common)")
    import tmg_dist
    import tmg_hip
    rank, world, local = tmg_dist.init_from_env()
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU product path"
    tmg_hip.lib()
    if os.environ.get("TMG_SINGLE_DEVICE"):   # functional test of the N>1 path on a one-GPU box: every rank on cuda:0
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = CONFIGS[args.config]
    B = args.batch or DEFAULT_BATCH[args.config]
    import tmg_ops
    # fp32 mixes everywhere except cfg5: BASELINE configs[4] NAMES fp16-operand 1x1 mixes, so that is what its line measures (tested:
    # tests/test_model_parity.py::test_cfg5_stated_batch_with_fp16_mixes).  The variant is not faster here - the stand-alone mixes are
    # bandwidth kernels on fp32 activations up to 32 channels (1.00x since round 6: profiles/r6_mix_f16_vs_f32.txt) - and the line says so: `mix_f16_speedup` is the fp32-mix step time over
    # the fp16-mix one on the same workload; `--mix f32` measures the fp32 mixes with the fp16 variant beside it (`mix_f16_variant`).
    mix = args.mix or ("f16" if args.config == "cfg5" else "f32")
    tmg_ops.set_mix_precision(mix)
    tmg_ops.set_winograd_precision(args.wino)
    _phase("process group, library load")
    model = build_model(cfg, dev)
    _phase("model construction + upload")
    forced = bool(args.force_bucket and world == 1)
    # RCCL prints a start-up banner (version, host, library path) to the C-level stdout when its communicator is created - at the first
    # collective.  The contract is ONE JSON line on rank 0's stdout: the first collective runs with file descriptor 1 pointing at stderr.
    sys.stdout.flush()
    fd1 = os.dup(1)
    os.dup2(2, 1)
    try:
        tmg_dist.broadcast_parameters(model, force=forced)
        if world > 1 or forced:
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(fd1, 1)
        os.close(fd1)
    bucket = tmg_dist.GradBucket(model.parameters(), measure=True, force=forced) if (world > 1 or forced) else None
    use_graph = args.graph and world == 1
    # the reference's optimizer (main.py:78: Adam, weight decay 1e-8, amsgrad); `fused` = torch's single-kernel multi-tensor
    # implementation of the same update (about 100 launches per step fewer than the default foreach one)
    if args.adam == "hip":
        from tmg_optim import HipAdam       # the same update, one launch for all ~1 000 parameter tensors
        opt = HipAdam(model.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True,
                               fused=(args.adam == "fused") or None)
    Hin, Win = cfg["_in_hw"]
    up = cfg["_up"]
    g = torch.Generator().manual_seed(12345 + rank)
    x = torch.randn(B, cfg["in_features"], Hin, Win, generator=g).to(dev)
    states = model.initLSTMStates(torch.arange(B) + rank * B, [Hin * up, Win * up])
    # the recurrent states a training loop carries are the model's own outputs (channels-last strides, re-anchored half-way to
    # the seed states, trainFlowParallel.py:294-297); hand the seed states over in that layout too, once, outside the timed region
    states = [(h.contiguous(memory_format=torch.channels_last), c.contiguous(memory_format=torch.channels_last)) for h, c in states]

    y_fwd = torch.randn(B, cfg["out_features"], Hin * up, Win * up, generator=g).to(dev) if args.direction == "forward" else None
    torch.cuda.synchronize()
    _phase("inputs + seed states (host RNG of the reference, once per seed)")

    def step():
        opt.zero_grad(set_to_none=True)
        if y_fwd is None:
            y, ld, _ = model.sample(x, states)
            loss = tmg_ops.reverse_loss(y, ld)      # = tests/common.py::loss_reverse (mean(y^2) + mean(logdet) / (noc H W)), two launches
        else:
            _, logp, _, _ = model.forward(x, y_fwd, states)
            loss = C.loss_forward(logp, y_fwd)
        loss.backward()
        if bucket is not None:
            bucket.allreduce_mean()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graph = None
    if use_graph:
        # forward + loss + backward recorded into one hipGraph (tmg_dist.CapturedWindow: a one-time-step window) and replayed; the
        # optimizer step stays eager.  Every replay draws fresh latents (graph-safe Philox offsets) and reads the weights in place.
        def body(x_, st_):
            if y_fwd is None:
                y_, ld_, _ = model.sample(x_, st_)
                return tmg_ops.reverse_loss(y_, ld_), ()
            _, logp_, _, _ = model.forward(x_, y_fwd, st_)
            return C.loss_forward(logp_, y_fwd), ()
        graph = tmg_dist.CapturedWindow(model, body, (x, states))

        def step():  # noqa: F811
            opt.zero_grad(set_to_none=True)
            loss, _ = graph(x, states)
            opt.step()
            return loss
        args.no_events = True  # per-launch events cannot be recorded inside a replayed graph
    for _ in range(args.warmup):
        step()
    barrier()
    _phase("warm-up steps (first launches load the code objects)")
    # which matrix-core kernel dominates the step: two untimed steps with every contraction launch bracketed by HIP events;
    # inside the timed region only THAT kernel keeps its events (a handful of pairs per step: timing all ~500 contraction
    # launches of a step cost 2 % of the headline number)
    prof_all, dom = {}, None
    if not args.no_events:
        tmg_hip.prof_enable(1)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        prof_all = tmg_hip.prof_collect()
        tmg_hip.prof_enable(False)
        cand = {k: v for k, v in prof_all.items() if not k.startswith("conv 1x1")}
        if cand:
            dom = max(cand.items(), key=lambda kv: kv[1][1])[0]
    barrier()
    _phase("event pass over all contraction launches (2 steps)")
    torch.cuda.reset_peak_memory_stats(dev)
    if dom is not None:
        tmg_hip.prof_enable(100 + tmg_hip.prof_kernel_id(dom))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    _phase("TIMED REGION (%d steps)" % args.steps)
    prof, prof_steps = {}, {}
    if not args.no_events:
        live = tmg_hip.prof_collect()      # the dominant kernel, timed live inside the timed region
        tmg_hip.prof_enable(False)
        prof = dict(prof_all)              # every other contraction kernel: from the two untimed steps before it
        prof_steps = {k: 2 for k in prof}
        prof.update(live)
        prof_steps.update({k: args.steps for k in live})
        # the bandwidth-bound kernel classes are timed in a short pass of their own AFTER the timed region (their ~2 000 extra
        # event pairs per step would otherwise be charged to the headline number)
        tmg_hip.prof_enable(2)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        hb = {k: v for k, v in tmg_hip.prof_collect().items() if k.startswith("hbm:")}
        prof.update(hb)
        prof_steps.update({k: 3 for k in hb})
        tmg_hip.prof_enable(False)
    peak_gb = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    _phase("event pass over the bandwidth-bound classes (3 steps)")
    mix_speedup = None
    dens = None
    if args.direction == "sample" and graph is None and args.config in ("M", "cfg4") and not args.no_events:
        # the density direction forward(x, y) with loss -mean(logp)/(noc*H*W) ("also report"), a few steps AFTER the timed
        # region on the same model and inputs
        yd = torch.randn(B, cfg["out_features"], Hin * up, Win * up, generator=g).to(dev)

        def dstep():
            opt.zero_grad(set_to_none=True)
            _, logp, _, _ = model.forward(x, yd, states)
            C.loss_forward(logp, yd).backward()
            if bucket is not None:
                bucket.allreduce_mean()
            opt.step()
        for _ in range(2):
            dstep()
        barrier()
        td = time.perf_counter()
        for _ in range(5):
            dstep()
        barrier()
        td = (time.perf_counter() - td) / 5
        dens = {"what": "forward(x,y)+logp+backward+Adam, same model and batch, 5 steps after 2 warm-up (rank-local clock)",
                "ms_per_step": round(1e3 * td, 3), "value": round(B * world / td, 2), "unit": "samples/s"}
        del yd
        _phase("density direction (7 steps)")
    f16_variant = None
    if args.config == "cfg5" and mix == "f32" and graph is None and args.direction == "sample":
        def few(n=4):
            step()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n
        t32 = few()
        tmg_ops.set_mix_precision("f16")
        t16 = few()
        tmg_ops.set_mix_precision("f32")
        f16_variant = {"what": "the same step with fp16-operand / fp32-accumulate 1x1 mixes (BASELINE configs[4]), 4 steps each after the timed region",
                       "ms_per_step_f16": round(1e3 * t16, 3), "ms_per_step_f32": round(1e3 * t32, 3), "speedup_of_f16": round(t32 / t16, 4)}
    wino3 = None
    if args.config in ("M", "cfg4") and graph is None and args.direction == "sample" and not args.no_events and args.wino == "f32":
        # secondary field: the same step with the wide Winograd contractions (ConvLSTM gate conv, level-wide
        # conditioning conv, out-conv input gradient) on the bf16 matrix pipe at fp32 accuracy - every fp32 operand split exactly into
        # three bf16 parts, six of the nine part products, fp32 accumulation; transforms in fp32.  Opt-in
        # (tmg_ops.set_winograd_precision); the headline above is fp32 MFMA throughout.  A few steps each, alternating, after the
        # timed region.
        def few(n=4):
            step()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n
        ts = {"f32": [], "bf16x3": []}
        for _ in range(2):
            for kind in ("f32", "bf16x3"):
                tmg_ops.set_winograd_precision(kind)
                ts[kind].append(few())
        tmg_ops.set_winograd_precision("f32")
        t32w, t3w = min(ts["f32"]), min(ts["bf16x3"])
        wino3 = {"what": "the same step with the wide Winograd contractions as six bf16 MFMAs per accumulator tile on a three-way exact split of "
                         "both fp32 operands (fp32-grade error: tools/micro/wino_bf16x3.hip, tests), 2 x 4 steps each, alternating, after the timed region",
                 "dtype": "f32 operands, bf16x3 split on the matrix pipe, f32 accumulate", "ms_per_step": round(1e3 * t3w, 3),
                 "ms_per_step_f32_mfma": round(1e3 * t32w, 3), "value": round(B * world / t3w, 2), "unit": "samples/s",
                 "speedup": round(t32w / t3w, 4)}
        _phase("bf16x3 Winograd variant (16 steps)")
    if mix == "f16" and graph is None:
        # the fp16-operand 1x1 mixes against this package's own fp32 mixes on the same workload, a few steps each AFTER the timed
        # region (every rank runs them: the steps contain the gradient exchange).  The stand-alone mixes read and write fp32
        # activations either way - bandwidth kernels at K = C <= 256 - so the variant buys no time; the line says so.
        def few(n=4):
            step()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n
        t16 = few()
        tmg_ops.set_mix_precision("f32")
        t32 = few()
        tmg_ops.set_mix_precision("f16")
        mix_speedup = round(t32 / t16, 4)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    allreduce_report = bucket.overlap_report() if bucket is not None else None     # (event pairs: no collective inside)
    backend = torch.distributed.get_backend() if (world > 1 or forced) else None
    if world > 1 or forced:
        # every rank leaves the process group HERE, together: nothing below communicates (round 4: ranks != 0 returned while rank 0
        # went on for minutes with the group alive)
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    value = B * world * args.steps / dt
    roof = None
    hbm = {k: v for k, v in prof.items() if k.startswith("hbm:")}
    mix1 = prof.get("conv 1x1 (invertible channel mix, fp32 MFMA)")
    prof = {k: v for k, v in prof.items() if not k.startswith("hbm:")}
    if prof:
        name = dom if dom in prof else max(prof.items(), key=lambda kv: kv[1][1] / prof_steps[kv[0]])[0]
        cnt, ms, fl = prof[name]
        alg = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # `achieved` / `frac` are what the MATRIX PIPE executes.  The Winograd kernels (minimal filtering F(2x2,3x3) / F(3x3,2x2)) issue
        # 16 matrix-core multiplies per 36 of the direct 3x3 algorithm, so their executed rate is 16/36 of the algorithmic
        # (direct-equivalent) rate, which is reported beside it and can exceed the peak; for direct kernels the two coincide.
        wino = name.startswith("wino")
        ach = alg * (16.0 / 36.0 if wino else 1.0)
        roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TF, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_FP32_MFMA_TF, 4), "traffic": None, "launches": cnt,
                "algorithmic_tflops": round(alg, 3),
                "algorithm": ("Winograd F(2x2,3x3) / F(3x3,2x2): 16 matrix-core multiplies per 36 algorithmic ones; achieved = executed"
                              if wino else "direct implicit GEMM: executed = algorithmic flops"),
                "avg_launch_us": round(1e3 * ms / max(cnt, 1), 2), "time_share_of_step": round(ms * 1e-3 / dt, 4),
                "measured": "HIP events on the launch stream inside the timed region (this kernel only; the table below comes from "
                            "two untimed steps with every contraction launch bracketed)",
                "all_contraction_kernels": {k: {"launches_per_step": round(v[0] / prof_steps[k], 1), "ms_per_step": round(v[1] / prof_steps[k], 3),
                                                "algorithmic_tflops": round(v[2] / max(v[1], 1e-9) / 1e9, 2),
                                                "executed_tflops": round(v[2] / max(v[1], 1e-9) / 1e9 * (16.0 / 36.0 if k.startswith("wino") else 1.0), 2)}
                                            for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1] / prof_steps[kv[0]])}}
        roof["traffic"] = pmc_traffic(name) if args.config == "M" and B == 64 else None
        # the second matrix-core class of the step beside the dominant one: the Winograd weight gradients (F(3x3, 2x2)) - or the Winograd
        # forward / input-gradient class when the weight gradients dominate - with its own executed fraction of the fp32 matrix peak
        other = [k for k in prof if k.startswith("wino") and k != name]
        if other:
            k2 = max(other, key=lambda k: prof[k][1] / prof_steps[k])
            c2, ms2, fl2 = prof[k2]
            alg2 = fl2 / (ms2 * 1e-3) / 1e12 if ms2 > 0 else 0.0
            roof["second_class"] = {"kernel": k2, "launches_per_step": round(c2 / prof_steps[k2], 1), "ms_per_step": round(ms2 / prof_steps[k2], 3),
                                    "algorithmic_tflops": round(alg2, 3), "achieved": round(alg2 * 16.0 / 36.0, 3), "peak": PEAK_FP32_MFMA_TF,
                                    "frac": round(alg2 * 16.0 / 36.0 / PEAK_FP32_MFMA_TF, 4), "traffic": pmc_traffic(k2) if args.config == "M" and B == 64 else None,
                                    "measured": "HIP events on the launch stream, %d step(s) %s the timed region" % (
                                        prof_steps[k2], "inside" if prof_steps[k2] == args.steps else "before")}
        if args.config in GFLOP_PER_SAMPLE:
            e2e = value / world * GFLOP_PER_SAMPLE[args.config] / 1e3
            roof["end_to_end_tflops_per_gpu"] = round(e2e, 2)
            roof["end_to_end_frac"] = round(e2e / PEAK_FP32_MFMA_TF, 4)
    opt_name = {"hip": "HipAdam (one-launch Adam, same update as torch.optim.Adam)", "foreach": "torch.optim.Adam (foreach)",
                "fused": "torch.optim.Adam (fused)"}[args.adam]
    out = {"metric": ("flow-field samples/sec (fwd+log-det+bwd), 64x256x256x4" if args.config == "M" else "flow-field samples/sec (fwd+log-det+bwd)")
                     + "; optimizer step: " + opt_name,
           "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": ("f32" if mix == "f32" else "f32 (stand-alone 1x1 mixes: fp16 operands, fp32 accumulate; the mixes inside the fused coupling kernels: fp32 MFMA)")
                    + ("" if args.wino == "f32" else " (wide Winograd contractions: f32 operands split into three bf16 parts on the matrix pipe, f32 accumulate)"),
           "data": "synthetic",
           "config": {"workload": "tmglow %s: %s+logdet+backward+Adam, out %dx%dx%d, L=%d, K=%d, batch %d/GPU" % (
               args.config, "sample()" if args.direction == "sample" else "forward(x,y)", Hin * up, Win * up, cfg["out_features"], len(cfg["glow_blocks"]), cfg["glow_blocks"][0], B),
               "global_batch": B * world, "parallelism": "dp%d" % world, "world_size_observed": world,
               "backend": backend, "rank0_device": torch.cuda.get_device_name(dev) + " cuda:%d" % local,
               "mix_precision": mix, "optimizer": "Adam(amsgrad, wd 1e-8): %s" % opt_name, "allreduce": allreduce_report, "loss_last": float(loss.detach()), "launch": "hipGraph replay" if graph is not None else "eager"},
           "peak_mem_gb": round(peak_gb, 2), "roofline": roof}
    if args.wino != "f32" and isinstance(roof, dict):
        roof["note_wino_bf16x3"] = ("--wino bf16x3: the wide launches of this class ran six bf16 MFMAs per accumulator tile (bf16 pipe, 2.5 PF dense); "
                                    "`peak` / `frac` above are still priced against the fp32 MFMA peak and the fp32 algorithm's 16 / 36 products")
    if dens is not None:
        out["density_direction"] = dens
    if wino3 is not None:
        out["wino_bf16x3_variant"] = wino3
    if f16_variant is not None:
        out["mix_f16_variant"] = f16_variant
    if mix_speedup is not None:
        out["mix_f16_speedup"] = mix_speedup   # step time with fp32 mixes / step time with the fp16-operand mixes this line was measured with
    if hbm:
        # bandwidth-bound kernel classes: algorithmic HBM bytes / HIP-event time on the launch stream, against the 8 TB/s HBM3E peak
        out["hbm_kernel_classes"] = {k[5:]: {"launches_per_step": round(v[0] / 3, 1), "ms_per_step": round(v[1] / 3, 3), "GB/s": round(v[2] / max(v[1], 1e-9) / 1e6, 1),
                                              "frac_of_8TBps": round(v[2] / max(v[1], 1e-9) / 1e6 / PEAK_HBM_GBS, 4)}
                                     for k, v in sorted(hbm.items(), key=lambda kv: -kv[1][1])}
    if mix1 and mix1[1] > 0:
        tf = mix1[2] / mix1[1] / 1e9
        out["mix_1x1_mfma"] = {"launches_per_step": mix1[0] / 2, "ms_per_step": round(mix1[1] / 2, 3), "tflops": round(tf, 2), "frac_of_fp32_mfma_peak": round(tf / PEAK_FP32_MFMA_TF, 4),
                               "note": "stand-alone 1x1 mixes only (wide levels, LSTM blocks); on the narrow levels the mix runs inside cpl_fwd_kernel"}
    if not args.no_cpu_baseline and world == 1:      # the contract: rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(args.config)
        _phase("cpu_baseline child (%s)" % args.config)
        if args.config != "cfg1":   # BASELINE configs[0], the reference's own CPU-runnable case, at its stated batch 8
            out["cpu_baseline_cfg1"] = cpu_baseline("cfg1", max_steps=10, budget_s=15.0)
            _phase("cpu_baseline child (cfg1)")
    out["host_phases_s"] = {n: d for n, d, _ in _PHASES}
    out["host_total_s"] = round(time.perf_counter() - _T0, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
