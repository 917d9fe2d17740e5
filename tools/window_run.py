#!/usr/bin/env python3
"""The trainer's real unit of work at the metric shape: one mini-batch of B samples
through `TrainFlow.trainParallel` = a 10-step BPTT window of `model.sample` (256x256 output, 4 flow levels, K = 16, default
widths; 3 output channels - the physics-constrained loss is defined on (u_x, u_y, p)) -> TMGLowLoss -> one backward through all
ten time-steps -> clip -> Adam(amsgrad) -> state re-anchoring.  Prints one JSON line with the window time and the peak HBM use.

    python tools/window_run.py [--batch 64] [--tsteps 10] [--windows 3]
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import common as C  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--tsteps", type=int, default=10)
    ap.add_argument("--windows", type=int, default=3)
    ap.add_argument("--noc", type=int, default=3, choices=[3, 4],
                    help="3: the trainer's real channel count, TrainFlow.trainParallel with the physics-constrained loss; 4: the metric "
                         "configuration M (256x256x4) with bench.py's loss through tmg_dist.train_window (the physics loss is defined "
                         "for 3 channels only)")
    ap.add_argument("--adam", default="torch", choices=["torch", "hip"],
                    help="torch: torch.optim.Adam as main.py:78 constructs it; hip: tmg_optim.HipAdam (the same update in one launch)")
    ap.add_argument("--capture", action="store_true",
                    help="forward passes + loss + backward of a window as one hipGraph replay (tmg_dist.CapturedWindow); clip, optimizer "
                         "step and state re-anchoring stay eager")
    ap.add_argument("--eager", action="store_true", help="windows launched eagerly (capture off); neither flag: the trainer's default - a "
                    "window shape is recorded when it comes round the second time, torch's Adam is adopted as the one-launch HipAdam")
    ap.add_argument("--no-adopt", action="store_true", help="keep torch.optim.Adam's own step (TMG_NO_HIP_ADAM)")
    ap.add_argument("--config", default="M", choices=["M", "cfg5"], help="M: 256x256 output, 4 levels; cfg5 (BASELINE configs[4]): 512x512x4, 5 levels "
                    "(implies --noc 4: the window runs through tmg_dist.train_window with bench.py's loss)")
    ap.add_argument("--recompute", action="store_true", help="tmg_ops.set_recompute(True): the narrow levels keep no per-layer activations, "
                    "backward rebuilds them from the level outputs (capacity mode)")
    a = ap.parse_args()
    if a.config == "cfg5":
        a.noc = 4
    if a.no_adopt:
        os.environ["TMG_NO_HIP_ADAM"] = "1"

    def make_opt(params):
        if a.adam == "hip":
            from tmg_optim import HipAdam
            return HipAdam(params, lr=1e-3, weight_decay=1e-8, amsgrad=True)
        return torch.optim.Adam(params, lr=1e-3, weight_decay=1e-8, amsgrad=True)
    import contextlib
    from nn.tmGlow import TMGlow
    from nn.trainFlowParallel import TrainFlow
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    cfg = dict(C.CFG5 if a.config == "cfg5" else C.CFG_M, in_features=4, out_features=a.noc)   # backward-step trainer: nic = 4 (3 fields + inlet velocity), noc = 3
    if a.recompute:
        import tmg_ops
        tmg_ops.set_recompute(True)
    C.seed_all(12345)
    with contextlib.redirect_stdout(sys.stderr):
        model = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(model, 7, *C.perturb_scales(cfg))
    model.out_std, model.out_mu = torch.tensor([1.0, 1.0, 1.0]), torch.tensor([0.0, 0.0, 0.0])
    model.to(dev).train()
    B, T = a.batch, a.tsteps
    h, w = cfg["_in_hw"]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T * a.windows, 4, h, w, generator=g).to(dev)
    if a.noc == 4:
        import tmg_dist
        opt = make_opt(model.parameters())
        key = model.initLSTMStates(torch.arange(B), [2 * h, 2 * w])

        cw = None
        if a.capture:
            cw = tmg_dist.CapturedWindow(model, tmg_dist.window_body(model, C.loss_reverse),
                                         ([x[:, t] for t in range(T)], [(hh.clone(), cc.clone()) for hh, cc in key]))

        def run():
            st = [(hh.clone(), cc.clone()) for hh, cc in key]
            tot = 0.0
            for wi in range(a.windows):
                loss, _, st, _ = tmg_dist.train_window(model, opt, [x[:, wi * T + t] for t in range(T)], st, key, C.loss_reverse,
                                                       max_grad_norm=0.01, captured=cw)
                tot = tot + loss
            return tot
        run()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        t0 = time.perf_counter()
        loss = run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"what": "tmg_dist.train_window, %d BPTT window(s) of %d sample() steps, batch %d, config %s (%dx%dx4, L=%d, K=16), "
                          "loss of bench.py summed over the steps, clip, Adam(amsgrad), state re-anchoring" % (
                              a.windows, T, B, a.config, 2 * h, 2 * w, len(cfg["glow_blocks"])), "recompute": bool(a.recompute),
                          "reserved_mem_gb": round(torch.cuda.memory_reserved(dev) / 2 ** 30, 2),
                          "seconds_per_window": round(dt / a.windows, 4), "sample_steps_per_s": round(B * T * a.windows / dt, 2),
                          "peak_mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2), "loss_sum": float(loss),
                          "warmup": "one untimed call of the same shape", "captured": bool(a.capture)}))
        return
    y = torch.randn(B, T * a.windows, 3, 2 * h, 2 * w, generator=g).to(dev)
    seeds = torch.arange(B)
    args = SimpleNamespace(beta=200.0, dx=2. / 64, dy=2. / 64, max_grad_norm=0.01)
    if a.capture or a.eager:
        args.capture_window = bool(a.capture)      # (absent: the trainer's default)
    opt = make_opt(model.parameters())
    trainer = TrainFlow(args, model, [(x, y, seeds)], None)
    # one call = `windows` BPTT windows of T steps each (trainParallel walks tmax // tback windows of a mini-batch); the first
    # call is an untimed warm-up (the caching allocator grows to the window's working set with one hipMalloc per block)
    trainer.trainParallel(model, opt, epoch=0)
    torch.cuda.synchronize()
    ti = time.perf_counter()
    model.initLSTMStates(seeds, [2 * h, 2 * w])     # what trainParallel does once per mini-batch: now a gather of HBM-resident seed states
    torch.cuda.synchronize()
    t_init = time.perf_counter() - ti
    ti = time.perf_counter()
    model._draw_seed_states([int(s_) for s_ in seeds.tolist()], [2 * h, 2 * w])
    t_cold = time.perf_counter() - ti
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    loss = trainer.trainParallel(model, opt, epoch=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0        # INCLUDES the mini-batch's seed states (initLSTMStates is called inside trainParallel)
    peak = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    print(json.dumps({"what": "TrainFlow.trainParallel, %d BPTT window(s) of %d sample() steps, batch %d, 256x256x3 output, L=4, K=16" % (
        a.windows, T, B), "seconds_per_window": round(dt / a.windows, 4), "sample_steps_per_s": round(B * T * a.windows / dt, 2),
        "peak_mem_gb": round(peak, 2), "reserved_mem_gb": round(torch.cuda.memory_reserved(dev) / 2 ** 30, 2), "loss_sum": float(loss), "warmup": "one untimed call of the same shape", "optimizer": a.adam,
        "captured": bool(trainer._captured), "capture_mode": trainer._capture, "capture_failed": list(trainer._capture_failed.values()),
        "optimizer_class": type(opt).__name__,
        "lstm_state_init_s_per_minibatch": round(t_init, 4), "lstm_state_host_draw_s_first_use": round(t_cold, 3),
        "note": "window time INCLUDES the per-mini-batch seed states: every distinct seed (the loaders draw them from random_(0, 1000)) is "
                "drawn once on the host with the reference's CPU generators (tmGlow.py:481-509) and kept in HBM; later mini-batches gather"}))


if __name__ == "__main__":
    main()
