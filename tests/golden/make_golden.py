#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference (build container only).

    python tests/golden/make_golden.py            # needs /root/reference/tmglow

The reference never travels to the GPU box; only the .npz files written here do.  Nothing in this
script is reference source: it drives the reference's public API (`nn.tmGlow.TMGlow` etc.) and
records inputs / outputs / gradients.

Shim (caveat 1): the reference was written for torch 1.6 and clamps a
`chunk()` view in place (flowUtils.py:163), which torch 2.10 rejects in grad mode.  We wrap
`Tensor.chunk` to hand out cloned chunks *in this process only*; values and gradients are
unchanged (clone is the identity).
Caveat 2: InvertibleConv1x1LU.reverse caches `self.W` keyed on `log_s_old` (glowConv.py:157,
209-212); we invalidate the key before every reverse call so each call rebuilds W.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
REF = "/root/reference/tmglow"

_orig_chunk = torch.Tensor.chunk


def _cloned_chunk(self, *a, **k):
    return tuple(c.clone() for c in _orig_chunk(self, *a, **k))


def import_reference():
    torch.Tensor.chunk = _cloned_chunk
    torch.chunk = lambda t, *a, **k: _cloned_chunk(t, *a, **k)
    sys.path.insert(0, REF)
    import nn.tmGlow as ref_tmglow  # noqa
    return ref_tmglow


def invalidate_w_cache(model):
    for m in model.modules():
        if hasattr(m, "log_s_old"):
            m.log_s_old.fill_(1.0e9)


from common import (CFG_TINY, CFG_TINY3, CFG_TINY5, CFG1, build_kwargs, perturb_, loss_forward, loss_reverse,  # noqa: E402
                    tensor_checksums, seed_all)


def sd_to_np(sd, prefix):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def grads_to_np(model, prefix):
    out = {}
    for k, p in model.named_parameters():
        if p.grad is not None:
            out[prefix + k] = p.grad.detach().cpu().numpy().copy()
    return out


def states_to_np(states, prefix):
    out = {}
    for i, (h, c) in enumerate(states):
        out["%s%d.h" % (prefix, i)] = h.detach().numpy().copy()
        out["%s%d.c" % (prefix, i)] = c.detach().numpy().copy()
    return out


GRAD_SAMPLE_STRIDE = 13   # compact fixtures keep every 13th entry of every gradient tensor (in named_parameters order)


def _compact(d):
    """Replace big state / eps tensors by a checksum + strided sample (cfg1-sized fixture)."""
    out = {}
    for k, v in d.items():
        if isinstance(v, np.ndarray) and v.size > 70000 and (".h_out." in k or ".eps." in k or k.startswith("h_in.")):
            flat = v.reshape(-1).astype(np.float64)
            out[k + "#sum"] = np.array([flat.sum(), np.abs(flat).sum()])
            out[k + "#sample"] = v.reshape(-1)[::97].copy()
        else:
            out[k] = v
    return out


def model_case(ref, cfg, B, fname, store_weights=True, scales=(0.05, 0.1, 0.05), full_grads=True, compact=False):
    """forward(x,y,h,return_eps) + backward, then reconstruct(x,h,eps) + backward."""
    seed_all(12345)
    model = ref.TMGlow(**build_kwargs(cfg))
    perturb_(model, 7, *scales)
    model.train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    Hin, Win = cfg["_in_hw"]
    up = cfg["_up"]
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, cfg["in_features"], Hin, Win, generator=g)
    y = torch.randn(B, cfg["out_features"], Hin * up, Win * up, generator=g)
    h_in = model.initLSTMStates(torch.arange(B), [Hin * up, Win * up])

    out = {"x": x.numpy(), "y": y.numpy()}
    out.update(states_to_np(h_in, "h_in."))
    if store_weights:
        out.update(sd_to_np(sd0, "sd."))
    else:
        cs = tensor_checksums(sd0)
        out["sd_checksum_keys"] = np.array(list(cs.keys()))
        out["sd_checksum_vals"] = np.array(list(cs.values()), dtype=np.float64)

    # ---- x -> z direction
    model.zero_grad()
    z, logp, h_out, eps = model.forward(x, y, h_in, return_eps=True)
    loss = loss_forward(logp, y)
    loss.backward()
    out["fwd.z"] = z.detach().numpy()
    out["fwd.logp"] = logp.detach().numpy()
    out["fwd.loss"] = np.array(loss.item())
    out.update(states_to_np(h_out, "fwd.h_out."))
    for i, e in enumerate(eps):
        out["fwd.eps.%d" % i] = e.detach().numpy()
    gf = grads_to_np(model, "fwd.grad.")
    if full_grads:
        out.update(gf)
    else:
        out["fwd.gradnorm_keys"] = np.array([k[len("fwd.grad."):] for k in gf])
        out["fwd.gradnorm_vals"] = np.array([float(np.sqrt((v.astype(np.float64) ** 2).sum())) for v in gf.values()])
        out["fwd.gradsample"] = np.concatenate([v.reshape(-1)[::GRAD_SAMPLE_STRIDE] for v in gf.values()])

    # BatchNorm buffers advanced during the forward call: restore so both directions start equal
    model.load_state_dict(sd0)

    # ---- z -> y direction (generative; what main.py trains through)
    model.zero_grad()
    invalidate_w_cache(model)
    eps_d = [e.detach() for e in eps]
    y_rec, logdet, h_out2 = model.reconstruct(x, h_in, eps_d)
    loss_r = loss_reverse(y_rec, logdet)
    loss_r.backward()
    out["rev.y"] = y_rec.detach().numpy()
    out["rev.logdet"] = logdet.detach().numpy()
    out["rev.loss"] = np.array(loss_r.item())
    out.update(states_to_np(h_out2, "rev.h_out."))
    gr = grads_to_np(model, "rev.grad.")
    if full_grads:
        out.update(gr)
    else:
        out["rev.gradnorm_keys"] = np.array([k[len("rev.grad."):] for k in gr])
        out["rev.gradnorm_vals"] = np.array([float(np.sqrt((v.astype(np.float64) ** 2).sum())) for v in gr.values()])
        out["rev.gradsample"] = np.concatenate([v.reshape(-1)[::GRAD_SAMPLE_STRIDE] for v in gr.values()])
    out["roundtrip_maxabs"] = np.array(float((y_rec.detach() - y).abs().max()))
    if compact:
        out = _compact(out)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "roundtrip", out["roundtrip_maxabs"], "logp", out["fwd.logp"][:2], "loss_r", loss_r.item())


def train_case(ref, cfg, B, fname, n_opt=3, tback=3):
    """A16: the trainer's inner step (trainFlowParallel.py:256-297) driven through reconstruct()
    with injected eps, captured for n_opt optimizer steps of tback time-steps each."""
    seed_all(12345)
    model = ref.TMGlow(**build_kwargs(cfg))
    perturb_(model, 7, 0.05, 0.1, 0.05)
    model.train()
    Hin, Win = cfg["_in_hw"]
    up = cfg["_up"]
    H, W = Hin * up, Win * up
    L = len(cfg["glow_blocks"])
    out = sd_to_np(model.state_dict(), "sd.")
    g = torch.Generator().manual_seed(4242)
    xs = torch.randn(n_opt, tback, B, cfg["in_features"], Hin, Win, generator=g)
    out["xs"] = xs.numpy()
    # eps per (opt step, time-step, level..., top)
    eps_shapes = []
    c = cfg["out_features"]
    for i in range(L):
        c = c * 2
        eps_shapes.append((B, c, H // 2 ** (i + 1), W // 2 ** (i + 1)))
    eps_shapes.append((B, cfg["out_features"] * 2 ** L, H // 2 ** L, W // 2 ** L))
    eps_all = [[[0.7 * torch.randn(s, generator=g) for s in eps_shapes] for _ in range(tback)] for _ in range(n_opt)]
    for a in range(n_opt):
        for t in range(tback):
            for i, e in enumerate(eps_all[a][t]):
                out["eps.%d.%d.%d" % (a, t, i)] = e.numpy()
    seeds = torch.arange(B) + 11
    out["seeds"] = seeds.numpy()
    a_key = model.initLSTMStates(seeds, [H, W])
    a0 = [(h.clone(), c_.clone()) for h, c_ in a_key]
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
    max_grad_norm = 0.1
    out["max_grad_norm"] = np.array(max_grad_norm)
    log_s_key = "glow.flow_blocks.0.revlayers.affine_layer2.conv.log_s"
    for a in range(n_opt):
        opt.zero_grad()
        loss = 0.0
        for t in range(tback):
            invalidate_w_cache(model)
            y_t, logp_t, a0 = model.reconstruct(xs[a, t], a0, eps_all[a][t])
            loss = loss + loss_reverse(y_t, logp_t)
            out["step%d.t%d.y" % (a, t)] = y_t.detach().numpy()
            out["step%d.t%d.logp" % (a, t)] = logp_t.detach().numpy()
        out.update(states_to_np(a0, "step%d.h_out." % a))
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
        opt.step()
        out["step%d.loss" % a] = np.array(loss.item())
        out["step%d.gradnorm" % a] = np.array(float(gn))
        out["step%d.log_s" % a] = dict(model.named_parameters())[log_s_key].detach().numpy().copy()
        a0 = [(0.5 * h.detach() + 0.5 * hk, 0.5 * c_.detach() + 0.5 * ck) for (h, c_), (hk, ck) in zip(a0, a_key)]
        print("train step", a, "loss", loss.item(), "gn", float(gn))
    out["log_s_key"] = np.array(log_s_key)
    np.savez_compressed(os.path.join(HERE, fname), **out)


def module_cases(ref, fname):
    """Per-module known answers with larger perturbations (block round-trips stay ~1e-6)."""
    sys.path.insert(0, REF)
    from nn.modules.flowLSTMBlock import LSTMFLowBlock
    from nn.modules.glowConv import InvertibleConv1x1
    from nn.modules.flowUtils import Squeeze
    out = {}
    seed_all(2024)
    # one whole flow level: C_in 2 -> squeeze 8, cond 3, rec 5, K = 3
    blk = LSTMFLowBlock(2, 3, 5, 3, LUdecompose=True, train_sampling=True, do_split=True, squeeze_type=0)
    perturb_(blk, 5, 0.1, 0.2, 0.1)
    out.update(sd_to_np(blk.state_dict(), "level.sd."))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 2, 8, 12, generator=g, requires_grad=True)
    cond = torch.randn(2, 3, 4, 6, generator=g, requires_grad=True)
    hs = (torch.rand(2, 5, 4, 6, generator=g) * 2 - 1).requires_grad_(True)
    cs = torch.randn(2, 5, 4, 6, generator=g).requires_grad_(True)
    z, ld, st, eps = blk.forward(x, cond, (hs, cs), return_eps=True)
    wz = torch.randn(z.shape, generator=g)
    wh = torch.randn(st[0].shape, generator=g)
    wc = torch.randn(st[1].shape, generator=g)
    loss = (z * wz).sum() + ld.sum() * 0.01 + (st[0] * wh).sum() + (st[1] * wc).sum()
    loss.backward()
    out.update({"level.x": x.detach().numpy(), "level.cond": cond.detach().numpy(), "level.h": hs.detach().numpy(),
                "level.c": cs.detach().numpy(), "level.fwd.z": z.detach().numpy(), "level.fwd.logdet": ld.detach().numpy(),
                "level.fwd.h_out": st[0].detach().numpy(), "level.fwd.c_out": st[1].detach().numpy(),
                "level.fwd.eps": eps.detach().numpy(), "level.wz": wz.numpy(), "level.wh": wh.numpy(), "level.wc": wc.numpy(),
                "level.fwd.dx": x.grad.numpy().copy(), "level.fwd.dcond": cond.grad.numpy().copy(),
                "level.fwd.dh": hs.grad.numpy().copy(), "level.fwd.dc": cs.grad.numpy().copy()})
    out.update(grads_to_np(blk, "level.fwd.grad."))
    # reverse of the same level
    blk.zero_grad()
    for t in (x, cond, hs, cs):
        t.grad = None
    invalidate_w_cache(blk)
    zin = z.detach().clone().requires_grad_(True)
    xr, ldr, str_ = blk.reverse(zin, cond, (hs, cs), eps=eps.detach())
    wx = torch.randn(xr.shape, generator=g)
    lossr = (xr * wx).sum() + ldr.sum() * 0.01 + (str_[0] * wh).sum() + (str_[1] * wc).sum()
    lossr.backward()
    out.update({"level.rev.x": xr.detach().numpy(), "level.rev.logdet": ldr.detach().numpy(), "level.wx": wx.numpy(),
                "level.rev.h_out": str_[0].detach().numpy(), "level.rev.c_out": str_[1].detach().numpy(),
                "level.rev.dz": zin.grad.numpy().copy(), "level.rev.dcond": cond.grad.numpy().copy(),
                "level.rev.dh": hs.grad.numpy().copy(), "level.rev.dc": cs.grad.numpy().copy()})
    out.update(grads_to_np(blk, "level.rev.grad."))
    print("level roundtrip", float((xr.detach() - x.detach()).abs().max()))

    # plain (non-LU) invertible 1x1, both train_sampling settings
    for ts in (True, False):
        seed_all(31)
        m = InvertibleConv1x1(6, train_sampling=ts)
        xx = torch.randn(2, 6, 5, 7, generator=g)
        yf, ldf = m.forward(xx)
        yr, ldr2 = m.reverse(xx)
        tag = "plain1x1.ts%d." % int(ts)
        out.update({tag + "weight": m.weight.detach().numpy(), tag + "x": xx.numpy(), tag + "fwd.y": yf.detach().numpy(),
                    tag + "fwd.logdet": np.array(ldf.item()), tag + "rev.y": yr.detach().numpy(),
                    tag + "rev.logdet": np.array(ldr2.item())})
    # Glow-style Squeeze (unused by TMGlow)
    sq = Squeeze(2)
    xx = torch.randn(2, 3, 4, 6, generator=g)
    out["squeeze.x"] = xx.numpy()
    out["squeeze.y"] = sq.forward(xx).numpy()
    np.savez_compressed(os.path.join(HERE, fname), **out)


def init_checksum_case(ref, cfg, fname):
    """Seeded construction only: per-tensor checksums of the reference's initial state_dict, so the
    product model's constructor can be checked to consume numpy/torch RNG identically."""
    seed_all(12345)
    model = ref.TMGlow(**build_kwargs(cfg))
    cs = tensor_checksums(model.state_dict())
    np.savez_compressed(os.path.join(HERE, fname), keys=np.array(list(cs.keys())),
                        vals=np.array(list(cs.values()), dtype=np.float64),
                        shapes=np.array([str(tuple(v.shape)) for v in model.state_dict().values()]))
    print(fname, len(cs), "tensors")


def loss_case(fname):
    """F1: the physics-constrained reverse-KL loss of the trainer (TMGLowLoss) with its input gradients."""
    from types import SimpleNamespace
    import nn.trainFlowParallel as tfp
    out = {}
    for tag, (B, T, Hh, Ww, amp, dx) in {"small": (2, 3, 10, 12, 1.0, 2.0 / 64), "clamped": (2, 4, 9, 8, 3.0, 5.0 / 64)}.items():
        g = torch.Generator().manual_seed(11)
        std = torch.tensor([1.3, 0.7, 2.1])
        mu = torch.tensor([0.2, -0.1, 0.4])
        model = SimpleNamespace(module=SimpleNamespace(out_std=std, out_mu=mu))
        args = SimpleNamespace(beta=200.0, dx=dx, dy=dx * 1.25)
        crit = tfp.TMGLowLoss(args, model)
        y = (amp * 0.05 * torch.randn(B, T, 3, Hh, Ww, generator=g)).requires_grad_(True)
        logp = (100 * torch.randn(B, T, generator=g)).requires_grad_(True)
        tgt = amp * 0.05 * torch.randn(B, T, 3, Hh, Ww, generator=g)
        tmean = tgt.mean(1)
        trms = torch.sqrt(torch.mean((tgt - tmean.unsqueeze(1)) ** 2, dim=1))
        loss = crit(y, logp, tgt, tmean, trms)
        loss.backward()
        out.update({tag + ".y": y.detach().numpy().copy(), tag + ".logp": logp.detach().numpy().copy(), tag + ".target": tgt.numpy(),
                    tag + ".tmean": tmean.numpy(), tag + ".trms": trms.numpy(), tag + ".std": std.numpy(), tag + ".mu": mu.numpy(),
                    tag + ".cfg": np.array([args.beta, args.dx, args.dy]), tag + ".loss": np.array(loss.item()),
                    tag + ".dy": y.grad.numpy().copy(), tag + ".dlogp": logp.grad.numpy().copy()})
        with torch.no_grad():
            hat = std.view(1, 3, 1, 1) * y.detach().reshape(-1, 3, Hh, Ww) + mu.view(1, 3, 1, 1)
            out[tag + ".pstar"] = crit.phys.calcPressurePoisson(hat[:, :2], hat[:, 2:]).numpy()
            out[tag + ".ustar"] = crit.phys.calcDivergence(hat[:, :2]).numpy()
        print("loss case", tag, loss.item(), "clamped frac", float((out[tag + ".pstar"].__abs__() >= 1).mean()))
    np.savez_compressed(os.path.join(HERE, fname), **out)


def phys_fields_case(fname):
    """F1, API completeness: PhysConstrainedLES.calcDivergence / calcPressurePoisson with the 3x3 and 5x5 stencils of pc/grad1Filter.py /
    pc/grad2Filter.py (every combination) and scale = True / False, on a field whose residuals are partly clamped."""
    import pc.physicsConstrained as rpc
    out = {}
    g = torch.Generator().manual_seed(23)
    B, Hh, Ww = 2, 11, 13
    dx, dy, rho = 0.05, 0.0625, 1.3
    u = 0.02 * torch.randn(B, 2, Hh, Ww, generator=g)
    pr = 0.01 * torch.randn(B, 1, Hh, Ww, generator=g)
    out.update({"u": u.numpy(), "p": pr.numpy(), "cfg": np.array([dx, dy, rho])})
    for k1 in (3, 5):
        for k2 in (3, 5):
            phys = rpc.PhysConstrainedLES(dx, dy, rho=rho, grad_kernels=[k1, k2])
            for scale in (True, False):
                tag = "k%d%d.%s" % (k1, k2, "scaled" if scale else "raw")
                # (amplitudes that leave part of each residual field inside the clamp, part outside)
                au, ap = (50.0, 45.0) if scale else (6.0, 0.45)
                with torch.no_grad():
                    out[tag + ".ustar"] = phys.calcDivergence(au * u, scale=scale).numpy()
                    out[tag + ".pstar"] = phys.calcPressurePoisson(ap * u, ap * pr, scale=scale).numpy()
                out[tag + ".amp"] = np.array([au, ap])
                print("phys fields", tag, "clamped frac", float((np.abs(out[tag + ".pstar"]) >= 1).mean()), float((np.abs(out[tag + ".ustar"]) >= 1).mean()))
    try:
        rpc.PhysConstrainedLES(dx, dy, grad_kernels=[7, 3])
        out["k7_raises"] = np.array(0)
    except ValueError:
        out["k7_raises"] = np.array(1)
    np.savez_compressed(os.path.join(HERE, fname), **out)


def loader_case(fname):
    """Row F4: the reference's BackwardStepLoader on synthetic files (tests/common.py writes them again at test time)."""
    import tempfile
    from common import write_synthetic_step_data, LOADER_U0
    sys.path.insert(0, REF)
    from utils.dataLoader import BackwardStepLoader

    class _Quiet(object):
        def log(self, *a, **k): pass
        warning = error = info = log

    out = {}
    with tempfile.TemporaryDirectory() as d:
        write_synthetic_step_data(d)
        seed_all(777)
        ld = BackwardStepLoader(d, d, shuffle=False, log=_Quiet())
        tr = ld.createTrainingLoader([0, 1], LOADER_U0, tSplit=2, inUpscale=2, batch_size=3, tar_noise_std=0)
        xs, ys, ss = zip(*[b for b in tr])
        out["train.x"], out["train.y"], out["train.seed"] = torch.cat(xs).numpy(), torch.cat(ys).numpy(), torch.cat(ss).numpy()
        out["train.nbatch"] = np.array([len(tr)])
        for k in ("input_mean", "input_std", "output_mean", "output_std"):
            out["norm." + k] = getattr(ld, k).numpy().copy()
        te = ld.createTestingLoader([1], LOADER_U0, inUpscale=2, batch_size=8)
        xs, ys, us = zip(*[b for b in te])
        out["test.x"], out["test.y"], out["test.u0"] = torch.cat(xs).numpy(), torch.cat(ys).numpy(), torch.cat(us).numpy()
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print("wrote", fname, {k: v.shape for k, v in out.items()})


def cylinder_loader_case(fname):
    """Row F4: the reference's CylinderArrayLoader (training: tSplit 2, drop_last; testing) and DataLoaderAuto's case selection
    (`setupCylinderLoaders`) on synthetic files that tests/common.py writes again at test time."""
    import tempfile
    from types import SimpleNamespace
    from common import write_synthetic_cylinder_data
    sys.path.insert(0, REF)
    from utils.dataLoader import CylinderArrayLoader, DataLoaderAuto

    class _Quiet(object):
        def log(self, *a, **k): pass
        warning = error = info = log

    out = {}
    with tempfile.TemporaryDirectory() as d:
        write_synthetic_cylinder_data(d, cases=(0, 1, 2))
        seed_all(778)
        ld = CylinderArrayLoader(d, d, shuffle=False, log=_Quiet())
        tr = ld.createTrainingLoader([0, 2, 1], tSplit=2, inUpscale=1, batch_size=4, tar_noise_std=0)
        xs, ys, ss = zip(*[b for b in tr])
        out["train.x"], out["train.y"], out["train.seed"] = torch.cat(xs).numpy(), torch.cat(ys).numpy(), torch.cat(ss).numpy()
        out["train.nbatch"] = np.array([len(tr)])
        for k in ("input_mean", "input_std", "output_mean", "output_std"):
            out["norm." + k] = getattr(ld, k).numpy().copy()
        te = ld.createTestingLoader([1, 2], batch_size=8)
        xs, ys, us = zip(*[b for b in te])
        out["test.x"], out["test.y"], out["test.u0"] = torch.cat(xs).numpy(), torch.cat(ys).numpy(), torch.cat(us).numpy()
    # DataLoaderAuto: which cases it reads and what it hands back (main.py:86)
    with tempfile.TemporaryDirectory() as d:
        write_synthetic_cylinder_data(d, cases=(0, 47, 95, 96, 97), seed=98)
        seed_all(779)
        args = SimpleNamespace(exp_type='cylinder-array', ntrain=3, ntest=2, training_data_dir=d, testing_data_dir=d, epoch_start=0,
                               batch_size=2, test_batch_size=2, noise_std=0.0, seed=1)
        holder = SimpleNamespace(module=torch.nn.Linear(1, 1))   # transferNormalizingParams assigns the four buffers by attribute
        auto, tr, te = DataLoaderAuto.init_data_loaders(args, holder, _Quiet())
        out["auto.train.n"] = np.array([len(tr.dataset), len(tr)])
        out["auto.test.n"] = np.array([len(te.dataset), len(te)])
        out["auto.train.x_all"] = tr.dataset.inputs.numpy().copy()
        out["auto.train.y_all"] = tr.dataset.targets.numpy().copy()
        out["auto.train.seed_all"] = tr.dataset.lstm_seeds.numpy().copy()
        out["auto.test.x_all"] = te.dataset.tensors[0].numpy().copy()
        out["auto.test.y_all"] = te.dataset.tensors[1].numpy().copy()
        for k in ("in_mu", "in_std", "out_mu", "out_std"):
            out["auto.buf." + k] = getattr(holder.module, k).numpy().copy()
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print("wrote", fname, {k: v.shape for k, v in out.items()})


def workspace_case(fname_zip):
    """Row F3: a workspace written by the reference's saveWorkspace for the tiny model + Adam (one step taken)."""
    import shutil
    import tempfile
    from types import SimpleNamespace
    ref = import_reference()
    from utils.utils import saveWorkspace, loadWorkspace
    seed_all(12345)
    model = ref.TMGlow(**build_kwargs(CFG_TINY))
    perturb_(model, 7, 0.05, 0.1, 0.05)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
    for p in model.parameters():
        p.grad = 0.01 * torch.ones_like(p)
    opt.step()
    with tempfile.TemporaryDirectory() as d:
        args = SimpleNamespace(ckpt_dir=d, device="cpu", beta=200.0, lr=1e-3, epochs=3, epoch_start=0, note="golden")
        saveWorkspace(args, model, opt, file_id=7)
        shutil.copy(os.path.join(d, "nsWorkspace7.zip"), os.path.join(HERE, fname_zip))
        # the other direction: a workspace written by the new code loads with the reference's loader
        sys.path.insert(0, os.path.join(HERE, "..", "..", "deep-turbulence_amd"))
        import importlib.util
        spec = importlib.util.spec_from_file_location("amd_utils", os.path.join(HERE, "..", "..", "deep-turbulence_amd", "utils", "utils.py"))
        mine = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mine)
        args2 = SimpleNamespace(ckpt_dir=d, device="cpu", beta=1.0, lr=5e-4, epochs=9, epoch_start=2, note="mine")
        mine.saveWorkspace(args2, model, opt, file_id=8)
        a3, sd, osd = loadWorkspace(SimpleNamespace(epochs=1, epoch_start=0), d, file_id=8)
        assert a3.note == "mine" and a3.epochs == 1 and all(torch.equal(sd[k], v) for k, v in model.state_dict().items())
        assert len(osd["state"]) == len(opt.state_dict()["state"])
    np.savez_compressed(os.path.join(HERE, fname_zip.replace(".zip", "_check.npz")), **{"sd." + k: np.array(v) for k, v in tensor_checksums(model.state_dict()).items()})
    print("wrote", fname_zip)


if __name__ == "__main__":
    ref = import_reference()
    if "loader" in sys.argv:
        loader_case("loader_case.npz")
        sys.exit(0)
    if "cylinder" in sys.argv:
        cylinder_loader_case("cylinder_loader_case.npz")
        sys.exit(0)
    if "tiny5" in sys.argv:
        torch.set_num_threads(8)
        model_case(ref, CFG_TINY5, 2, "tiny5_model.npz", store_weights=False, scales=(0.004, 0.02, 0.004), full_grads=False, compact=True)
        sys.exit(0)
    if "cfg1" in sys.argv:
        torch.set_num_threads(8)
        model_case(ref, CFG1, 2, "cfg1_model.npz", store_weights=False, scales=(0.004, 0.02, 0.004), full_grads=False, compact=True)
        sys.exit(0)
    if "workspace" in sys.argv:
        workspace_case("ref_workspace7.zip")
        sys.exit(0)
    if "loss" in sys.argv:
        loss_case("phys_loss.npz")
        sys.exit(0)
    if "physfields" in sys.argv:
        phys_fields_case("phys_fields.npz")
        sys.exit(0)
    torch.set_num_threads(8)
    model_case(ref, CFG_TINY, 2, "tiny_model.npz")
    model_case(ref, CFG_TINY3, 3, "tiny3_model.npz", scales=(0.03, 0.05, 0.03))
    model_case(ref, CFG_TINY5, 2, "tiny5_model.npz", store_weights=False, scales=(0.004, 0.02, 0.004), full_grads=False, compact=True)
    train_case(ref, CFG_TINY, 2, "tiny_train.npz")
    module_cases(ref, "modules.npz")
    init_checksum_case(ref, CFG1, "cfg1_init_checksums.npz")
    model_case(ref, CFG1, 2, "cfg1_model.npz", store_weights=False, scales=(0.004, 0.02, 0.004), full_grads=False,
               compact=True)
