"""GPU parity of the assembled TM-Glow HIP path against the golden fixtures (outputs of the reference
itself) and against the CPU oracle, through the drop-in module API."""
import os

import numpy as np
import pytest
import torch

import common as C

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(cfg, sd):
    from nn.tmGlow import TMGlow
    m = TMGlow(**C.build_kwargs(cfg))
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).train()


def _grads(m):
    return {k: p.grad for k, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("name,cfg", [("tiny_model.npz", C.CFG_TINY), ("tiny3_model.npz", C.CFG_TINY3)])
def test_forward_reverse_and_grads_match_reference(name, cfg):
    d = C.load_npz(name)
    L = len(cfg["glow_blocks"])
    sd = {k: torch.from_numpy(v) for k, v in C.sub(d, "sd.").items()}
    m = _model(cfg, sd)
    x, y = torch.from_numpy(d["x"]).to(DEV), torch.from_numpy(d["y"]).to(DEV)
    h_in = C.states_from(d, "h_in.", L, DEV)
    z, logp, h_out, eps = m.forward(x, y, h_in, return_eps=True)
    assert z.shape == tuple(d["fwd.z"].shape)
    C.assert_field(z, d["fwd.z"], "z")
    C.assert_logdet(logp, d["fwd.logp"], "logp")
    for i in range(L):
        C.assert_field(h_out[i][0], d["fwd.h_out.%d.h" % i], "h_out", atol=C.STATE_ATOL)
        C.assert_field(h_out[i][1], d["fwd.h_out.%d.c" % i], "c_out", atol=C.STATE_ATOL)
    for i in range(L + 1):
        C.assert_field(eps[i], d["fwd.eps.%d" % i], "eps%d" % i)
    C.loss_forward(logp, y).backward()
    C.assert_grads(_grads(m), C.sub(d, "fwd.grad."), "fwd grads")

    m = _model(cfg, sd)
    eps_in = [torch.from_numpy(d["fwd.eps.%d" % i]).to(DEV) for i in range(L + 1)]
    yr, logdet, h_out2 = m.reconstruct(x, h_in, eps_in)
    C.assert_field(yr, d["rev.y"], "y_rec")
    C.assert_logdet(logdet, d["rev.logdet"], "logdet")
    for i in range(L):
        C.assert_field(h_out2[i][0], d["rev.h_out.%d.h" % i], "h_out", atol=C.STATE_ATOL)
        C.assert_field(h_out2[i][1], d["rev.h_out.%d.c" % i], "c_out", atol=C.STATE_ATOL)
    C.loss_reverse(yr, logdet).backward()
    C.assert_grads(_grads(m), C.sub(d, "rev.grad."), "rev grads")
    # dead parameters of the reference stay dead
    assert all(p.grad is None for k, p in m.named_parameters() if ".norm2." in k)
    # invertibility (the reference's own self-test property)
    assert float((yr.detach() - y).abs().max()) < 5e-4


@pytest.mark.parametrize("name,cfg", [("cfg1_model.npz", C.CFG1), ("tiny5_model.npz", C.CFG_TINY5)])
def test_seeded_models_match_reference(name, cfg):
    """BASELINE configs[0] widths (cfg1: L=3, K=16, Cc=32, R=64) and a five-level model (depth of configs[4], deepest level on
    2x2 maps, encoder down to one pixel): weights re-created from the seeds (checksums verified against the reference's),
    then outputs, states, latents, per-tensor gradient norms and every 13th entry of every gradient against the values
    recorded from the reference."""
    from nn.tmGlow import TMGlow
    d = C.load_npz(name)
    L = len(cfg["glow_blocks"])
    m = C.seeded_state_dict(TMGlow, cfg, d)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(DEV).train()
    x, y = torch.from_numpy(d["x"]).to(DEV), torch.from_numpy(d["y"]).to(DEV)
    B = x.shape[0]
    h_in = m.initLSTMStates(torch.arange(B), [y.shape[2], y.shape[3]])
    z, logp, h_out, eps = m.forward(x, y, h_in, return_eps=True)
    C.assert_field(z, d["fwd.z"], "z")
    C.assert_logdet(logp, d["fwd.logp"], "logp")
    for i in range(L):
        C.assert_compact_field(d, "fwd.h_out.%d.h" % i, h_out[i][0], "h_out", atol=C.STATE_ATOL)
        C.assert_compact_field(d, "fwd.h_out.%d.c" % i, h_out[i][1], "c_out", atol=C.STATE_ATOL)
    for i in range(L + 1):
        C.assert_compact_field(d, "fwd.eps.%d" % i, eps[i], "eps%d" % i)
    # fp64 evaluation of the same case by the CPU oracle: measures the fp32 noise the reference-recorded gradients carry
    import os
    import sys
    sys.path.insert(0, os.path.join(C.ROOT, "oracle"))
    import tmglow_oracle as O
    r64 = _oracle_pass(O, sd0, cfg, x.cpu(), y.cpu(), torch.arange(B), torch.float64)
    C.loss_forward(logp, y).backward()
    C.assert_compact_grads(d, "fwd.", _grads(m), name + " fwd grads", truth=r64["gf"])
    m.load_state_dict(sd0)
    m.zero_grad()
    yr, logdet, h_out2 = m.reconstruct(x, h_in, [e.detach() for e in eps])
    C.assert_field(yr, d["rev.y"], "y_rec")
    C.assert_logdet(logdet, d["rev.logdet"], "logdet")
    for i in range(L):
        C.assert_compact_field(d, "rev.h_out.%d.h" % i, h_out2[i][0], "h_out", atol=C.STATE_ATOL)
    C.loss_reverse(yr, logdet).backward()
    C.assert_compact_grads(d, "rev.", _grads(m), name + " rev grads", truth=r64["gr"])


# cfg5 at a reduced field: the widths and the five levels of BASELINE configs[4] on a 64x64 output (level 5 works on 2x2 maps
# with 256 channels); the full 512x512 field is covered by test_cfg5_full_size_properties
CFG5_REDUCED = dict(C.CFG5, _in_hw=(32, 32))
YARDSTICK = 3.0        # fields / log-dets / states may deviate from the fp64 oracle by 3x what the reference's own fp32 arithmetic does
GRAD_YARDSTICK = 1.5   # gradients: bound = max(stated tolerance, 1.5 x the fp32 oracle's own error against fp64)
# Why gradients need a yardstick at all (profiles/r3_kink_scan_M.json, tools/kink_scan.py): at the metric configuration, batch 1, the
# generative pass evaluates 48 M ReLUs, and for EVERY input seed 14-63 of them have an fp64 pre-activation within 1e-5 of zero that the
# reference's own fp32 arithmetic puts on the other side; whether a given fp32 evaluation flips a given one depends on the last bit of
# the encoder's BatchNorm sums.  One flipped element of a deep 16x16 map moves the deepest level's weight gradients by 1e-3 of their
# scale: the SAME HIP build lands at 6.5e-5 or at 2.9e-4 global rel-L2 from fp64 on config M (profiles/r3_parity_report_M.json, the
# variant run twice), the second value within 4e-5 of the fp32 oracle, which sits at 2.9e-4 itself.  The arithmetic noise proper of
# the HIP path is the smaller number; the flips are a property of the case, shared with the reference.
# Which elements flip differs between two fp32 evaluations.  Round 3 let any 2 of the ~900 tensors exceed the per-tensor bound 3x;
# round 4 ties the allowance to evidence: the fp64 oracle pass runs under common.KinkProbe, which records every ReLU pre-activation
# within 1e-5 (relative to its tensor's scale) of zero and measures - one extra backward pass through the retained graph - how far
# flipping them moves EACH parameter gradient.  A tensor may exceed the per-tensor bound by its own measured allowance only; a tensor
# that no near-kink ReLU feeds has none, and the global bound has none.  Comparisons of two HIP paths that share the arithmetic order
# (test_level_kernels_across_field_and_batch_sizes) get no allowance at all.


def _maxabs(a, b):
    return float((torch.as_tensor(a).detach().cpu().double() - torch.as_tensor(b).detach().cpu().double()).abs().max())


def _grad_err(got, ref):
    """(global relative L2, worst per-tensor relative max) of name -> tensor dicts against `ref`."""
    num = den = worst = 0.0
    for k, r in ref.items():
        g = torch.as_tensor(got[k]).detach().cpu().double().reshape(-1)
        r = torch.as_tensor(r).detach().cpu().double().reshape(-1)
        num += float(((g - r) ** 2).sum())
        den += float((r ** 2).sum())
        if float(r.abs().max()) > 0:
            worst = max(worst, float((g - r).abs().max()) / float(r.abs().max()))
    return (num / max(den, 1e-300)) ** 0.5, worst


def _oracle_pass(O, sd, cfg, x, y, seeds, dtype, eps=None, probe=False):
    """forward(+grads) and reconstruct(+grads) of the CPU oracle in `dtype`; eps: latents for the generative direction
    (default: the forward pass's own).  probe: also measure the per-tensor ReLU-kink allowances of both directions
    (common.KinkProbe; keys kf / kr, and the number of near-kink elements nkf / nkr)."""
    H_, W_ = y.shape[2], y.shape[3]
    P = O.params_from_state_dict(sd, dtype=dtype)
    st = [(h.to(dtype), c.to(dtype)) for h, c in O.init_lstm_states(cfg, seeds, [H_, W_])]
    xx, yy = x.to(dtype), y.to(dtype)
    import contextlib
    kp = C.KinkProbe() if probe else contextlib.nullcontext()
    with kp:
        z, lp, ho, eo = O.tmglow_forward(P, cfg, xx, yy, st, return_eps=True, training=True)
        C.loss_forward(lp, yy).backward(retain_graph=probe)
    gf = {k: v.grad.clone() for k, v in O.trainable(P).items() if v.grad is not None}
    nkf = kp.n_elements if probe else 0
    kf = kp.allowances(O.trainable(P), gf) if probe else None
    P = O.params_from_state_dict(sd, dtype=dtype)
    # identical inputs for every evaluation: the latents are ROUNDED to fp32 (what the fp32 paths can be given) before the fp64 pass too
    e_in = [t.detach().float().to(dtype) for t in (eps if eps is not None else eo)]
    kp = C.KinkProbe() if probe else contextlib.nullcontext()
    with kp:
        yr, ld, ho2 = O.tmglow_reconstruct(P, cfg, xx, st, e_in, training=True)
        C.loss_reverse(yr, ld).backward(retain_graph=probe)
    gr = {k: v.grad.clone() for k, v in O.trainable(P).items() if v.grad is not None}
    nkr = kp.n_elements if probe else 0
    kr = kp.allowances(O.trainable(P), gr) if probe else None
    det = lambda t: t.detach()  # noqa: E731
    return dict(z=det(z), lp=det(lp), h=[(det(a), det(b)) for a, b in ho], eps=[det(t) for t in eo], gf=gf, y=det(yr), ld=det(ld),
                h2=[(det(a), det(b)) for a, b in ho2], gr=gr, kf=kf, kr=kr, nkf=nkf, nkr=nkr)


@pytest.mark.parametrize("name,cfg,B", [("cfg5-64x64", CFG5_REDUCED, 2)] + (
    [("cfg2", C.CFG2, 2), ("cfg3", C.CFG3, 1), ("M", C.CFG_M, 1)] if os.environ.get("TMG_TEST_SMALL_BATCH_CONFIGS") else []))
def test_baseline_configs_match_fp64_oracle(name, cfg, B):
    """BASELINE configs[4]'s five-level network (reduced field) at the default widths, small batch: forward, reconstruct and all
    gradients of the HIP path against the CPU oracle evaluated in FP64 on the same seeded weights.  configs[1] (64x64x3 -> 128x128x3,
    L=3), configs[2] (64x128x4 -> 128x256x4, L=4) and the metric configuration M / configs[3] are held to the same oracle at their STATED
    batches, both directions, by test_stated_batches_match_oracle_with_gradients; their small-batch cases of rounds 2-4 (85 + 79 + 78 s
    of host-side fp64 arithmetic in a suite the driver gives 20 minutes) run with TMG_TEST_SMALL_BATCH_CONFIGS=1, and
    `tools/parity_report.py --config ...` is the long form.

    Tolerance rule.  SURVEY 8-C states fp32 tolerances as 10x the reference's fp32-vs-fp64 noise measured on config 1; that
    noise grows with depth and field size (48-80 coupling layers here), so every bound below is
    max(stated tolerance, YARDSTICK x the error of the fp32 ORACLE against the same fp64 values), the fp32 oracle being the
    reference's own arithmetic - measured here, in this test, not hard-coded.  The measured floors and the HIP errors are
    printed (pytest -s) and written to gpurun_out/parity_<name>.json."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(C.ROOT, "oracle"))
    import tmglow_oracle as O
    from nn.tmGlow import TMGlow
    import contextlib
    import io
    C.seed_all(12345)
    with contextlib.redirect_stdout(io.StringIO()):
        m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 7, *C.perturb_scales(cfg))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(DEV).train()
    h, w = cfg["_in_hw"]
    H_, W_ = h * cfg["_up"], w * cfg["_up"]
    # Inputs must stay clear of ReLU kinks, as in any gradient check: with input seed 31 the five-level case has ONE pre-activation
    # of the level-2 ConvLSTM out conv at +2.2e-7 (tensor scale 1), which fp32 evaluation orders that are all within 1e-6 of the fp64
    # forward (direct contraction, Winograd contraction, either 1x1 kernel) put on different sides of zero; that element happens to
    # be the largest entry of its upstream gradient, so the flip moves two weight-gradient tensors by 2-10 % while the forward pass
    # and every other gradient agree to 1e-6 (measured element by element when the Winograd kernels came in).  Seed 37 has no such
    # element for any of the kernel selections (TMG_NO_WINOGRAD / TMG_NO_MIX32 on or off).
    g = torch.Generator().manual_seed(int(os.environ.get("TMG_TEST_INPUT_SEED", 37 if name == "cfg5-64x64" else 31)))
    x = torch.randn(B, cfg["in_features"], h, w, generator=g)
    y = torch.randn(B, cfg["out_features"], H_, W_, generator=g)
    seeds = torch.arange(B) + 3
    r64 = _oracle_pass(O, sd, cfg, x, y, seeds, torch.float64, probe=True)
    r32 = _oracle_pass(O, sd, cfg, x, y, seeds, torch.float32, eps=r64["eps"])
    # ---- HIP
    st = m.initLSTMStates(seeds, [H_, W_])
    z, lp, ho, e = m.forward(x.to(DEV), y.to(DEV), st, return_eps=True)
    C.loss_forward(lp, y.to(DEV)).backward()
    gf = {k: v.clone() for k, v in _grads(m).items()}
    m.load_state_dict(sd)
    m.zero_grad()
    yr, ld, ho2 = m.reconstruct(x.to(DEV), st, [t.float().to(DEV) for t in r64["eps"]])
    C.loss_reverse(yr, ld).backward()
    gr = _grads(m)
    rep = {}

    def field(tag, got, key, stated):
        ref = r64[key] if isinstance(key, str) else key[0]
        o32 = r32[key] if isinstance(key, str) else key[1]
        floor = _maxabs(o32, ref)
        err = _maxabs(got, ref)
        rep[tag] = {"hip_vs_fp64": err, "oracle_fp32_vs_fp64": floor, "stated": stated}
        C.assert_field(got, ref, "%s %s" % (name, tag), atol=max(stated, YARDSTICK * floor), rtol=C.FIELD_RTOL)

    def logdet(tag, got, key):
        ref, o32 = r64[key], r32[key]
        floor = float(((o32.double() - ref).abs() / ref.abs().clamp_min(1.0)).max())
        err = float(((got.detach().cpu().double() - ref).abs() / ref.abs().clamp_min(1.0)).max())
        rep[tag] = {"hip_vs_fp64_rel": err, "oracle_fp32_vs_fp64_rel": floor, "stated": C.LOGDET_RTOL}
        C.assert_logdet(got, ref, "%s %s" % (name, tag), rtol=max(C.LOGDET_RTOL, YARDSTICK * floor))

    def grads(tag, got, key):
        fl = _grad_err(r32[key], r64[key])
        er = _grad_err(got, r64[key])
        kink = r64["k" + key[1]]
        rep[tag] = {"hip_vs_fp64": er, "oracle_fp32_vs_fp64": fl, "hip_vs_oracle_fp32": _grad_err(got, r32[key]),
                    "stated": (C.GRAD_GLOBAL_REL_L2, C.GRAD_TENSOR_REL_MAX), "yardstick": GRAD_YARDSTICK,
                    "near_kink_relus": r64["nk" + key[1]], "tensors_with_kink_allowance": len(kink)}
        C.assert_grads(got, r64[key], "%s %s" % (name, tag), global_tol=max(C.GRAD_GLOBAL_REL_L2, GRAD_YARDSTICK * fl[0]),
                       tensor_tol=max(C.GRAD_TENSOR_REL_MAX, GRAD_YARDSTICK * fl[1]), kink=kink, report=rep[tag])

    try:
        field("z", z, "z", C.FIELD_ATOL)
        logdet("logp", lp, "lp")
        for i, ((a, b_), (a64, b64), (a32, b32)) in enumerate(zip(ho, r64["h"], r32["h"])):
            field("h%d" % i, a, (a64, a32), C.STATE_ATOL)
            field("c%d" % i, b_, (b64, b32), C.STATE_ATOL)
        for i, (ee, e64, e32) in enumerate(zip(e, r64["eps"], r32["eps"])):
            field("eps%d" % i, ee, (e64, e32), C.FIELD_ATOL)
        grads("forward grads", gf, "gf")
        field("y", yr, "y", C.FIELD_ATOL)
        logdet("logdet", ld, "ld")
        for i, ((a, b_), (a64, b64), (a32, b32)) in enumerate(zip(ho2, r64["h2"], r32["h2"])):
            field("rev h%d" % i, a, (a64, a32), C.STATE_ATOL)
        grads("reverse grads", gr, "gr")
    finally:
        print("\nparity %s (B=%d): %s" % (name, B, json.dumps(rep, indent=1, default=float)))
        out = os.path.join(C.ROOT, "gpurun_out")
        if os.path.isdir(out):
            with open(os.path.join(out, "parity_%s.json" % name), "w") as f:
                json.dump({"config": name, "batch": B, "yardstick": YARDSTICK, "quantities": rep}, f, indent=1, default=float)


_ORACLE_RESULTS = {}      # (configuration, B, n) -> the CPU oracle's fp64 / fp32 evaluation (shared by the arithmetic variants of one case)


def _loss_indices(B, n):
    """(indices of the n loss samples of a stated-batch case, roll that moves one of them to the last position): scattered over the
    batch - n = 1: the middle sample, 2: first and last, 3: first, middle, last."""
    idx = {1: [B // 2 - 1], 2: [0, B - 1], 3: [0, B // 2 - 1, B - 1]}[n]
    shift = {1: B - B // 2, 2: B // 4 + 1, 3: B // 4 + 1}[n]
    shift = next(s_ for s_ in range(shift, shift + B) if max((i + s_) % B for i in idx) == B - 1)
    return idx, shift


def _extra_indices(B, idx):
    """Up to six more sample positions, spread over the batch, none of them a loss sample."""
    cand = [1, B // 4, B // 3, B // 2, (3 * B) // 4 - 1, B - 2]
    out = []
    for c in cand:
        if 0 <= c < B and c not in idx and c not in out:
            out.append(c)
    return out


def _stated_batch_case(name, cfg, B, n=2, density=True, oracle_key=None):
    """One BASELINE configuration at its STATED batch size on the HIP path - generative direction, loss on `n` samples SCATTERED over
    the batch (first / middle / last: _loss_indices; round 5 took the first n, so that a backward fault tied to the position of a
    tile inside the persistent kernels' ranges could only show at the two ends), backward - against the CPU oracle with gradients.
    Samples are independent through the flow and coupled only by the encoder's BatchNorm batch statistics, so the oracle runs its
    encoder on the FULL batch (< 1 % of the work; its gradient flows through the batch statistics of all B samples) and the flow on
    the `n` samples the loss sees, in fp64 (truth) and in fp32 (the reference's arithmetic: the yardstick).  Compared: y, log-det,
    the recurrent states and ALL parameter gradients; y / log-det / states of up to SIX MORE scattered samples against the fp32
    oracle's forward pass (no gradients: cheap).  The same step is then repeated with the batch rolled (the loss samples at other
    positions of every tensor, one of them the last) and must give the same gradients."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(C.ROOT, "oracle"))
    import tmglow_oracle as O
    from nn.tmGlow import TMGlow
    import contextlib
    import io
    C.seed_all(12345)
    with contextlib.redirect_stdout(io.StringIO()):
        m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 7, *C.perturb_scales(cfg))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(DEV).train()
    L = len(cfg["glow_blocks"])
    h, w = cfg["_in_hw"]
    H_, W_ = h * cfg["_up"], w * cfg["_up"]
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, cfg["in_features"], h, w, generator=g)
    y = torch.randn(B, cfg["out_features"], H_, W_, generator=g)
    seeds = torch.arange(B) + 11
    st = m.initLSTMStates(seeds, [H_, W_])
    with torch.no_grad():    # latents of realistic size and shape: the density direction's own, whole batch
        _, _, _, eps = m.forward(x.to(DEV), y.to(DEV), st, return_eps=True)
        m.load_state_dict(sd)    # (BatchNorm running statistics back to their initial values)
    eps = [e.detach() for e in eps]

    def hip_step(xx, states, ee, sl):
        m.load_state_dict(sd)
        m.zero_grad()
        yr, ld, ho = m.reconstruct(xx, states, ee)
        C.loss_reverse(yr.index_select(0, sl), ld.index_select(0, sl)).backward()
        return yr.detach(), ld.detach(), [(a.detach(), b.detach()) for a, b in ho], {k: v.detach().clone() for k, v in _grads(m).items()}

    idx, shift = _loss_indices(B, n)
    it = torch.tensor(idx)
    sel = lambda t: t.index_select(0, it.to(t.device))  # noqa: E731
    yr, ld, ho, gr = hip_step(x.to(DEV), st, eps, it.to(DEV))
    assert float((yr - y.to(DEV)).abs().max()) < 2e-3          # forward -> reconstruct round trip over the WHOLE batch
    import contextlib
    okey = (oracle_key or name, B, n)
    res = _ORACLE_RESULTS.get(okey, {})
    for dt in ((torch.float64, torch.float32) if not res else ()):
        P = O.params_from_state_dict(sd, dtype=dt)
        kp = C.KinkProbe() if dt == torch.float64 else contextlib.nullcontext()
        with kp:
            z_out, c_out = O.encoder(P, cfg, x.to(dt), True)
            cmean, clsd = sel(z_out).chunk(2, 1)
            clsd = clsd.clamp(-10.0, O.LOG5)
            z = cmean + torch.exp(clsd) * sel(eps[-1].cpu()).to(dt)
            sto = [(a.to(dt), b.to(dt)) for a, b in O.init_lstm_states(cfg, seeds[it], [H_, W_])]
            yo, ldo, hoo = O.decoder_reverse(P, cfg, z, [sel(c) for c in c_out], sto, [sel(e.cpu()).to(dt) for e in eps[:-1]])
            C.loss_reverse(yo, ldo).backward(retain_graph=(dt == torch.float64))
        res[dt] = dict(y=yo.detach(), ld=ldo.detach(), h=[(a.detach(), b.detach()) for a, b in hoo],
                       g={k: v.grad.clone() for k, v in O.trainable(P).items() if v.grad is not None})
        if dt == torch.float64:
            res[dt]["nk"] = kp.n_elements
            res[dt]["kink"] = kp.allowances(O.trainable(P), res[dt]["g"])
        else:
            # forward-only check of more samples, scattered over the batch: the fp32 oracle's y / log-det / states
            ex = _extra_indices(B, idx)
            et = torch.tensor(ex)
            selx = lambda t: t.index_select(0, et)  # noqa: E731
            with torch.no_grad():
                cm, cs = selx(z_out).chunk(2, 1)
                zx = cm + torch.exp(cs.clamp(-10.0, O.LOG5)) * selx(eps[-1].cpu()).to(dt)
                stx = [(a.to(dt), b.to(dt)) for a, b in O.init_lstm_states(cfg, seeds[et], [H_, W_])]
                yx, ldx, hx = O.decoder_reverse(P, cfg, zx, [selx(c) for c in c_out], stx, [selx(e.cpu()).to(dt) for e in eps[:-1]])
            res["extra"] = dict(idx=ex, y=yx, ld=ldx, h=hx)
        del yo, ldo, hoo, z, z_out, c_out
    _ORACLE_RESULTS[okey] = res
    r64, r32 = res[torch.float64], res[torch.float32]
    rep = {"config": name, "batch": B, "loss_samples": n}
    try:
        floor = _maxabs(r32["y"], r64["y"])
        rep["loss_sample_indices"] = idx
        rep["y"] = {"hip_vs_fp64": _maxabs(sel(yr), r64["y"]), "oracle_fp32_vs_fp64": floor}
        C.assert_field(sel(yr), r64["y"], name + " y at the stated batch", atol=max(C.FIELD_ATOL, YARDSTICK * floor))
        lfloor = float(((r32["ld"].double() - r64["ld"]).abs() / r64["ld"].abs().clamp_min(1.0)).max())
        C.assert_logdet(sel(ld), r64["ld"], name + " logdet at the stated batch", rtol=max(C.LOGDET_RTOL, YARDSTICK * lfloor))
        sfl = []
        for i in range(L):
            for j in range(2):
                fl = _maxabs(r32["h"][i][j], r64["h"][i][j])
                sfl.append(fl)
                C.assert_field(sel(ho[i][j]), r64["h"][i][j], "%s state %d.%d" % (name, i, j), atol=max(C.STATE_ATOL, YARDSTICK * fl))
        # more samples, forward only, against the fp32 oracle (two fp32 evaluations: the bound is the sum of their distances from fp64,
        # the oracle's measured on the loss samples above)
        xr = res["extra"]
        if xr["idx"]:
            ext = torch.tensor(xr["idx"])
            selx = lambda t: t.index_select(0, ext.to(t.device))  # noqa: E731
            rep["forward_only_samples"] = {"indices": xr["idx"], "y_hip_vs_oracle_fp32": _maxabs(selx(yr), xr["y"])}
            C.assert_field(selx(yr), xr["y"], name + " y of the forward-only samples", atol=max(C.FIELD_ATOL, (YARDSTICK + 1) * floor))
            C.assert_logdet(selx(ld), xr["ld"], name + " logdet of the forward-only samples", rtol=max(C.LOGDET_RTOL, (YARDSTICK + 1) * lfloor))
            for i in range(L):
                for j in range(2):
                    C.assert_field(selx(ho[i][j]), xr["h"][i][j], "%s state %d.%d of the forward-only samples" % (name, i, j),
                                   atol=max(C.STATE_ATOL, (YARDSTICK + 1) * sfl[2 * i + j]))
        fl = _grad_err(r32["g"], r64["g"])
        er = _grad_err(gr, r64["g"])
        rep["reverse grads"] = {"hip_vs_fp64": er, "oracle_fp32_vs_fp64": fl, "hip_vs_oracle_fp32": _grad_err(gr, r32["g"]),
                                "stated": (C.GRAD_GLOBAL_REL_L2, C.GRAD_TENSOR_REL_MAX), "yardstick": GRAD_YARDSTICK,
                                "near_kink_relus": r64["nk"], "tensors_with_kink_allowance": len(r64["kink"])}
        assert set(gr) == set(r64["g"]), sorted(set(gr) ^ set(r64["g"]))[:5]
        C.assert_grads(gr, r64["g"], name + " reverse grads at the stated batch", global_tol=max(C.GRAD_GLOBAL_REL_L2, GRAD_YARDSTICK * fl[0]),
                       tensor_tol=max(C.GRAD_TENSOR_REL_MAX, GRAD_YARDSTICK * fl[1]), kink=r64["kink"], report=rep["reverse grads"])
        # ---- density direction (forward(x, y), loss on the same n samples): its backward runs the per-op chain, not the fused kernels
        # (on the metric configuration only: two more oracle passes per case, and the per-op chain is what cfg2 / cfg3's wide levels
        # run in the generative direction anyway)
        if density:
            m.load_state_dict(sd)
            m.zero_grad()
            zf, lpf, _, _ = m.forward(x.to(DEV), y.to(DEV), st, return_eps=True)
            C.loss_forward(sel(lpf), sel(y)).backward()
            gfw = {k: v.detach().clone() for k, v in _grads(m).items()}
            fres = {}
            for dt in (torch.float64, torch.float32):
                P = O.params_from_state_dict(sd, dtype=dt)
                kp = C.KinkProbe() if dt == torch.float64 else contextlib.nullcontext()
                with kp:
                    z_out, c_out = O.encoder(P, cfg, x.to(dt), True)
                    cmean, clsd = sel(z_out).chunk(2, 1)
                    clsd = clsd.clamp(-10.0, O.LOG5)
                    sto = [(a.to(dt), b.to(dt)) for a, b in O.init_lstm_states(cfg, seeds[it], [H_, W_])]
                    zo, ldo, _, _ = O.decoder_forward(P, cfg, sel(y).to(dt), [sel(c) for c in c_out], sto, False)
                    lpo = O.gauss_logp(cmean, clsd, zo) + ldo
                    C.loss_forward(lpo, sel(y).to(dt)).backward(retain_graph=(dt == torch.float64))
                fres[dt] = dict(z=zo.detach(), lp=lpo.detach(), g={k: v.grad.clone() for k, v in O.trainable(P).items() if v.grad is not None})
                if dt == torch.float64:
                    fres[dt]["kink"] = kp.allowances(O.trainable(P), fres[dt]["g"])
                del zo, ldo, lpo, z_out, c_out
            f64, f32 = fres[torch.float64], fres[torch.float32]
            C.assert_field(sel(zf), f64["z"], name + " z at the stated batch", atol=max(C.FIELD_ATOL, YARDSTICK * _maxabs(f32["z"], f64["z"])))
            lfl = float(((f32["lp"].double() - f64["lp"]).abs() / f64["lp"].abs().clamp_min(1.0)).max())
            C.assert_logdet(sel(lpf), f64["lp"], name + " logp at the stated batch", rtol=max(C.LOGDET_RTOL, YARDSTICK * lfl))
            ffl = _grad_err(f32["g"], f64["g"])
            rep["forward grads"] = {"hip_vs_fp64": _grad_err(gfw, f64["g"]), "oracle_fp32_vs_fp64": ffl, "hip_vs_oracle_fp32": _grad_err(gfw, f32["g"])}
            C.assert_grads(gfw, f64["g"], name + " forward grads at the stated batch", global_tol=max(C.GRAD_GLOBAL_REL_L2, GRAD_YARDSTICK * ffl[0]),
                           tensor_tol=max(C.GRAD_TENSOR_REL_MAX, GRAD_YARDSTICK * ffl[1]), kink=f64["kink"], report=rep["forward grads"])
        # the same samples at other positions of the batch (rolled by `shift`: one of them is now the LAST sample)
        roll = lambda t: torch.roll(t, shift, 0)  # noqa: E731
        it2 = (it + shift) % B
        assert int(it2.max()) == B - 1
        sel2 = lambda t: t.index_select(0, it2.to(t.device))  # noqa: E731
        yr2, ld2, _, gr2 = hip_step(roll(x).to(DEV), [(roll(a), roll(b)) for a, b in st], [roll(e) for e in eps], it2.to(DEV))
        C.assert_field(sel2(yr2), sel(yr), name + " y, batch rolled", atol=2e-5, rtol=1e-5)
        C.assert_logdet(sel2(ld2), sel(ld), name + " logdet, batch rolled", rtol=2e-6, atol=1e-3)
        rep["rolled batch grads vs first"] = _grad_err(gr2, gr)
        # (equal up to the order of the atomics and of the BatchNorm sums; the reference's fp32 noise floor is the scale)
        # (two fp32 evaluations that differ in the order of the BatchNorm sums flip different near-zero ReLUs: each tensor's own
        # measured kink allowance, nothing more)
        C.assert_grads(gr2, gr, name + " grads with the loss samples at the end of the batch", global_tol=max(1e-4, fl[0]),
                       tensor_tol=max(2e-3, fl[1]), kink=r64["kink"])
    finally:
        print("\nparity %s at batch %d: %s" % (name, B, json.dumps(rep, default=float)))
        out = os.path.join(C.ROOT, "gpurun_out")
        if os.path.isdir(out):
            with open(os.path.join(out, "parity_%s_batch%d.json" % (name, B)), "w") as f:
                json.dump(rep, f, indent=1, default=float)


STATED_CASES = [("cfg2", C.CFG2, 32, 2), ("cfg3", C.CFG3, 64, 2), ("M", C.CFG_M, 64, 3), ("cfg4", C.CFG_M, 32, 1), ("cfg5", C.CFG5, 32, 1)]


@pytest.mark.parametrize("name,cfg,B,n", [("cfg1", C.CFG1, 8, 2)] + STATED_CASES)
def test_stated_batches_match_oracle_with_gradients(name, cfg, B, n):
    """BASELINE configs[0] at its stated batch 8 (round 6: on the HIP path it had only run at the fixture's batch 2),
    configs[1] / configs[2] / the metric configuration at their STATED batch sizes (32 / 64 / 64: the benchmarked
    shapes), configs[3] (cfg4: the metric network at 32 samples per GPU = global 256 over 8; the launch plans depend on the pixel
    count), with gradients - see _stated_batch_case - and configs[4]'s five-level network at its FULL 512x512 field at batch 32
    (its 256-channel level then works on 8 192 pixels, the size at which the launch plans of the 128-channel level went wrong),
    loss on one sample, fp32 mixes (the fp16-operand variant is outside the fp32 tolerances by construction: its stated batch
    is test_cfg5_stated_batch_with_fp16_mixes).  Loss samples: scattered (M: first, middle, last)."""
    _stated_batch_case(name, cfg, B, n=n, density=(name == "M"))


@pytest.mark.parametrize("name,cfg,B,n", STATED_CASES)
def test_stated_batches_with_bf16x3_winograd_match_oracle(name, cfg, B, n):
    """ALL five stated-batch configurations with the opt-in bf16x3 arithmetic of the wide Winograd contractions
    (tmg_ops.set_winograd_precision: six bf16 MFMAs per accumulator tile on an exact three-way split of both fp32 operands) against the
    fp64 oracle with every parameter gradient - _stated_batch_case with the bounds of the fp32-MFMA path, unchanged; the oracle
    evaluations are those of the fp32 cases above (cached per process).  Round 5 pinned the metric configuration only."""
    import tmg_ops as ops
    try:
        ops.set_winograd_precision("bf16x3")
        _stated_batch_case(name + "_wino_bf16x3", cfg, B, n=n, density=False, oracle_key=name)
    finally:
        ops.set_winograd_precision("f32")


def test_cfg5_full_size_properties():
    """BASELINE configs[4] at its full field (512x512x4 output, five flow levels, default widths), batch 1, with the fp16-input
    1x1 mixes that configuration names AND with the fp32 mixes: (i) forward -> reconstruct is the identity, (ii) forward
    log-prob minus the top prior equals the generative direction's log-det on the same latents (appendix A.8), (iii) one
    generative training step has finite gradients for every live parameter."""
    import tmg_ops as ops
    from nn.tmGlow import TMGlow
    from nn.modules.flowUtils import GaussianDiag
    cfg = C.CFG5
    C.seed_all(12345)
    m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 7, *C.perturb_scales(cfg))
    m.to(DEV).train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 4, 256, 256, generator=g).to(DEV)
    y = torch.randn(1, 4, 512, 512, generator=g).to(DEV)
    h_in = m.initLSTMStates(torch.arange(1), [512, 512])
    try:
        for prec, rt_tol in (("f32", 1e-3), ("f16", 5e-2)):
            ops.set_mix_precision(prec)
            with torch.no_grad():
                z, logp, h_out, eps = m.forward(x, y, h_in, return_eps=True)
                yr, logdet, h_out2 = m.reconstruct(x, h_in, eps)
                z_out, _ = m.encoder.forward(x)
                cmean, clsd = z_out.chunk(2, 1)
                top = GaussianDiag(cmean, clsd).log_prob(z)
            assert z.shape == (1, 128, 16, 16) and yr.shape == y.shape and len(h_out) == 5
            assert float((yr - y).abs().max()) < rt_tol, (prec, float((yr - y).abs().max()))
            C.assert_logdet(logp - top, logdet, "cfg5 %s: forward logp - top prior vs reverse logdet" % prec,
                            rtol=2e-5 if prec == "f32" else 2e-3, atol=1.0)
            m.zero_grad()
            ys, ld, _ = m.sample(x, h_in)
            C.loss_reverse(ys, ld).backward()
            gr = _grads(m)
            live = [k for k, _ in m.named_parameters() if ".norm2." not in k]   # norm2 is dead in the reference
            assert all(k in gr and bool(torch.isfinite(gr[k]).all()) for k in live), [k for k in live if k not in gr][:5]
    finally:
        ops.set_mix_precision("f32")


def test_cfg5_stated_batch_with_fp16_mixes():
    """BASELINE configs[4] as stated: 512x512x4, five flow levels, 64 samples per GPU, fp16-operand / fp32-accumulate 1x1 mixes
    (reference call sites glowConv.py:193-194, :219-220).  The fp16 variant is outside the fp32 tolerances by construction
    (SURVEY 8-C), so the yardstick is this package's own fp32 path on the SAME batch (which test_stated_batches_match_oracle_with_
    gradients pins against the fp64 oracle at batch 32): forward -> reconstruct over the whole batch in both precisions, the A.8
    log-det identity, and one generative training step whose gradients are compared tensor by tensor with loose fp16 bounds.
    At this batch the 128- and 256-channel levels run the launch plans of 65 536 / 16 384 pixels that no smaller case reaches."""
    import json
    import os
    import tmg_ops as ops
    from nn.tmGlow import TMGlow
    from nn.modules.flowUtils import GaussianDiag
    cfg, B = C.CFG5, 64
    C.seed_all(12345)
    m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 7, *C.perturb_scales(cfg))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(DEV).train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 4, 256, 256, generator=g).to(DEV)
    y = torch.randn(B, 4, 512, 512, generator=g).to(DEV)
    st = m.initLSTMStates(torch.arange(B) + 2, [512, 512])
    res = {}
    try:
        for prec in ("f32", "f16"):
            ops.set_mix_precision(prec)
            m.load_state_dict(sd)
            with torch.no_grad():
                z, logp, _, eps = m.forward(x, y, st, return_eps=True)
                z_out, _ = m.encoder.forward(x)
                cmean, clsd = z_out.chunk(2, 1)
                top = GaussianDiag(cmean, clsd).log_prob(z)
                m.load_state_dict(sd)
                # the generative direction on the FP32 run's latents in both precisions: the same function evaluated twice
                e_in = [t.detach() for t in (res["f32"]["eps"] if prec == "f16" else eps)]
                y_own, ld_own, _ = m.reconstruct(x, st, [t.detach() for t in eps])      # round trip / A.8 on this precision's own latents
            m.load_state_dict(sd)
            m.zero_grad()
            yr, ld, _ = m.reconstruct(x, st, e_in)
            C.loss_reverse(yr, ld).backward()
            res[prec] = dict(z=z, logp=logp, top=top, eps=eps, y=yr.detach(), ld=ld.detach(), ld_own=ld_own,
                             g={k: v.clone() for k, v in _grads(m).items()}, roundtrip=float((y_own - y).abs().max()))
            del z, logp, eps, yr, ld, y_own, ld_own
    finally:
        ops.set_mix_precision("f32")
    a, b = res["f16"], res["f32"]
    live = [k for k, _ in m.named_parameters() if ".norm2." not in k]
    assert all(k in a["g"] and bool(torch.isfinite(a["g"][k]).all()) for k in live)
    ge = _grad_err(a["g"], b["g"])
    rep = {"batch": B, "roundtrip_f32": b["roundtrip"], "roundtrip_f16": a["roundtrip"],
           "z_maxabs": _maxabs(a["z"], b["z"]), "z_scale": float(b["z"].abs().max()),
           "y_maxabs": _maxabs(a["y"], b["y"]),
           "logp_rel": float(((a["logp"] - b["logp"]).abs() / b["logp"].abs().clamp_min(1.0)).max()),
           "logdet_rel": float(((a["ld"] - b["ld"]).abs() / b["ld"].abs().clamp_min(1.0)).max()),
           "reverse_grads_f16_vs_f32": ge}
    print("\ncfg5 at batch %d, fp16-operand mixes against the fp32 HIP path: %s" % (B, json.dumps(rep, default=float)))
    out = os.path.join(C.ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_cfg5_batch64_fp16_mix.json"), "w") as f:
            json.dump(rep, f, indent=1, default=float)
    assert rep["z_maxabs"] > 0.0, "the fp16 variant must actually run"
    assert b["roundtrip"] < 2e-3 and a["roundtrip"] < 5e-2, (b["roundtrip"], a["roundtrip"])
    for r in (a, b):     # A.8: forward log-prob minus the top prior = the generative direction's log-det on the same latents
        C.assert_logdet(r["logp"] - r["top"], r["ld_own"], "cfg5 batch 64: logp - top prior vs reverse logdet",
                        rtol=2e-5 if r is b else 2e-3, atol=1.0)
    assert rep["z_maxabs"] < 5e-2 * max(rep["z_scale"], 1.0) and rep["y_maxabs"] < 5e-2
    assert rep["logp_rel"] < 2e-3 and rep["logdet_rel"] < 2e-3
    # gradients: the loss sums 64 x 262 144 pixels of fields that are close to the N(0,1) targets the latents were made from, so the
    # sums nearly cancel and the fp16 rounding of 80 mixes shows up amplified (measured 6.1e-2 global at this batch, 1.1e-2 on the
    # 64x64 / batch-2 case of test_fp16_mix_variant_deviation_is_reported_separately); a plan fault in mix16_kernel at these pixel
    # counts would be O(1), like the round-3 1x1 weight-gradient fault was
    assert ge[0] < 0.15, ge
    # ... and per tensor, so that a dead or garbage tensor cannot hide in the global norm: the deviation of every gradient tensor relative
    # to max(its own scale, the RMS over ALL gradient elements) - a tensor the fp16 variant left at zero (or filled with noise of the
    # model's gradient scale) reads >= 1 here whatever its size; tensors whose own gradients nearly cancel are held to the global scale
    tot = sum(float((b["g"][k].double() ** 2).sum()) for k in live)
    cnt = sum(b["g"][k].numel() for k in live)
    rms = (tot / cnt) ** 0.5
    per = []
    for k in live:
        ga, gb = a["g"][k].double(), b["g"][k].double()
        scale = max(float(gb.abs().max()), rms)
        per.append((float((ga - gb).abs().max()) / scale, k))
        assert float(gb.abs().max()) == 0.0 or float(ga.abs().max()) > 0.0, "fp16 variant: dead gradient tensor %s" % k
        # (garbage - uninitialised memory, a half-summed slab - is orders of magnitude off in size)
        assert float(ga.abs().max()) <= 8.0 * max(float(gb.abs().max()), rms), "fp16 variant: gradient tensor %s of another magnitude" % k
    per.sort(reverse=True)
    # The mixes' OWN parameters (PLU factors l, u, log_s and the ActNorm folded into them) are a class of their own: their gradient is
    # a sum of dy x^T over 64 x 262 144 pixels that cancels to ~1 / sqrt(N) = 2.4e-4 of its terms, while the fp16 operands shift every
    # term COHERENTLY by up to 2^-11 = 4.9e-4 - the deviation is of the size of the gradient itself (measured: up to 2.4x its scale;
    # in fp32 the same tensors are the ill-conditioned ones of the kink analysis).  They are held to "alive and of the right
    # magnitude" (above); every other tensor to 0.8 of max(its scale, global RMS) - a dead tensor reads exactly 1.0 - with the worst of
    # them at 0.52 (the encoder's first convs: pixel sums upstream of all 80 mixes; distribution in the report, median 0.07)
    own = lambda k: any(t in k for t in (".conv.l", ".conv.u", ".conv.log_s", ".norm.weight", ".norm.bias"))  # noqa: E731
    others = [(v, k) for v, k in per if not own(k)]
    rep["per_tensor_rel_to_max_of_scale_and_global_rms"] = {
        "global_rms": rms, "worst_mix_parameters": [{"tensor": k, "rel": v} for v, k in per if own(k)][:5],
        "worst_other_tensors": [{"tensor": k, "rel": v} for v, k in others[:8]], "median": per[len(per) // 2][0],
        "n_over_0.5": sum(1 for v, _ in per if v > 0.5), "n_other_over_0.25": sum(1 for v, _ in others if v > 0.25)}
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_cfg5_batch64_fp16_mix.json"), "w") as f:
            json.dump(rep, f, indent=1, default=float)
    assert others[0][0] < 0.8, others[:3]


def test_fp16_mix_variant_deviation_is_reported_separately():
    """The fp16-input / fp32-accumulate 1x1 mixes (BASELINE configs[4]; reference call sites glowConv.py:193-194, :219-220) are
    outside the fp32 tolerances by construction: their deviation from this package's own fp32 path is measured
    on the five-level network and bounded loosely (fp16 has 11 significant bits and the error compounds through 80 mixes);
    the numbers go to gpurun_out/parity_fp16_mix.json."""
    import json
    import os
    import tmg_ops as ops
    from nn.tmGlow import TMGlow
    cfg = CFG5_REDUCED
    C.seed_all(12345)
    m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 7, *C.perturb_scales(cfg))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(DEV).train()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 4, 32, 32, generator=g).to(DEV)
    y = torch.randn(2, 4, 64, 64, generator=g).to(DEV)
    st = m.initLSTMStates(torch.arange(2), [64, 64])
    res = {}
    try:
        for prec in ("f32", "f16"):
            ops.set_mix_precision(prec)
            m.load_state_dict(sd)
            m.zero_grad()
            z, lp, _, e = m.forward(x, y, st, return_eps=True)
            C.loss_forward(lp, y).backward()
            gf = {k: v.clone() for k, v in _grads(m).items()}
            m.load_state_dict(sd)
            m.zero_grad()
            eps_in = [t.detach() for t in (res["f32"]["e"] if prec == "f16" else e)]
            yr, ld, _ = m.reconstruct(x, st, eps_in)
            C.loss_reverse(yr, ld).backward()
            res[prec] = dict(z=z.detach(), lp=lp.detach(), e=e, gf=gf, y=yr.detach(), ld=ld.detach(), gr=dict(_grads(m)))
    finally:
        ops.set_mix_precision("f32")
    a, b = res["f16"], res["f32"]
    rep = {"z_maxabs": _maxabs(a["z"], b["z"]), "z_scale": float(b["z"].abs().max()),
           "logp_rel": float(((a["lp"] - b["lp"]).abs() / b["lp"].abs()).max()),
           "y_maxabs": _maxabs(a["y"], b["y"]), "logdet_rel": float(((a["ld"] - b["ld"]).abs() / b["ld"].abs().clamp_min(1.0)).max()),
           "forward_grads": _grad_err(a["gf"], b["gf"]), "reverse_grads": _grad_err(a["gr"], b["gr"])}
    print("\nfp16-mix deviation from the fp32 HIP path:", json.dumps(rep, default=float))
    out = os.path.join(C.ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_fp16_mix.json"), "w") as f:
            json.dump(rep, f, indent=1, default=float)
    assert rep["z_maxabs"] > 0.0, "the fp16 variant must actually run (identical results mean the switch did nothing)"
    assert rep["z_maxabs"] < 5e-2 * max(rep["z_scale"], 1.0) and rep["y_maxabs"] < 5e-2
    assert rep["logp_rel"] < 2e-3 and rep["logdet_rel"] < 2e-3
    assert rep["forward_grads"][0] < 5e-2 and rep["reverse_grads"][0] < 5e-2


def test_flow_level_module_matches_reference():
    from nn.modules.flowLSTMBlock import LSTMFLowBlock
    d = C.load_npz("modules.npz")
    blk = LSTMFLowBlock(2, 3, 5, 3, LUdecompose=True, train_sampling=True, do_split=True, squeeze_type=0)
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in C.sub(d, "level.sd.").items()}, strict=True)
    blk.to(DEV)
    t = lambda k: torch.from_numpy(d[k]).to(DEV)  # noqa: E731
    x, cond, h, c = (t(k).requires_grad_(True) for k in ("level.x", "level.cond", "level.h", "level.c"))
    z, ld, st, eps = blk.forward(x, cond, (h, c), return_eps=True)
    C.assert_field(z, d["level.fwd.z"], atol=2e-5)
    C.assert_logdet(ld, d["level.fwd.logdet"])
    C.assert_field(eps, d["level.fwd.eps"], atol=2e-5)
    loss = (z * t("level.wz")).sum() + ld.sum() * 0.01 + (st[0] * t("level.wh")).sum() + (st[1] * t("level.wc")).sum()
    loss.backward()
    got = {k: p.grad for k, p in blk.named_parameters() if p.grad is not None}
    got.update({"@dx": x.grad, "@dcond": cond.grad, "@dh": h.grad, "@dc": c.grad})
    ref = C.sub(d, "level.fwd.grad.")
    ref.update({"@dx": d["level.fwd.dx"], "@dcond": d["level.fwd.dcond"], "@dh": d["level.fwd.dh"], "@dc": d["level.fwd.dc"]})
    C.assert_grads(got, ref, "level fwd grads")
    blk.zero_grad()
    for v in (cond, h, c):
        v.grad = None
    zin = t("level.fwd.z").requires_grad_(True)
    xr, ldr, st2 = blk.reverse(zin, cond, (h, c), eps=t("level.fwd.eps"))
    C.assert_field(xr, d["level.rev.x"], atol=2e-5)
    C.assert_logdet(ldr, d["level.rev.logdet"])
    loss = (xr * t("level.wx")).sum() + ldr.sum() * 0.01 + (st2[0] * t("level.wh")).sum() + (st2[1] * t("level.wc")).sum()
    loss.backward()
    got = {k: p.grad for k, p in blk.named_parameters() if p.grad is not None}
    got.update({"@dz": zin.grad, "@dcond": cond.grad, "@dh": h.grad, "@dc": c.grad})
    ref = C.sub(d, "level.rev.grad.")
    ref.update({"@dz": d["level.rev.dz"], "@dcond": d["level.rev.dcond"], "@dh": d["level.rev.dh"], "@dc": d["level.rev.dc"]})
    C.assert_grads(got, ref, "level rev grads")


def test_standalone_primitives_match_oracle():
    """ActNorm, InvertibleConv1x1LU, plain InvertibleConv1x1, Conv2dZeros, CheckerSqueeze, Squeeze stand-alone."""
    from nn.modules.actNorm import ActNorm
    from nn.modules.glowConv import InvertibleConv1x1, InvertibleConv1x1LU
    from nn.modules.flowUtils import Conv2dZeros, Squeeze
    from oracle import tmglow_oracle as O
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 8, 6, 10, generator=g)
    an = ActNorm(8)
    with torch.no_grad():
        an.weight.add_(0.3 * torch.randn(8, 1, 1, generator=g))
        an.bias.add_(0.3 * torch.randn(8, 1, 1, generator=g))
    P = {"weight": an.weight.detach().clone(), "bias": an.bias.detach().clone()}
    an.to(DEV)
    for rev in (False, True):
        y, ld = (an.reverse if rev else an.forward)(x.to(DEV))
        yo, ldo = O.actnorm(P, "", x, rev)
        C.assert_field(y, yo, atol=1e-5)
        assert abs(float(ld) - float(ldo)) < 1e-3
    C.seed_all(5)
    lu = InvertibleConv1x1LU(8)
    P = {k: v.detach().clone() for k, v in lu.state_dict().items()}
    lu.to(DEV)
    for rev in (False, True):
        y, ld = (lu.reverse if rev else lu.forward)(x.to(DEV))
        yo, ldo = O.invconv_lu(P, "", x, rev)
        C.assert_field(y, yo, atol=1e-5)
        assert abs(float(ld) - float(ldo)) < 1e-3
    d = C.load_npz("modules.npz")
    for ts in (1, 0):
        tag = "plain1x1.ts%d." % ts
        pl = InvertibleConv1x1(6, train_sampling=bool(ts))
        with torch.no_grad():
            pl.weight.copy_(torch.from_numpy(d[tag + "weight"]))
        pl.to(DEV)
        xx = torch.from_numpy(d[tag + "x"]).to(DEV)
        y, ld = pl.forward(xx)
        C.assert_field(y, d[tag + "fwd.y"], atol=1e-5)
        assert abs(float(ld) - float(d[tag + "fwd.logdet"])) < 1e-3
        y, ld = pl.reverse(xx)
        C.assert_field(y, d[tag + "rev.y"], atol=1e-5)
    zc = Conv2dZeros(8, 6)
    with torch.no_grad():
        zc.conv.weight.copy_(0.2 * torch.randn(6, 8, 3, 3, generator=g))
        zc.conv.bias.copy_(0.2 * torch.randn(6, generator=g))
        zc.scale.fill_(0.4)
    P = {k: v.detach().clone() for k, v in zc.state_dict().items()}
    zc.to(DEV)
    C.assert_field(zc(x.to(DEV)), O.zero_conv(P, "", x), atol=1e-5)
    sq = Squeeze(2)
    xs = torch.from_numpy(d["squeeze.x"])
    assert np.array_equal(sq(xs).numpy(), d["squeeze.y"]) and np.array_equal(sq.reverse(sq(xs)).numpy(), d["squeeze.x"])


def test_cpu_tensors_are_rejected_loudly():
    from nn.modules.flowUtils import Conv2dZeros
    with pytest.raises(RuntimeError):
        Conv2dZeros(4, 4)(torch.randn(1, 4, 4, 4))


def test_full_size_round_trip_and_logdet_bookkeeping():
    """Metric configuration (256x256x4 output, L=4, K=16, default widths) at batch 2: size-independent properties.
    (i) forward -> reconstruct is the identity (the reference's own self-test, tmGlow.py:511-530);
    (ii) forward log-prob minus the top prior equals the log-det reported by the generative direction on the same
         latents (every per-layer term has the same sign in both directions)."""
    from nn.tmGlow import TMGlow
    from nn.modules.flowUtils import GaussianDiag
    cfg = C.CFG_M
    C.seed_all(12345)
    m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 7, 0.004, 0.02, 0.004)
    m.to(DEV).train()
    g = torch.Generator().manual_seed(3)
    B = 2
    x = torch.randn(B, 4, 128, 128, generator=g).to(DEV)
    y = torch.randn(B, 4, 256, 256, generator=g).to(DEV)
    h_in = m.initLSTMStates(torch.arange(B), [256, 256])
    with torch.no_grad():
        z, logp, h_out, eps = m.forward(x, y, h_in, return_eps=True)
        yr, logdet, h_out2 = m.reconstruct(x, h_in, eps)
        z_out, _ = m.encoder.forward(x)
        cmean, clsd = z_out.chunk(2, 1)
        top = GaussianDiag(cmean, clsd).log_prob(z)
    assert z.shape == (B, 64, 16, 16) and yr.shape == y.shape
    assert float((yr - y).abs().max()) < 5e-4
    C.assert_logdet(logp - top, logdet, "forward logp - top prior vs reverse logdet", rtol=2e-5, atol=0.5)
    for (h1, c1), (h2, c2) in zip(h_out, h_out2):
        C.assert_field(h1, h2, "h states both directions", atol=5e-4)


def test_level_fused_node_matches_per_layer_path():
    """The level-fused coupling node (cond contributions batched over layers) against the per-layer path, both on HIP."""
    import os
    from nn.modules.flowLSTMBlock import LSTMFLowBlock
    C.seed_all(77)
    blk = LSTMFLowBlock(4, 32, 16, 5, LUdecompose=True, train_sampling=True, do_split=True, squeeze_type=0)
    C.perturb_(blk, 5, 0.05, 0.1, 0.05)
    blk.to(DEV)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 4, 24, 40, generator=g).to(DEV)
    cond = torch.randn(3, 32, 12, 20, generator=g).to(DEV)
    res = {}
    for tag, env in (("fused", None), ("plain", "1")):
        if env:
            os.environ["TMG_NO_LEVEL_FUSION"] = env
        else:
            os.environ.pop("TMG_NO_LEVEL_FUSION", None)
        blk.zero_grad()
        xi = x.clone().requires_grad_(True)
        ci = cond.clone().requires_grad_(True)
        z, ld, st, eps = blk.forward(xi, ci, None, return_eps=True)
        xr, ldr, _ = blk.reverse(z, ci, None, eps=eps.detach())
        ((z ** 2).sum() + ld.sum() * 0.01 + (xr ** 2).sum() * 0.5 + ldr.sum() * 0.02).backward()
        res[tag] = (z.detach(), ld.detach(), xr.detach(), {k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None},
                    xi.grad.clone(), ci.grad.clone())
    os.environ.pop("TMG_NO_LEVEL_FUSION", None)
    a, b = res["fused"], res["plain"]
    C.assert_field(a[0], b[0], "z")
    C.assert_logdet(a[1], b[1])
    C.assert_field(a[2], b[2], "x_rec")
    ga, gb = dict(a[3]), dict(b[3])
    ga.update({"@dx": a[4], "@dcond": a[5]})
    gb.update({"@dx": b[4], "@dcond": b[5]})
    C.assert_grads(ga, gb, "fused vs per-layer grads", global_tol=1e-4, tensor_tol=2e-3)


@pytest.mark.parametrize("cin,hw", [(16, (16, 32)), (32, (8, 16)), (16, (6, 10))])
def test_fused_coupling_mix_launches_match_the_per_op_chain(cin, hw):
    """tmg_mix_f32_affine_fwd / _bwd (64- / 128-channel levels, generative direction: coupling + trailing mix, and the mix input
    gradient + coupling backward, one launch each) against the separate affine and mix launches they replace (TMG_NO_MIX_AFFINE=1):
    level output, log-det and every gradient.  The last case has a pixel count per image the fused kernels do not take (15 pixels):
    the level node must fall back to the per-op chain by itself."""
    import os
    from nn.modules.flowLSTMBlock import LSTMFLowBlock
    C.seed_all(79)
    blk = LSTMFLowBlock(cin, 32, 16, 4, LUdecompose=True, train_sampling=True, do_split=False, squeeze_type=0)
    C.perturb_(blk, 7, 0.05, 0.1, 0.05)
    blk.to(DEV)
    g = torch.Generator().manual_seed(11)
    hs, ws = hw[0] // 2, hw[1] // 2
    z = torch.randn(3, 4 * cin, hs, ws, generator=g).to(DEV)
    cond = torch.randn(3, 32, hs, ws, generator=g).to(DEV)
    res = {}
    for tag, env in (("fused", None), ("per-op", "1")):
        if env:
            os.environ["TMG_NO_MIX_AFFINE"] = env
        try:
            blk.zero_grad()
            ci = cond.clone().requires_grad_(True)
            zi = z.clone().requires_grad_(True)
            xr, ldr, _ = blk.reverse(zi, ci, None)
            ((xr ** 2).sum() * 0.5 + ldr.sum() * 0.02).backward()
        finally:
            os.environ.pop("TMG_NO_MIX_AFFINE", None)
        res[tag] = (xr.detach(), ldr.detach(), ci.grad.clone(), zi.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None})
    a, b = res["fused"], res["per-op"]
    C.assert_field(a[0], b[0], "x", atol=2e-5, rtol=1e-5)
    C.assert_logdet(a[1], b[1], rtol=2e-6, atol=1e-3)
    ga, gb = dict(a[4]), dict(b[4])
    ga.update({"@dcond": a[2], "@dz": a[3]})
    gb.update({"@dcond": b[2], "@dz": b[3]})
    C.assert_grads(ga, gb, "fused coupling + mix vs per-op chain", global_tol=1e-5, tensor_tol=2e-4)


def test_layer_plane_addends_give_identical_results():
    """The growth convs' conditioning addends as one float2 plane per layer (tmg_layer_planes, used on large images) against the
    interleaved [B,H,W,2K] tensor: same numbers in another layout, so outputs and gradients are identical bit for bit."""
    import os
    from nn.modules.flowLSTMBlock import LSTMFLowBlock
    C.seed_all(78)
    blk = LSTMFLowBlock(4, 32, 16, 5, LUdecompose=True, train_sampling=True, do_split=False, squeeze_type=0)   # (no split: no fresh latents per call)
    C.perturb_(blk, 6, 0.05, 0.1, 0.05)
    blk.to(DEV)
    g = torch.Generator().manual_seed(10)
    z = torch.randn(3, 16, 12, 20, generator=g).to(DEV)
    cond = torch.randn(3, 32, 12, 20, generator=g).to(DEV)
    res = {}
    for tag, env in (("planes", "1"), ("interleaved", str(1 << 40))):
        os.environ["TMG_LAYER_PLANES_MIN"] = env
        try:
            blk.zero_grad()
            ci = cond.clone().requires_grad_(True)
            zi = z.clone().requires_grad_(True)
            xr, ldr, _ = blk.reverse(zi, ci, None)
            ((xr ** 2).sum() * 0.5 + ldr.sum() * 0.02).backward()
        finally:
            os.environ.pop("TMG_LAYER_PLANES_MIN", None)
        res[tag] = (xr.detach(), ci.grad.clone(), zi.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None})
    a, b = res["planes"], res["interleaved"]
    assert torch.equal(a[0], b[0])
    # (log-det sums, weight gradients and the corner pixels of the replicate-padding adjoint meet in float atomics: equal up to the
    # summation order, i.e. to a few ulp OF THE LARGEST TERMS - measured run to run on one layout: 3e-5 on entries of 4e2, 1.4e-6 on an
    # entry of -0.024 that is the difference of such terms.  The bound is therefore relative to the tensor's scale, not to each entry.)
    def same(x, y, rel):
        return float((x - y).abs().max()) <= rel * float(y.abs().max()) + 1e-12
    assert same(a[1], b[1], 1e-6) and same(a[2], b[2], 1e-6)
    for k in a[3]:
        assert same(a[3][k], b[3][k], 1e-5), k


def test_training_window_capture_matches_reference():
    """A16 on the HIP path: three optimizer steps of the trainer's inner loop (3-step BPTT windows with LSTM-state
    gradients flowing across time-steps, clip, Adam(amsgrad), state re-anchoring) against the capture recorded from
    the reference; latents injected through reconstruct()."""
    import tmg_dist
    d = C.load_npz("tiny_train.npz")
    cfg = C.CFG_TINY
    L = len(cfg["glow_blocks"])
    m = _model(cfg, {k: torch.from_numpy(v) for k, v in C.sub(d, "sd.").items()})
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
    a_key = m.initLSTMStates(torch.from_numpy(d["seeds"]), [16, 16])
    a0 = [(h.clone(), c.clone()) for h, c in a_key]
    xs = torch.from_numpy(d["xs"]).to(DEV)
    for a in range(xs.shape[0]):
        tol = 1.0 + 4.0 * a
        eps = [[torch.from_numpy(d["eps.%d.%d.%d" % (a, t, i)]).to(DEV) for i in range(L + 1)] for t in range(xs.shape[1])]
        loss, gn, a0, outs = tmg_dist.train_window(m, opt, [xs[a, t] for t in range(xs.shape[1])], a0, a_key, C.loss_reverse,
                                                   max_grad_norm=float(d["max_grad_norm"]),
                                                   sample=lambda mod, x, st, t: mod.reconstruct(x, st, eps[t]))
        for t, (y, logp) in enumerate(outs):
            C.assert_field(y, d["step%d.t%d.y" % (a, t)], "y", atol=C.FIELD_ATOL * tol, rtol=C.FIELD_RTOL * tol)
            C.assert_logdet(logp, d["step%d.t%d.logp" % (a, t)], rtol=C.LOGDET_RTOL * 10 * tol)
        assert abs(float(loss) - float(d["step%d.loss" % a])) < 2e-4 * tol
        assert abs(float(gn) - float(d["step%d.gradnorm" % a])) < 2e-3 * float(d["step%d.gradnorm" % a]) * tol
        log_s = dict(m.named_parameters())[str(d["log_s_key"])]
        C.assert_field(log_s, d["step%d.log_s" % a], "log_s", atol=2e-5 * tol)


@pytest.mark.parametrize("cfg_name", ["tiny", "tiny3"])
def test_fused_grad_accumulation_matches_autograd_accumulation(cfg_name):
    """tmg_ops.bptt_window (parameter-only tensors evaluated once per window, the window's parameter gradients summed with multi-tensor
    launches instead of one AccumulateGrad add per parameter and time-step; trainFlowParallel.py:256-287 is the loop it serves): a
    three-step window with recurrent states gives the gradients of plain `loss.backward()` - every live parameter, including the
    zero-padded level of a 3-channel field - also on top of gradients that exist already, and nothing leaks out of the contexts.
    Two forward passes followed by two SEPARATE backward passes (outside a window) still work: nothing with a graph is shared there."""
    import tmg_ops as ops
    name, cfg = ("tiny_model.npz", C.CFG_TINY) if cfg_name == "tiny" else ("tiny3_model.npz", C.CFG_TINY3)
    d = C.load_npz(name)
    L = len(cfg["glow_blocks"])
    m = _model(cfg, {k: torch.from_numpy(v) for k, v in C.sub(d, "sd.").items()})
    x = torch.from_numpy(d["x"]).to(DEV)
    h_in = C.states_from(d, "h_in.", L, DEV)
    eps = [torch.from_numpy(d["fwd.eps.%d" % i]).to(DEV) for i in range(L + 1)]

    def window(fused, twice=False):
        m.zero_grad(set_to_none=True)
        for rep in range(2 if twice else 1):
            if fused:     # the trainer's form: shared parameter-only tensors + summed parameter gradients
                with ops.bptt_window() as win:
                    st, loss = h_in, 0.0
                    for t in range(3):
                        y, ld, st = m.reconstruct(x * (1.0 + 0.1 * t), st, eps)
                        loss = loss + C.loss_reverse(y, ld)
                    win.backward(loss)
                assert ops._GradSink.active is None and ops.DerivedCache.window_depth == 0
            else:         # plain autograd: every forward pass builds its own folded mixes / padded weights
                st, loss = h_in, 0.0
                for t in range(3):
                    y, ld, st = m.reconstruct(x * (1.0 + 0.1 * t), st, eps)
                    loss = loss + C.loss_reverse(y, ld)
                loss.backward()
        return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    for twice in (False, True):
        ref, got = window(False, twice), window(True, twice)
        assert set(ref) == set(got) and len(ref) > 50
        # (identical sums in the same order; the kernels' float atomics are the only difference between two runs)
        C.assert_grads(got, ref, "fused accumulation", global_tol=2e-6, tensor_tol=2e-5)
    # outside a window: two forward passes, then two separate backward passes
    m.zero_grad(set_to_none=True)
    ya, la, _ = m.reconstruct(x, h_in, eps)
    yb, lb, _ = m.reconstruct(x * 1.1, h_in, eps)
    C.loss_reverse(ya, la).backward()
    C.loss_reverse(yb, lb).backward()
    with pytest.raises(RuntimeError, match="re-entrant"):
        with ops.fused_grad_accumulation():
            with ops.fused_grad_accumulation():
                pass
    assert ops._GradSink.active is None


@pytest.mark.parametrize("cfg_name,recompute", [("tiny", False), ("tiny3", False), ("tiny", True)])
def test_captured_window_matches_eager_window(cfg_name, recompute, request):
    """tmg_dist.CapturedWindow (forward passes + loss + backward of a BPTT window as one hipGraph replay; the loop it serves is
    trainFlowParallel.py:256-281): three windows of three time-steps with recurrent states, an optimizer step between them (the
    replay must read the updated parameters in place) and new inputs per window (copied into the graph's input tensors) give the
    loss, every parameter gradient, the new states and the parameters of the same windows run eagerly - also for the zero-padded
    level of a 3-channel field, whose grouped weight-gradient tables are written by a kernel during capture (tmg_fill_i64)."""
    import tmg_dist
    import tmg_ops as ops
    name, cfg = ("tiny_model.npz", C.CFG_TINY) if cfg_name == "tiny" else ("tiny3_model.npz", C.CFG_TINY3)
    d = C.load_npz(name)
    L = len(cfg["glow_blocks"])
    sd = {k: torch.from_numpy(v) for k, v in C.sub(d, "sd.").items()}
    me, mc = _model(cfg, sd), _model(cfg, sd)
    x = torch.from_numpy(d["x"]).to(DEV)
    h_in = C.states_from(d, "h_in.", L, DEV)
    eps = [torch.from_numpy(d["fwd.eps.%d" % i]).to(DEV) for i in range(L + 1)]

    def body_of(m):
        def body(xs, states):
            loss = 0.0
            for t in range(len(xs)):
                y, ld, states = m.reconstruct(xs[t], states, eps)
                loss = loss + C.loss_reverse(y, ld)
            return loss, states
        return body

    inputs = lambda w: [x * (1.0 + 0.1 * t + 0.05 * w) for t in range(3)]  # noqa: E731
    if recompute:      # (round 5) the recompute-from-output backward inside the recorded window, against eager windows in the same mode
        ops.set_recompute(True)
        request.addfinalizer(lambda: ops.set_recompute(False))
    cw = tmg_dist.CapturedWindow(mc, body_of(mc), (inputs(0), h_in))
    assert all(p.grad is None for p in mc.parameters())      # recording leaves no gradient behind
    oe, oc = torch.optim.SGD(me.parameters(), lr=1e-3), torch.optim.SGD(mc.parameters(), lr=1e-3)
    st_e = st_c = h_in
    for w in range(3):
        xs = inputs(w)
        oe.zero_grad(set_to_none=True)
        with ops.bptt_window() as win:
            loss_e, new_e = body_of(me)(xs, st_e)
            win.backward(loss_e)
        oc.zero_grad(set_to_none=True)
        loss_c, new_c = cw(xs, st_c)
        le, lc = float(loss_e.detach()), float(loss_c)
        assert abs(lc - le) <= 2e-6 * abs(le) + 1e-6, (w, lc, le)
        ge = {k: p.grad for k, p in me.named_parameters() if p.grad is not None}
        gc = {k: p.grad for k, p in mc.named_parameters() if p.grad is not None}
        assert set(ge) == set(gc) and len(ge) > 50
        C.assert_grads(gc, ge, "captured window %d" % w, global_tol=2e-6, tensor_tol=2e-5)
        for (he, ce), (hc, cc) in zip(new_e, new_c):
            assert torch.allclose(hc, he.detach(), rtol=1e-5, atol=1e-6) and torch.allclose(cc, ce.detach(), rtol=1e-5, atol=1e-6)
        oe.step()
        oc.step()
        st_c = [(0.5 * h + 0.5 * hk, 0.5 * c_ + 0.5 * ck) for (h, c_), (hk, ck) in zip(new_c, h_in)]
        # every window is compared from the SAME parameters and states: the kernels' float atomics make two runs differ in the last
        # bits, and an optimizer step and a recurrent state would carry that difference into the next window's comparison
        with torch.no_grad():
            for (k, pe), (_, pc) in zip(me.named_parameters(), mc.named_parameters()):
                assert torch.allclose(pc, pe, rtol=1e-5, atol=1e-7), (w, k)
                pe.copy_(pc)
            for (k, be), (_, bc) in zip(me.named_buffers(), mc.named_buffers()):
                be.copy_(bc)
        st_e = [(h.clone(), c_.clone()) for h, c_ in st_c]
    assert cw.replays == 3
    with pytest.raises(ValueError, match="structure / shape"):
        cw(inputs(0)[:2], h_in)
    # the graph reads the parameters in place: a parameter whose storage moved makes the recorded window unusable, loudly
    p0 = next(p_ for p_ in mc.parameters() if p_.requires_grad)
    keep = p0.data
    p0.data = keep.clone()
    with pytest.raises(RuntimeError, match="storage moved"):
        cw(inputs(0), h_in)
    p0.data = keep
    cw(inputs(0), h_in)
    # fresh latents inside a replayed graph (model.sample draws them with torch's graph-safe Philox offsets): two replays on the
    # same inputs give different fields
    def sample_body(xs, states):
        y, ld, states = mc.sample(xs[0], states)
        return C.loss_reverse(y, ld), (y, states)
    cs = tmg_dist.CapturedWindow(mc, sample_body, (inputs(0)[:1], h_in))
    _, (ya, _) = cs(inputs(0)[:1], h_in)
    ya = ya.clone()
    _, (yb, _) = cs(inputs(0)[:1], h_in)
    assert torch.isfinite(ya).all() and torch.isfinite(yb).all() and float((ya - yb).abs().max()) > 1e-3


def test_forward_default_arguments():
    """forward(x, y) with the reference's defaults (no states, return_eps=False): same z / log-likelihood as with
    return_eps=True, eps is None (reference tmGlow.py:378-414)."""
    d = C.load_npz("tiny_model.npz")
    cfg = C.CFG_TINY
    m = _model(cfg, {k: torch.from_numpy(v) for k, v in C.sub(d, "sd.").items()})
    x, y = torch.from_numpy(d["x"]).to(DEV), torch.from_numpy(d["y"]).to(DEV)
    with torch.no_grad():
        z0, lp0, h0, e0 = m.forward(x, y)
        z1, lp1, h1, e1 = m.forward(x, y, None, return_eps=True)
    assert e0 is None and len(e1) == len(cfg["glow_blocks"]) + 1
    assert torch.equal(z0, z1) and torch.allclose(lp0, lp1, rtol=2e-6, atol=0)   # (log-likelihood: atomically accumulated sums)
    assert lp0.shape == (x.shape[0],) and len(h0) == len(cfg["glow_blocks"])
    # generative direction without states: zero LSTM states (reference convLSTM.py:87-104), fresh latents
    with torch.no_grad():
        ys, ld, hs = m.sample(x)
        yr, ldr, _ = m.reconstruct(x, None, e1)
    assert ys.shape == y.shape and ld.shape == (x.shape[0],) and len(hs) == len(cfg["glow_blocks"])
    assert torch.isfinite(ys).all() and torch.isfinite(ld).all()
    C.assert_field(yr, y.cpu().numpy(), "forward -> reconstruct round trip", atol=5e-4, rtol=1e-3)


def test_sample_draws_standard_normal_independent_latents():
    """TMGlow.sample's device-drawn latents (reference tmGlow.py:417-440, flowUtils.py:206 / :328: randn_like per level).  The latents
    of a call are recovered through the flow's own inverse - forward(x, sample(x)) returns them as eps - and must be N(0, 1) per level
    (mean, variance, skewness, kurtosis within a few standard errors), uncorrelated between levels, between the samples of a batch,
    between neighbouring pixels / channels and between two calls; the same torch.manual_seed gives the same field again, the log-det
    returned by sample() is the one reconstruct() returns for those latents."""
    cfg = C.CFG1
    C.seed_all(12345)
    import contextlib
    import io
    from nn.tmGlow import TMGlow
    with contextlib.redirect_stdout(io.StringIO()):
        m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 7, *C.perturb_scales(cfg))
    m.to(DEV).train()
    B, (h, w) = 16, cfg["_in_hw"]
    H_, W_ = h * cfg["_up"], w * cfg["_up"]
    L = len(cfg["glow_blocks"])
    x = torch.randn(B, cfg["in_features"], h, w, generator=torch.Generator().manual_seed(5)).to(DEV)
    st = m.initLSTMStates(torch.arange(B) + 2, [H_, W_])
    sd = {k: v.clone() for k, v in m.state_dict().items()}

    def draw(seed=None):
        m.load_state_dict(sd)              # (BatchNorm running statistics: identical encoder output for every call)
        if seed is not None:
            torch.manual_seed(seed)
        with torch.no_grad():
            y, ld, _ = m.sample(x, st)
            m.load_state_dict(sd)
            _, _, _, eps = m.forward(x, y, st, return_eps=True)
            m.load_state_dict(sd)
            y2, ld2, _ = m.reconstruct(x, st, eps)
        C.assert_field(y2, y, "reconstruct(recovered latents) == sample", atol=2e-3, rtol=1e-3)
        C.assert_logdet(ld2, ld, "log-det of sample() == log-det of reconstruct() on its latents", rtol=1e-4, atol=1e-2)
        return y, [e.float().cpu().double() for e in eps]

    y_a, eps_a = draw(seed=99)
    y_b, eps_b = draw()
    y_c, _ = draw(seed=99)
    C.assert_field(y_c, y_a, "same torch.manual_seed, same field", atol=1e-4, rtol=1e-4)   # (up to the order of the BatchNorm atomics)
    assert float((y_b - y_a).abs().max()) > 1e-2     # fresh latents per call
    assert len(eps_a) == L + 1
    flat = [e.reshape(-1) for e in eps_a]
    for lvl, e in enumerate(flat):
        n = e.numel()
        se = n ** -0.5
        assert abs(float(e.mean())) < 5 * se, (lvl, float(e.mean()), n)
        assert abs(float(e.var()) - 1.0) < 5 * 1.42 * se + 2e-3, (lvl, float(e.var()), n)          # sd of s^2 = sqrt(2 / n); + round-trip error
        assert abs(float((e ** 3).mean())) < 5 * 3.9 * se + 5e-3, (lvl, float((e ** 3).mean()))      # sd = sqrt(15 / n)
        assert abs(float((e ** 4).mean()) - 3.0) < 5 * 9.8 * se + 1e-2, (lvl, float((e ** 4).mean()))  # sd = sqrt(96 / n)

    def corr(a, b):
        k = min(a.numel(), b.numel())
        return float((a[:k] * b[:k]).mean()), k ** -0.5

    for i in range(L + 1):
        for j in range(i + 1, L + 1):                # between levels
            c, se = corr(flat[i], flat[j])
            assert abs(c) < 5 * se, ("levels", i, j, c)
        e = eps_a[i]
        for other, what in ((torch.roll(e, 1, 0), "samples"), (torch.roll(e, 1, 1), "channels"), (torch.roll(e, 1, 3), "pixels"),
                            (eps_b[i], "calls")):
            c, se = corr(flat[i], other.reshape(-1))
            assert abs(c) < 5 * se + 1e-3, (what, i, c)


def test_eval_mode_uses_running_statistics():
    """model.eval() (the reference's prediction path, utils/utils.py:162): the encoder's BatchNorm layers use their
    running moments and leave them untouched; checked against the oracle with training=False."""
    import os
    import sys
    sys.path.insert(0, os.path.join(C.ROOT, "oracle"))
    import tmglow_oracle as O
    d = C.load_npz("tiny_model.npz")
    cfg = C.CFG_TINY
    L = len(cfg["glow_blocks"])
    sd = {k: torch.from_numpy(v).clone() for k, v in C.sub(d, "sd.").items()}
    g = torch.Generator().manual_seed(5)
    for k in sd:  # non-trivial running statistics
        if k.endswith("running_mean"):
            sd[k] = 0.3 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
    m = _model(cfg, sd).eval()
    x, y = torch.from_numpy(d["x"]), torch.from_numpy(d["y"])
    h_in = [(torch.from_numpy(d["h_in.%d.h" % i]), torch.from_numpy(d["h_in.%d.c" % i])) for i in range(L)]
    P = O.params_from_state_dict(sd, requires_grad=False)
    with torch.no_grad():
        zo, lpo, ho, eo = O.tmglow_forward(P, cfg, x, y, h_in, return_eps=True, training=False)
        z, lp, h, e = m.forward(x.to(DEV), y.to(DEV), [(a.to(DEV), b.to(DEV)) for a, b in h_in], return_eps=True)
        yo, ldo, _ = O.tmglow_reconstruct(P, cfg, x, h_in, eo, training=False)
        yr, ldr, _ = m.reconstruct(x.to(DEV), [(a.to(DEV), b.to(DEV)) for a, b in h_in], [t.to(DEV) for t in eo])
    C.assert_field(z, zo.numpy(), "eval z")
    C.assert_logdet(lp, lpo.numpy())
    C.assert_field(yr, yo.numpy(), "eval reconstruct")
    C.assert_logdet(ldr, ldo.numpy())
    after = m.state_dict()
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
            assert torch.equal(after[k].cpu(), sd[k]), k


@pytest.mark.parametrize("mode", ["eager", "captured"])
def test_trainer_epoch_matches_oracle_loop(mode):
    """Rows F1+F2 together: `TrainFlow.trainParallel` (BPTT window of model.sample steps -> physics-constrained loss ->
    backward -> clip -> optimizer step -> state re-anchoring) on the HIP path against the same loop written with the CPU
    oracles; latents injected through reconstruct() so that both sides see the same noise.  captured: the window's forward passes,
    loss and backward recorded as one hipGraph and replayed (args.capture_window; the default for a repeated window shape)."""
    from types import SimpleNamespace
    import sys
    import os
    sys.path.insert(0, os.path.join(C.ROOT, "oracle"))
    import tmglow_oracle as O
    import physics_oracle as PO
    from nn.tmGlow import TMGlow
    from nn.trainFlowParallel import TrainFlow
    cfg = C.CFG_TINY3
    L = len(cfg["glow_blocks"])
    B, T, (h, w) = 2, 4, cfg["_in_hw"]
    H, W = h * cfg["_up"], w * cfg["_up"]
    C.seed_all(2468)
    m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 7, 0.03, 0.05, 0.03)
    m.out_std, m.out_mu = torch.tensor([1.3, 0.7, 2.1]), torch.tensor([0.2, -0.1, 0.4])
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, T, cfg["in_features"], h, w, generator=g)
    tgt = torch.randn(B, T, 3, H, W, generator=g)
    seeds = torch.tensor([11, 505])
    args = SimpleNamespace(beta=20.0, dx=0.05, dy=0.0625, max_grad_norm=0.25, capture_window=(mode == "captured"))

    # shapes of the latents are the model's business: take them from one oracle forward instead of trusting the formula above
    P = O.params_from_state_dict(sd)
    with torch.no_grad():
        st0 = O.init_lstm_states(cfg, seeds, [H, W])
        _, _, _, e_ref = O.tmglow_forward(P, cfg, x[:, 0], tgt[:, 0], st0, return_eps=True, training=True)
    eps = [[torch.randn(v.shape, generator=g) for v in e_ref] for _ in range(T)]

    # ---- HIP path through the trainer
    step = {"t": 0}

    eps_dev = [[e.to(DEV) for e in et] for et in eps]      # (resident before any recording: a capture admits no host-to-device copy)

    def sample_with_fixed_noise(x_t, states):
        out = m.reconstruct(x_t, states, eps_dev[step["t"] % T])       # (a recording runs the window's T steps more than once)
        step["t"] += 1
        return out

    m.sample = sample_with_fixed_noise
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    loader = [(x, tgt, seeds)]
    total = TrainFlow(args, m, loader, None).trainParallel(m, opt, tback=10, epoch=0)

    # ---- the same window with the oracles
    states = O.init_lstm_states(cfg, seeds, [H, W])
    ys, lps = [], []
    for t in range(T):
        y, lp, states = O.tmglow_reconstruct(P, cfg, x[:, t], states, eps[t], training=True)
        ys.append(y)
        lps.append(lp)
    tmean = tgt.mean(1)
    trms = torch.sqrt(((tgt - tmean.unsqueeze(1)) ** 2).mean(1))
    loss = PO.tmglow_loss(torch.stack(ys, 1), torch.stack(lps, 1), tgt, tmean, trms, torch.tensor([1.3, 0.7, 2.1]),
                          torch.tensor([0.2, -0.1, 0.4]), args.beta, args.dx, args.dy)
    tr = O.trainable(P)
    grads = torch.autograd.grad(loss, list(tr.values()), allow_unused=True)
    live = [(k, p, gr) for (k, p), gr in zip(tr.items(), grads) if gr is not None]
    gn = torch.sqrt(sum((gr.double() ** 2).sum() for _, _, gr in live)).float()
    coef = torch.clamp(args.max_grad_norm / (gn + 1e-6), max=1.0)
    assert abs(float(total) - float(loss)) <= 2e-5 * abs(float(loss)) + 2e-5, (float(total), float(loss))
    new = dict(m.named_parameters())
    worst = 0.0
    for k, p, gr in live:
        want = p.detach() - 0.05 * coef * gr
        got = new[k].detach().cpu()
        scale = float((0.05 * coef * gr).abs().max()) + 1e-12
        worst = max(worst, float((got - want).abs().max()) / scale)
    assert worst < 2e-2, worst  # updates agree to 2 % of the largest step taken in each tensor


def test_trainer_default_records_a_repeated_window_shape():
    """The trainer's DEFAULT on HIP devices (no `args.capture_window`): the first window of a shape runs eagerly, the second occurrence
    of the shape is recorded as a hipGraph (after eager passes on the default stream: their autograd graphs must not leak into the
    recording) and replayed from then on, and main.py's torch.optim.Adam is adopted as the one-launch HipAdam - same parameters after
    four mini-batches as a trainer with both switched off."""
    from types import SimpleNamespace
    import copy
    import tmg_optim
    from nn.tmGlow import TMGlow
    from nn.trainFlowParallel import TrainFlow
    cfg = C.CFG_TINY3
    B, T, (h, w) = 2, 3, cfg["_in_hw"]
    H, W = h * cfg["_up"], w * cfg["_up"]
    C.seed_all(1357)
    m0 = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m0, 7, 0.03, 0.05, 0.03)
    m0.out_std, m0.out_mu = torch.tensor([1.3, 0.7, 2.1]), torch.tensor([0.2, -0.1, 0.4])
    g = torch.Generator().manual_seed(77)
    loader = [(torch.randn(B, T, cfg["in_features"], h, w, generator=g), torch.randn(B, T, 3, H, W, generator=g), torch.tensor([3 + i, 40 + i]))
              for i in range(4)]
    with torch.no_grad():
        z, _, _, e_ref = copy.deepcopy(m0).to(DEV).forward(loader[0][0][:, 0].to(DEV), loader[0][1][:, 0].to(DEV), None, return_eps=True)
    eps_dev = [[torch.randn(v.shape, generator=g).to(DEV) for v in e_ref] for _ in range(T)]
    out = {}
    for mode in ("default", "off"):
        m = copy.deepcopy(m0).to(DEV)
        step = {"t": 0}

        def sample_with_fixed_noise(x_t, states, m=m, step=step):
            r = m.reconstruct(x_t, states, eps_dev[step["t"] % T])
            step["t"] += 1
            return r

        m.sample = sample_with_fixed_noise
        args = SimpleNamespace(beta=20.0, dx=0.05, dy=0.0625, max_grad_norm=0.25)
        if mode == "off":
            args.capture_window = False
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
        tr = TrainFlow(args, m, loader, None)
        if mode == "off":
            tr.use_hip_adam = False
        total = tr.trainParallel(m, opt, epoch=0)
        out[mode] = (float(total), {k: v.detach().clone() for k, v in m.named_parameters()}, tr, opt)
    trd, optd = out["default"][2], out["default"][3]
    assert trd._capture == "auto" and len(trd._captured) == 1 and not trd._capture_failed
    assert next(iter(trd._captured.values())).replays == 3           # mini-batch 1 eager, 2 recorded + replayed, 3 and 4 replayed
    assert isinstance(optd, tmg_optim.HipAdam) and type(out["off"][3]) is torch.optim.Adam and not out["off"][2]._captured
    assert abs(out["default"][0] - out["off"][0]) <= 1e-5 * abs(out["off"][0]) + 1e-5, (out["default"][0], out["off"][0])
    for k, v in out["off"][1].items():
        step_size = 4 * 1e-3                    # Adam: |update| <= lr per step
        assert float((out["default"][1][k] - v).abs().max()) <= 2e-2 * step_size, k


def test_trainer_failed_recording_leaves_the_eager_trajectory(monkeypatch):
    """the default trainer records a repeated window shape and falls back to eager windows when the recording fails - the
    failed construction must leave NOTHING behind.  The recording pass is made to raise after it has run the window's body for real
    (BatchNorm statistics advanced, device generator advanced, gradients bound): afterwards buffers, generator state and gradients are
    what they were, the trainer notes the failure, and four mini-batches end on the parameters and BatchNorm buffers of a trainer that
    never tried (same optimizer on both sides)."""
    from types import SimpleNamespace
    import copy
    import tmg_dist
    from nn.tmGlow import TMGlow
    from nn.trainFlowParallel import TrainFlow
    cfg = C.CFG_TINY3
    B, T, (h, w) = 2, 3, cfg["_in_hw"]
    H, W = h * cfg["_up"], w * cfg["_up"]
    C.seed_all(1357)
    m0 = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m0, 7, 0.03, 0.05, 0.03)
    m0.out_std, m0.out_mu = torch.tensor([1.3, 0.7, 2.1]), torch.tensor([0.2, -0.1, 0.4])
    g = torch.Generator().manual_seed(78)
    loader = [(torch.randn(B, T, cfg["in_features"], h, w, generator=g), torch.randn(B, T, 3, H, W, generator=g), torch.tensor([3 + i, 40 + i]))
              for i in range(4)]

    # ---- the constructor on its own: state before == state after a failed recording
    m = copy.deepcopy(m0).to(DEV)
    x0 = loader[0][0][:, 0].to(DEV)
    st = m.initLSTMStates(loader[0][2], [H, W])
    marks = [torch.full_like(p, 0.5) for p in m.parameters()]
    for p, mk in zip(m.parameters(), marks):
        p.grad = mk
    buf0 = {k: v.detach().clone() for k, v in m.named_buffers()}
    rng0 = torch.cuda.get_rng_state(DEV)
    ran = {"n": 0}

    def failing_record(run):
        out = run()                      # the body really runs inside the capture: statistics / generator / gradients all move
        ran["n"] += 1
        raise RuntimeError("forced recording failure")

    monkeypatch.setattr(tmg_dist.CapturedWindow, "_record", staticmethod(failing_record))

    def body(x, states):
        y, ld, states = m.sample(x, states)             # device-drawn latents: advances the Philox offset
        return C.loss_reverse(y, ld), (states,)

    with pytest.raises(RuntimeError, match="forced recording failure"):
        tmg_dist.CapturedWindow(m, body, (x0, st))
    torch.cuda.synchronize()
    assert ran["n"] == 1
    assert torch.equal(torch.cuda.get_rng_state(DEV), rng0)
    for k, v in m.named_buffers():
        assert torch.equal(v, buf0[k]), k
    for p, mk in zip(m.parameters(), marks):
        assert p.grad is mk

    # ---- through the trainer: the fallback continues on the eager trajectory
    e_model = copy.deepcopy(m0).to(DEV)
    with torch.no_grad():
        _, _, _, e_ref = e_model.forward(loader[0][0][:, 0].to(DEV), loader[0][1][:, 0].to(DEV), None, return_eps=True)
    eps_dev = [[torch.randn(v.shape, generator=g).to(DEV) for v in e_ref] for _ in range(T)]
    out = {}
    for mode in ("default", "off"):
        m = copy.deepcopy(m0).to(DEV)
        step = {"t": 0}

        def sample_with_fixed_noise(x_t, states, m=m, step=step):
            r = m.reconstruct(x_t, states, eps_dev[step["t"] % T])
            step["t"] += 1
            return r

        m.sample = sample_with_fixed_noise
        args = SimpleNamespace(beta=20.0, dx=0.05, dy=0.0625, max_grad_norm=0.25)
        if mode == "off":
            args.capture_window = False
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-8, amsgrad=True)
        tr = TrainFlow(args, m, loader, None)
        tr.use_hip_adam = False
        total = tr.trainParallel(m, opt, epoch=0)
        out[mode] = (float(total), {k: v.detach().clone() for k, v in m.state_dict().items()}, tr)
    trd = out["default"][2]
    assert trd._capture == "auto" and not trd._captured and len(trd._capture_failed) == 1
    assert "forced recording failure" in next(iter(trd._capture_failed.values()))
    assert ran["n"] == 2                                  # tried once for the shape, never again
    assert abs(out["default"][0] - out["off"][0]) <= 1e-5 * abs(out["off"][0]) + 1e-5, (out["default"][0], out["off"][0])
    for k, v in out["off"][1].items():
        got = out["default"][1][k]
        if k.endswith("num_batches_tracked"):
            assert torch.equal(got, v), k                 # a leaked statistics update would count twice
        elif "running_" in k:
            # two independent trainings: the float atomics' summation order differs, Adam turns that into parameter differences (the
            # bound below) and the statistics of the layers downstream follow them (measured: up to 2e-5 of the largest entry).  A
            # leaked update would move an entry by momentum * 0.9^12 = 3e-2 of it
            assert float((got - v).abs().max()) <= 3e-4 * float(v.abs().max()) + 1e-6, k
        elif v.dtype.is_floating_point and "log_s_old" not in k:
            assert float((got - v).abs().max()) <= 2e-2 * 4e-3, k


def test_trainer_test_loop_error_measure():
    """`TrainFlow.test` (reference trainFlowParallel.py:313-382): un-normalisation with out_std / out_mu, sample mean over the
    roll-outs, squared error summed over time-steps 1..tmax (step 0 excluded) and divided by ntest * tmax * H * W - checked
    against the closed form for a model whose `sample` is replaced by a known function of the time-step; then one un-patched run
    on the HIP path (finite, states re-anchored every 10 steps without error)."""
    from types import SimpleNamespace
    from nn.tmGlow import TMGlow
    from nn.trainFlowParallel import TrainFlow
    cfg = C.CFG_TINY3
    B, T, (h, w) = 2, 13, cfg["_in_hw"]
    Hh, Ww = h * cfg["_up"], w * cfg["_up"]
    C.seed_all(11)
    m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 7, 0.03, 0.05, 0.03)
    m.out_std, m.out_mu = torch.tensor([1.5, 0.5, 2.0]), torch.tensor([0.1, -0.2, 0.3])
    m = m.to(DEV)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, T, cfg["in_features"], h, w, generator=g)
    tgt = torch.randn(B, T, 3, Hh, Ww, generator=g)
    args = SimpleNamespace(beta=1.0, dx=0.1, dy=0.1, max_grad_norm=1.0, ntest=B, device=torch.device(DEV))
    trainer = TrainFlow(args, m, None, [(x, tgt, torch.ones(B))])
    # (i) closed form: the "model" predicts target + 0.25 * (t + 1) in normalised units
    calls = {"t": 0}
    real_sample = m.sample

    def fake(x_t, states):
        t = calls["t"] % T
        calls["t"] += 1
        return tgt[:, t].to(DEV) + 0.25 * (t + 1), torch.zeros(B, device=DEV), states

    m.sample = fake
    mse = trainer.test(m, samples=2, epoch=0, plot=False, tmax=T - 1)
    sd = torch.tensor([1.5, 0.5, 2.0])
    want = sum(float(((sd * 0.25 * (t + 1)) ** 2).sum()) * Hh * Ww * B for t in range(1, T)) / (B * (T - 1) * Hh * Ww)
    assert abs(float(mse) - want) <= 1e-5 * want, (float(mse), want)
    assert calls["t"] == 2 * T and m.training      # every step of both roll-outs; the model is put back into its previous mode
    # (ii) the real path
    m.sample = real_sample
    mse = trainer.test(m, samples=1, epoch=0, plot=False, tmax=T - 1)
    assert torch.isfinite(mse) and float(mse) > 0


def test_model_pred_rollout_layout_and_scaling():
    """`utils.utils.modelPred` (reference utils/utils.py:151-235): shapes of the three returned tensors, every `stride`-th step
    kept, un-normalisation and the inlet-velocity scaling (u0, u0, u0^2) - against the closed form for a patched `sample`,
    then one un-patched roll-out on the HIP path."""
    from types import SimpleNamespace
    from nn.tmGlow import TMGlow
    from utils.utils import modelPred
    cfg = C.CFG_TINY3
    B, T, (h, w) = 3, 6, cfg["_in_hw"]
    Hh, Ww = h * cfg["_up"], w * cfg["_up"]
    C.seed_all(12)
    m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 8, 0.03, 0.05, 0.03)
    m.out_std, m.out_mu = torch.tensor([1.5, 0.5, 2.0]), torch.tensor([0.1, -0.2, 0.3])
    m.in_std, m.in_mu = torch.tensor([2.0, 3.0, 0.5]), torch.tensor([0.5, 0.0, -1.0])
    m = m.to(DEV)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, cfg["in_features"], h, w, generator=g)
    tgt = torch.randn(B, T, 3, Hh, Ww, generator=g)
    u0 = torch.tensor([1.0, 2.0, 0.5])
    loader = [(x[:2], tgt[:2], u0[:2]), (x[2:], tgt[2:], u0[2:])]
    log = SimpleNamespace(log=lambda *a, **k: None)
    args = SimpleNamespace(device=torch.device(DEV))
    calls = {"t": 0}
    real_sample = m.sample

    def fake(x_t, states):
        t = calls["t"] % 4
        calls["t"] += 1
        return torch.full((x_t.size(0), 3, Hh, Ww), float(t), device=DEV), torch.zeros(x_t.size(0), device=DEV), states

    m.sample = fake
    yp, yt, yi = modelPred(args, m, loader, log, samples=2, stride=2, tmax=4)
    assert tuple(yp.shape) == (2, B, 2, 3, Hh, Ww) and tuple(yt.shape) == (B, T, 3, Hh, Ww) and tuple(yi.shape) == (B, T, 3, h, w)
    sc = torch.stack([u0, u0, u0 ** 2], 1)                                       # [B,3]
    sd, mu = torch.tensor([1.5, 0.5, 2.0]), torch.tensor([0.1, -0.2, 0.3])
    for k, t in enumerate((0, 2)):                                               # kept steps
        want = (sc * (sd * t + mu)).view(1, B, 3, 1, 1).expand(2, B, 3, Hh, Ww)
        assert torch.allclose(yp[:, :, k], want, rtol=1e-6, atol=1e-6)
    assert torch.allclose(yt, sc.view(B, 1, 3, 1, 1) * (sd.view(1, 1, 3, 1, 1) * tgt + mu.view(1, 1, 3, 1, 1)), rtol=1e-5, atol=1e-6)
    isd, imu = torch.tensor([2.0, 3.0, 0.5]), torch.tensor([0.5, 0.0, -1.0])
    assert torch.allclose(yi, sc.view(B, 1, 3, 1, 1) * (isd.view(1, 1, 3, 1, 1) * x[:, :, :3] + imu.view(1, 1, 3, 1, 1)), rtol=1e-5, atol=1e-6)
    m.sample = real_sample
    yp, _, _ = modelPred(args, m, loader, log, samples=1, stride=1, tmax=3)
    assert tuple(yp.shape) == (1, B, 3, 3, Hh, Ww) and torch.isfinite(yp).all() and not (yp > 9999).any()


def test_encoder_dropout_option():
    """`--drop-rate` (reference main.py:71 -> denseBlock.py:54-55, tmGlow.py:183-184: nn.Dropout3d after the encoder's dense-layer
    and transition convs).  Evaluation mode is the rate-0 model exactly; training mode runs forward + backward on the HIP path with
    the mask torch draws for an un-batched 4-D input (whole samples of a feature map dropped, survivors scaled by 1 / (1 - p))."""
    from nn.tmGlow import TMGlow
    cfg = C.CFG_TINY3
    (h, w) = cfg["_in_hw"]
    C.seed_all(13)
    kw = C.build_kwargs(cfg)
    m0 = TMGlow(**kw)
    C.perturb_(m0, 9, 0.03, 0.05, 0.03)
    m1 = TMGlow(**dict(kw, drop_rate=0.5))
    m1.load_state_dict(m0.state_dict(), strict=True)     # dropout adds no parameters or buffers
    m0, m1 = m0.to(DEV).eval(), m1.to(DEV).eval()
    assert any(isinstance(mod, torch.nn.Dropout3d) for mod in m1.modules())
    B = 4
    x = torch.randn(B, cfg["in_features"], h, w, generator=torch.Generator().manual_seed(6)).to(DEV)
    st = m0.initLSTMStates(torch.arange(B), [h * cfg["_up"], w * cfg["_up"]])
    torch.manual_seed(3)
    y0, ld0, _ = m0.sample(x, st)
    torch.manual_seed(3)
    y1, ld1, _ = m1.sample(x, st)
    # (the per-sample log-det is accumulated with float atomics: equal up to the order of the additions)
    assert torch.equal(y0, y1) and torch.allclose(ld0, ld1, rtol=2e-6, atol=0)
    m1.train()
    # the mask semantics on one dense layer: each sample's new channels are either zero or 2x the rate-0 values
    layer1 = m1.encoder.encoding_blocks[0][-1].denselayer1
    layer0 = m0.train().encoder.encoding_blocks[0][-1].denselayer1
    a = torch.randn(B, 6, 6, layer0.norm1.num_features, device=DEV)
    ref = layer0.grow(a)
    out = layer1.grow(a)
    for b in range(B):
        assert bool((out[b] == 0).all()) or torch.allclose(out[b], 2.0 * ref[b], rtol=1e-5, atol=1e-6)
    y, ld, _ = m1.sample(x, st)
    C.loss_reverse(y, ld).backward()
    assert torch.isfinite(y).all() and all(p.grad is None or torch.isfinite(p.grad).all() for p in m1.parameters())


@pytest.mark.parametrize("reverse,ts", [(True, True), (False, False), (False, True)])
def test_lu_fold_kernels_match_torch_folding(reverse, ts):
    """tmg_lu_fold_fwd / _bwd (ActNorm + PLU folding of a whole level in two launches) against the same folding written with
    differentiable torch ops in fp64 (the reference's arithmetic: glowConv.py:151-161, actNorm.py:66-83): mix matrices, biases,
    log-det and every parameter gradient, for the direction that applies W (reverse with train_sampling, the one training uses)
    and - through the torch path the module keeps for it - the inverse direction."""
    import os
    from nn.tmGlow import TMGlow
    cfg = C.CFG_TINY3
    C.seed_all(21)
    m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 11, 0.03, 0.2, 0.05)
    m = m.to(DEV)
    g = torch.Generator().manual_seed(8)
    for blk in m.glow.flow_blocks:
        for layer in blk.revlayers._modules.values():
            layer.conv.train_sampling = ts      # (False, False): W is applied in the density direction -> the kernels' forward fold
        names = [n for n, _ in blk.named_parameters() if n.endswith((".l", ".u", ".log_s", "norm.weight", "norm.bias"))]
        res = {}
        for mode in ("hip", "split", "torch"):
            os.environ.pop("TMG_NO_LU_FOLD_KERNEL", None)
            if mode == "torch":
                os.environ["TMG_NO_LU_FOLD_KERNEL"] = "1"
            blk.zero_grad()
            lm = blk._level_mix(reverse, 37)
            Wm, bm, ld = lm[:3]
            if mode == "hip":
                gW = torch.randn(Wm.shape, generator=g).to(DEV)
                gb = torch.randn(bm.shape, generator=g).to(DEV)
            if mode == "split":
                # the (first K-1 layers, last layer) views of the same node, whose gradients the backward launch reads in place
                if len(lm) < 4:
                    continue
                Wh, bh, Wt, bt = lm[3]
                ((Wh * gW[:-1]).sum() + (Wt * gW[-1]).sum() + (bh * gb[:-1]).sum() + (bt * gb[-1]).sum() + 0.3 * ld).backward()
            else:
                ((Wm * gW).sum() + (bm * gb).sum() + 0.3 * ld).backward()
            res[mode] = (Wm.detach().clone(), bm.detach().clone(), ld.detach().clone(), {n: p.grad.detach().clone() for n, p in blk.named_parameters() if n in names})
        os.environ.pop("TMG_NO_LU_FOLD_KERNEL", None)
        if "split" in res:
            for n in res["hip"][3]:
                assert torch.equal(res["hip"][3][n], res["split"][3][n]), n
        a, b = res["hip"], res["torch"]
        for x, y, what in ((a[0], b[0], "Wm"), (a[1], b[1], "bm"), (a[2], b[2], "ld")):
            assert float((x - y).abs().max()) <= 2e-5 * max(float(y.abs().max()), 1e-6), what
        assert set(a[3]) == set(b[3]) and len(a[3]) >= 5
        for n in a[3]:
            sc = max(float(b[3][n].abs().max()), 1e-6)
            assert float((a[3][n] - b[3][n]).abs().max()) <= 5e-5 * sc, (n, float((a[3][n] - b[3][n]).abs().max()), sc)


def test_lu_fold_cache_follows_reloaded_permutations():
    """The level's cached row permutations / sign vectors (derived VALUES of the buffers `p` and `sign_s`) must follow a
    `load_state_dict` that copies a different permutation and signs into those buffers IN PLACE (same addresses): forward with
    the first weights, load a model initialised from another seed, and compare the level output with the torch folding
    (TMG_NO_LU_FOLD_KERNEL=1), which reads the buffers on every call."""
    import os
    from nn.modules.flowLSTMBlock import LSTMFLowBlock

    def make(seed):
        C.seed_all(seed)
        b = LSTMFLowBlock(4, 8, 8, 4, LUdecompose=True, train_sampling=True, do_split=False, squeeze_type=0)
        C.perturb_(b, seed, 0.05, 0.1, 0.05)
        return b

    blk, other = make(5).to(DEV), make(6)
    sd_other = {k: v.clone() for k, v in other.state_dict().items()}
    assert any(not torch.equal(sd_other[k], v.cpu()) for k, v in blk.state_dict().items() if k.endswith(".p"))   # another permutation
    g = torch.Generator().manual_seed(3)
    z = torch.randn(2, 16, 8, 8, generator=g).to(DEV)
    cond = torch.randn(2, 8, 8, 8, generator=g).to(DEV)
    with torch.no_grad():
        blk.reverse(z, cond, None)                      # fills the cache with the first model's permutations
        blk.load_state_dict(sd_other)                   # in-place copies: the buffers keep their addresses
        x_hip, ld_hip, _ = blk.reverse(z, cond, None)
        os.environ["TMG_NO_LU_FOLD_KERNEL"] = "1"
        try:
            x_ref, ld_ref, _ = blk.reverse(z, cond, None)
        finally:
            os.environ.pop("TMG_NO_LU_FOLD_KERNEL", None)
    C.assert_field(x_hip, x_ref, "level output after reloading other permutations", atol=2e-5, rtol=1e-5)
    C.assert_logdet(ld_hip, ld_ref, rtol=2e-6, atol=1e-3)


def test_bptt_window_at_stated_batch_matches_oracle():
    """Two time-steps of a BPTT window (trainFlowParallel.py:256-287: the recurrent states of step 0 feed step 1 and carry a
    gradient back) at BASELINE configs[1]'s stated batch 32 - 3-channel fields, i.e. the zero-padded channel layout on the first
    level - with the loss on two samples: the gate conv's input gradient then covers all Cin + 64 channels and the cell state's
    gradient is live, paths the single-step cases never take.  Oracle as in _stated_batch_case (encoder on the whole batch per
    step, flow on the two samples), fp64 truth and fp32 yardstick, all parameter gradients."""
    import contextlib
    import io
    import os
    import sys
    sys.path.insert(0, os.path.join(C.ROOT, "oracle"))
    import tmglow_oracle as O
    from nn.tmGlow import TMGlow
    name, cfg, B, n, T = "cfg2", C.CFG2, 32, 2, 2
    C.seed_all(12345)
    with contextlib.redirect_stdout(io.StringIO()):
        m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 7, *C.perturb_scales(cfg))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(DEV).train()
    h, w = cfg["_in_hw"]
    H_, W_ = h * cfg["_up"], w * cfg["_up"]
    g = torch.Generator().manual_seed(78)
    xs = [torch.randn(B, cfg["in_features"], h, w, generator=g) for _ in range(T)]
    y = torch.randn(B, cfg["out_features"], H_, W_, generator=g)
    seeds = torch.arange(B) + 5
    st0 = m.initLSTMStates(seeds, [H_, W_])
    with torch.no_grad():
        _, _, _, e_ = m.forward(xs[0].to(DEV), y.to(DEV), st0, return_eps=True)
        m.load_state_dict(sd)
    eps = [[torch.randn(e.shape, generator=g) for e in e_] for _ in range(T)]
    # ---- HIP
    m.zero_grad()
    st, loss = st0, 0.0
    for t in range(T):
        yt, ldt, st = m.reconstruct(xs[t].to(DEV), st, [e.to(DEV) for e in eps[t]])
        loss = loss + C.loss_reverse(yt[:n], ldt[:n])
    loss.backward()
    gr = {k: v.detach().clone() for k, v in _grads(m).items()}
    # ---- oracle
    res = {}
    for dt in (torch.float64, torch.float32):
        P = O.params_from_state_dict(sd, dtype=dt)
        sto = [(a[:n].to(dt), b[:n].to(dt)) for a, b in O.init_lstm_states(cfg, seeds[:n], [H_, W_])]
        lo = 0.0
        if dt == torch.float64:
            kp = C.KinkProbe().__enter__()
        for t in range(T):
            z_out, c_out = O.encoder(P, cfg, xs[t].to(dt), True)
            cmean, clsd = z_out[:n].chunk(2, 1)
            clsd = clsd.clamp(-10.0, O.LOG5)
            z = cmean + torch.exp(clsd) * eps[t][-1][:n].to(dt)
            yo, ldo, sto = O.decoder_reverse(P, cfg, z, [c[:n] for c in c_out], sto, [e[:n].to(dt) for e in eps[t][:-1]])
            lo = lo + C.loss_reverse(yo, ldo)
        lo.backward(retain_graph=(dt == torch.float64))
        res[dt] = dict(y=yo.detach(), loss=float(lo.detach()), g={k: v.grad.clone() for k, v in O.trainable(P).items() if v.grad is not None})
        if dt == torch.float64:
            kp.__exit__(None, None, None)
            res[dt]["kink"] = kp.allowances(O.trainable(P), res[dt]["g"])
    r64, r32 = res[torch.float64], res[torch.float32]
    C.assert_field(yt[:n], r64["y"], "y of the second step", atol=max(C.FIELD_ATOL, YARDSTICK * _maxabs(r32["y"], r64["y"])))
    assert abs(float(loss) - r64["loss"]) <= 1e-5 * abs(r64["loss"]) + 1e-6
    fl = _grad_err(r32["g"], r64["g"])
    er = _grad_err(gr, r64["g"])
    print("\nBPTT window %s at batch %d: hip vs fp64 %s, fp32 oracle vs fp64 %s" % (name, B, er, fl))
    assert set(gr) == set(r64["g"])
    C.assert_grads(gr, r64["g"], "two-step window grads at the stated batch", global_tol=max(C.GRAD_GLOBAL_REL_L2, GRAD_YARDSTICK * fl[0]),
                   tensor_tol=max(C.GRAD_TENSOR_REL_MAX, GRAD_YARDSTICK * fl[1]), kink=r64["kink"])


@pytest.mark.parametrize("cin,hw,B", [(4, (64, 64), 8), (8, (17, 33), 4), (3, (30, 30), 4), (16, (16, 16), 8)])
def test_recompute_backward_matches_stored_activations_level(cin, hw, B):
    """tmg_ops.set_recompute (round 5): the level node of a narrow flow level keeps NO per-layer activations and its backward rebuilds
    every layer's input from its output (inverse channel mix -> growth layers -> x2 = (y2 + shift) e^{sg}; flowAffine.py:85-109,
    flowLSTMBlock.py:323-361) - same forward values bit for bit, gradients within the strict path-vs-path bounds; C = 8, 16, the
    zero-padded 12 and 32; the 64-channel level (cin = 16 -> C = 64 is not on the fused path) keeps its activations: identical."""
    import tmg_ops as ops
    from nn.modules.flowLSTMBlock import LSTMFLowBlock
    hs, ws = hw
    C.seed_all(cin * 13 + B)
    blk = LSTMFLowBlock(cin, 32, 64, 6, LUdecompose=True, train_sampling=True, do_split=True, squeeze_type=0)
    C.perturb_(blk, 5, 0.02, 0.05, 0.02)
    blk.to(DEV)
    g = torch.Generator().manual_seed(21)
    z, eps = (torch.randn(B, 2 * cin, hs, ws, generator=g).to(DEV) for _ in range(2))
    cond = torch.randn(B, 32, hs, ws, generator=g).to(DEV)
    hst, cst = (torch.randn(B, 64, hs, ws, generator=g).to(DEV) for _ in range(2))
    res = {}
    try:
        for tag in ("stored", "recompute"):
            ops.set_recompute(tag == "recompute")
            blk.zero_grad()
            zi, ci, hi, cc = (t.clone().requires_grad_(True) for t in (z, cond, hst, cst))
            xr, ldr, st = blk.reverse(zi, ci, (hi, cc), eps=eps)
            ((xr[:2] ** 2).sum() * 0.5 + ldr[:2].sum() * 0.02 + (st[0][:2] ** 2).sum() * 0.1).backward()
            gr = {k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None}
            gr.update({"@dz": zi.grad.clone(), "@dcond": ci.grad.clone(), "@dh": hi.grad.clone(), "@dc": cc.grad.clone()})
            res[tag] = (xr.detach(), ldr.detach(), gr)
    finally:
        ops.set_recompute(False)
    a, b = res["recompute"], res["stored"]
    assert torch.equal(a[0], b[0])          # the forward pass is the same launches (the log-det sums are float atomics: last bits vary)
    C.assert_logdet(a[1], b[1], rtol=2e-6, atol=1e-3)
    C.assert_grads(a[2], b[2], "recompute vs stored activations", global_tol=2e-4, tensor_tol=5e-3)


@pytest.mark.parametrize("case", ["tiny", "tiny3", "M64"])
def test_recompute_mode_whole_model(case):
    """The whole model in recompute mode against the default path: generative direction (reconstruct with fixed latents) + loss +
    backward; tiny / tiny3 from the reference fixtures, and the metric configuration at its stated batch 64, where the mode's point -
    the activation memory of a step - is asserted as well."""
    import tmg_ops as ops
    from nn.tmGlow import TMGlow
    if case == "M64":
        cfg, B = C.CFG_M, 64
        C.seed_all(12345)
        m = TMGlow(**C.build_kwargs(cfg))
        C.perturb_(m, 7, *C.perturb_scales(cfg))
        m.to(DEV).train()
        g = torch.Generator().manual_seed(3)
        Hin, Win = cfg["_in_hw"]
        x = torch.randn(B, cfg["in_features"], Hin, Win, generator=g).to(DEV)
        yt = torch.randn(B, cfg["out_features"], Hin * cfg["_up"], Win * cfg["_up"], generator=g).to(DEV)
        st = m.initLSTMStates(torch.arange(B), [Hin * cfg["_up"], Win * cfg["_up"]])
        with torch.no_grad():
            _, _, _, eps = m.forward(x, yt, st, return_eps=True)
        eps = [e.detach() for e in eps]
        del yt
    else:
        name, cfg = ("tiny_model.npz", C.CFG_TINY) if case == "tiny" else ("tiny3_model.npz", C.CFG_TINY3)
        d = C.load_npz(name)
        L = len(cfg["glow_blocks"])
        m = _model(cfg, {k: torch.from_numpy(v) for k, v in C.sub(d, "sd.").items()})
        x = torch.from_numpy(d["x"]).to(DEV)
        st = C.states_from(d, "h_in.", L, DEV)
        eps = [torch.from_numpy(d["fwd.eps.%d" % i]).to(DEV) for i in range(L + 1)]
    res, peak = {}, {}
    try:
        for tag in ("stored", "recompute"):
            ops.set_recompute(tag == "recompute")
            m.zero_grad()
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            y, ld, _ = m.reconstruct(x, st, eps)
            held = torch.cuda.memory_allocated() - base          # what the forward pass keeps alive for backward
            C.loss_reverse(y, ld).backward()
            torch.cuda.synchronize()
            peak[tag] = (held / 2 ** 30, (torch.cuda.max_memory_allocated() - base) / 2 ** 30)
            res[tag] = (y.detach(), ld.detach(), {k: v.clone() for k, v in _grads(m).items()})
            del y, ld
    finally:
        ops.set_recompute(False)
    a, b = res["recompute"], res["stored"]
    C.assert_field(a[0], b[0], "y", atol=1e-6, rtol=1e-6)       # (same launches; BatchNorm / log-det sums are float atomics)
    C.assert_logdet(a[1], b[1], rtol=2e-6, atol=1e-3)
    ge = _grad_err(a[2], b[2])
    print("\nrecompute mode, %s: gradients vs the stored-activation path global rel-L2 %.2e, worst tensor %.2e; forward keeps %.2f GB (stored: %.2f), "
          "peak %.2f GB (stored: %.2f)" % (case, ge[0], ge[1], peak["recompute"][0], peak["stored"][0], peak["recompute"][1], peak["stored"][1]))
    C.assert_grads(a[2], b[2], "recompute vs stored activations (%s)" % case, global_tol=2e-4, tensor_tol=5e-3)
    if case == "M64":
        import json
        import os
        out = os.path.join(C.ROOT, "gpurun_out")
        if os.path.isdir(out):
            with open(os.path.join(out, "recompute_M_batch64.json"), "w") as f:
                json.dump({"case": "config M, batch 64, one generative step + backward", "grads_vs_stored_global_rel_l2": ge[0],
                           "grads_vs_stored_worst_tensor_rel_max": ge[1], "forward_keeps_gb": {k: v[0] for k, v in peak.items()},
                           "peak_gb": {k: v[1] for k, v in peak.items()}}, f, indent=1)
        # the plain coupling layers of the 16- and 32-channel levels hold ~3.3 of the ~7 GB a step keeps at this configuration
        assert peak["recompute"][0] < 0.65 * peak["stored"][0], peak


@pytest.mark.parametrize("cin,hw,B", [(8, (32, 32), 8), (16, (12, 20), 4)])
def test_level_with_bf16x3_winograd_matches_fp32_winograd(cin, hw, B):
    """The opt-in bf16x3 arithmetic of the wide Winograd contractions (tmg_ops.set_winograd_precision: three-way exact bf16 split of
    both operands, six part products, fp32 accumulation - the ConvLSTM gate conv and the level-wide conditioning conv, forward) against
    the default fp32-MFMA path on one flow level, generative direction + backward, with the STRICT path-vs-path bounds of
    test_level_kernels_across_field_and_batch_sizes: the two differ by fp32 rounding only."""
    import tmg_ops as ops
    from nn.modules.flowLSTMBlock import LSTMFLowBlock
    hs, ws = hw
    C.seed_all(cin * 11 + B)
    blk = LSTMFLowBlock(cin, 32, 64, 6, LUdecompose=True, train_sampling=True, do_split=True, squeeze_type=0)
    C.perturb_(blk, 5, 0.02, 0.05, 0.02)
    blk.to(DEV)
    g = torch.Generator().manual_seed(13)
    z, eps = (torch.randn(B, 2 * cin, hs, ws, generator=g).to(DEV) for _ in range(2))
    cond = torch.randn(B, 32, hs, ws, generator=g).to(DEV)
    hst, cst = (torch.randn(B, 64, hs, ws, generator=g).to(DEV) for _ in range(2))
    res = {}
    try:
        for tag in ("f32", "bf16x3"):
            ops.set_winograd_precision(tag)
            assert ops.winograd_precision() == tag
            blk.zero_grad()
            zi, ci, hi, cc = (t.clone().requires_grad_(True) for t in (z, cond, hst, cst))
            xr, ldr, st = blk.reverse(zi, ci, (hi, cc), eps=eps)
            ((xr[:2] ** 2).sum() * 0.5 + ldr[:2].sum() * 0.02 + (st[0][:2] ** 2).sum() * 0.1).backward()
            gr = {k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None}
            gr.update({"@dz": zi.grad.clone(), "@dcond": ci.grad.clone(), "@dh": hi.grad.clone(), "@dc": cc.grad.clone()})
            res[tag] = (xr.detach(), ldr.detach(), st[0].detach(), gr)
    finally:
        ops.set_winograd_precision("f32")
    a, b = res["bf16x3"], res["f32"]
    assert float((a[2] - b[2]).abs().max()) > 0.0, "the bf16x3 kernel must actually run (the gate conv feeds the new state)"
    C.assert_field(a[0], b[0], "level output", atol=1e-4 * float(b[0].abs().max()), rtol=1e-5)
    C.assert_field(a[2], b[2], "new hidden state", atol=1e-5)
    C.assert_logdet(a[1], b[1], rtol=5e-6, atol=1e-3)
    C.assert_grads(a[3], b[3], "bf16x3 vs fp32 Winograd", global_tol=2e-4, tensor_tol=5e-3)


@pytest.mark.parametrize("cin,hw,B", [(4, (64, 64), 64), (8, (17, 33), 32), (16, (16, 16), 128), (16, (9, 23), 64), (32, (8, 16), 64), (3, (30, 30), 64)])
def test_level_kernels_across_field_and_batch_sizes(cin, hw, B):
    """One flow level (generative direction + backward, recurrent states with gradients, loss on two samples) through the level-fused
    node and its fused / grouped kernels against the per-layer path on the general kernels (TMG_NO_LEVEL_FUSION=1), over the model's
    channel widths (16 .. 128, and the padded 12), ragged fields and batch sizes up to 128: the launch plans of the persistent, grouped
    and fused kernels depend on the pixel count (tools/level_sweep.py is the long form)."""
    import os
    from nn.modules.flowLSTMBlock import LSTMFLowBlock
    hs, ws = hw
    C.seed_all(cin * 7 + B)
    blk = LSTMFLowBlock(cin, 32, 64, 6, LUdecompose=True, train_sampling=True, do_split=True, squeeze_type=0)
    C.perturb_(blk, 5, 0.02, 0.05, 0.02)
    blk.to(DEV)
    g = torch.Generator().manual_seed(9)
    z, eps = (torch.randn(B, 2 * cin, hs, ws, generator=g).to(DEV) for _ in range(2))
    cond = torch.randn(B, 32, hs, ws, generator=g).to(DEV)
    hst, cst = (torch.randn(B, 64, hs, ws, generator=g).to(DEV) for _ in range(2))
    res = {}
    for tag, env in (("fused", None), ("plain", "1")):
        if env:
            os.environ["TMG_NO_LEVEL_FUSION"] = env
        try:
            blk.zero_grad()
            zi, ci, hi, cc = (t.clone().requires_grad_(True) for t in (z, cond, hst, cst))
            xr, ldr, st = blk.reverse(zi, ci, (hi, cc), eps=eps)
            ((xr[:2] ** 2).sum() * 0.5 + ldr[:2].sum() * 0.02 + (st[0][:2] ** 2).sum() * 0.1).backward()
        finally:
            os.environ.pop("TMG_NO_LEVEL_FUSION", None)
        gr = {k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None}
        gr.update({"@dz": zi.grad.clone(), "@dcond": ci.grad.clone(), "@dh": hi.grad.clone(), "@dc": cc.grad.clone()})
        res[tag] = (xr.detach(), ldr.detach(), gr)
    a, b = res["fused"], res["plain"]
    C.assert_field(a[0], b[0], "level output", atol=1e-4 * float(b[0].abs().max()), rtol=1e-5)
    C.assert_logdet(a[1], b[1], rtol=5e-6, atol=1e-3)
    # (two fp32 evaluation orders flip different near-zero ReLUs: a local difference in @dz of a per cent of its scale, nothing global)
    C.assert_grads(a[2], b[2], "level-fused vs per-layer path", global_tol=2e-4, tensor_tol=5e-3)
