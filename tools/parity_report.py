#!/usr/bin/env python3
"""Per-tensor parity report of the generative direction's gradients.

For one configuration the CPU oracle is evaluated once in fp64 (the truth) and once in fp32 (the reference's own arithmetic:
the yardstick); the HIP path is then run once per VARIANT (a set of environment switches that select other kernels for the
same arithmetic) in a child process each, and every gradient tensor is compared with the fp64 truth.  The report lists, per
variant and direction, the global rel-L2, the worst per-tensor rel-max and the five worst tensors by either measure, next
to the fp32 oracle's error on the same tensors.

  python tools/parity_report.py --config M --batch 1 --out gpurun_out/parity_report_M.json
  python tools/parity_report.py --config cfg3 --variants default,no_wino,no_wino_wgrad,no_fused_bwd

The oracle is used here as the CHECKER only (tools/ is diagnostics, like tests/)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import common as C  # noqa: E402

VARIANTS = {
    "default": {},
    "no_wino": {"TMG_NO_WINOGRAD": "1"},
    "no_wino_wgrad": {"TMG_NO_WINOGRAD_WGRAD": "1"},
    "no_fused_bwd": {"TMG_NO_FUSED_COUPLING_BWD": "1"},
    "no_fused": {"TMG_NO_FUSED_COUPLING": "1"},
    "no_grouped_wgrad": {"TMG_NO_GROUPED_WGRAD": "1"},
    "no_thin_wgrad": {"TMG_NO_THIN_WGRAD": "1", "TMG_NO_MIX_WGRAD_KERNEL": "1"},
    "no_level_fusion": {"TMG_NO_LEVEL_FUSION": "1"},
    "no_lu_fold": {"TMG_NO_LU_FOLD_KERNEL": "1"},
    "no_mix32": {"TMG_NO_MIX32": "1"},
    "plain_wino": {"TMG_NO_FUSED_COUPLING": "1", "TMG_NO_THIN_WGRAD": "1", "TMG_NO_MIX_WGRAD_KERNEL": "1", "TMG_NO_LU_FOLD_KERNEL": "1", "TMG_NO_MIX32": "1"},
    "plain_fused": {"TMG_NO_WINOGRAD": "1", "TMG_NO_THIN_WGRAD": "1", "TMG_NO_MIX_WGRAD_KERNEL": "1", "TMG_NO_LU_FOLD_KERNEL": "1", "TMG_NO_MIX32": "1"},
    "plain_thin": {"TMG_NO_WINOGRAD": "1", "TMG_NO_FUSED_COUPLING": "1", "TMG_NO_LU_FOLD_KERNEL": "1", "TMG_NO_MIX32": "1"},
    "plain_lufold": {"TMG_NO_WINOGRAD": "1", "TMG_NO_FUSED_COUPLING": "1", "TMG_NO_THIN_WGRAD": "1", "TMG_NO_MIX_WGRAD_KERNEL": "1", "TMG_NO_MIX32": "1"},
    "plain_mix32": {"TMG_NO_WINOGRAD": "1", "TMG_NO_FUSED_COUPLING": "1", "TMG_NO_THIN_WGRAD": "1", "TMG_NO_MIX_WGRAD_KERNEL": "1", "TMG_NO_LU_FOLD_KERNEL": "1"},
    "plain": {"TMG_NO_WINOGRAD": "1", "TMG_NO_FUSED_COUPLING": "1", "TMG_NO_THIN_WGRAD": "1", "TMG_NO_MIX_WGRAD_KERNEL": "1",
              "TMG_NO_LU_FOLD_KERNEL": "1", "TMG_NO_MIX32": "1"},
}
CONFIGS = {"M": C.CFG_M, "cfg3": C.CFG3, "cfg2": C.CFG2, "cfg1": C.CFG1, "tiny": C.CFG_TINY, "cfg5": C.CFG5}


def build(cfg):
    import contextlib
    import io
    from nn.tmGlow import TMGlow
    C.seed_all(12345)
    with contextlib.redirect_stdout(io.StringIO()):
        m = TMGlow(**C.build_kwargs(cfg))
    C.perturb_(m, 7, *C.perturb_scales(cfg))
    return m


def inputs(cfg, B, seed):
    h, w = cfg["_in_hw"]
    H_, W_ = h * cfg["_up"], w * cfg["_up"]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg["in_features"], h, w, generator=g)
    y = torch.randn(B, cfg["out_features"], H_, W_, generator=g)
    return x, y, torch.arange(B) + 3, (H_, W_)


def oracle_subset_pass(cfg, sd, x, seeds, hw, dtype, eps, n):
    """Generative direction with the loss on the first n samples: encoder on the whole batch (BatchNorm statistics), flow on n."""
    import tmglow_oracle as O
    P = O.params_from_state_dict(sd, dtype=dtype)
    z_out, c_out = O.encoder(P, cfg, x.to(dtype), True)
    cmean, clsd = z_out[:n].chunk(2, 1)
    clsd = clsd.clamp(-10.0, O.LOG5)
    z = cmean + torch.exp(clsd) * eps[-1][:n].float().to(dtype)
    st = [(a[:n].to(dtype), b[:n].to(dtype)) for a, b in O.init_lstm_states(cfg, seeds[:n], list(hw))]
    yo, ldo, _ = O.decoder_reverse(P, cfg, z, [c[:n] for c in c_out], st, [e[:n].float().to(dtype) for e in eps[:-1]])
    C.loss_reverse(yo, ldo).backward()
    return {"gr": {k: v.grad.clone() for k, v in O.trainable(P).items() if v.grad is not None}, "y": yo.detach(), "ld": ldo.detach()}


def oracle_pass(cfg, sd, x, y, seeds, dtype, eps=None, want_forward=True):
    import tmglow_oracle as O
    H_, W_ = y.shape[2], y.shape[3]
    st = [(h.to(dtype), c.to(dtype)) for h, c in O.init_lstm_states(cfg, seeds, [H_, W_])]
    xx, yy = x.to(dtype), y.to(dtype)
    res = {}
    if want_forward or eps is None:
        P = O.params_from_state_dict(sd, dtype=dtype)
        z, lp, ho, eo = O.tmglow_forward(P, cfg, xx, yy, st, return_eps=True, training=True)
        C.loss_forward(lp, yy).backward()
        res["gf"] = {k: v.grad.clone() for k, v in O.trainable(P).items() if v.grad is not None}
        res["eps"] = [t.detach() for t in eo]
    P = O.params_from_state_dict(sd, dtype=dtype)
    # identical inputs for every evaluation: the latents are the fp64 forward pass's, ROUNDED to fp32 (what the fp32 paths can be given)
    e_in = [t.detach().float().to(dtype) for t in (eps if eps is not None else res["eps"])]
    yr, ld, _ = O.tmglow_reconstruct(P, cfg, xx, st, e_in, training=True)
    C.loss_reverse(yr, ld).backward()
    res["gr"] = {k: v.grad.clone() for k, v in O.trainable(P).items() if v.grad is not None}
    res["y"], res["ld"] = yr.detach(), ld.detach()
    return res


def hip_child(path):
    """Child process: one HIP pass with the environment of its variant; results to `path`.out."""
    d = torch.load(path)
    cfg = CONFIGS[d["config"]]
    m = build(cfg)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.to("cuda").train()
    x, y, seeds, (H_, W_) = d["x"], d["y"], d["seeds"], d["hw"]
    st = m.initLSTMStates(seeds, [H_, W_])
    out = {}
    if d["forward"]:
        z, lp, ho, e = m.forward(x.cuda(), y.cuda(), st, return_eps=True)
        C.loss_forward(lp, y.cuda()).backward()
        out["gf"] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}
        m.load_state_dict(sd)
        m.zero_grad()
    yr, ld, _ = m.reconstruct(x.cuda(), st, [t.float().cuda() for t in d["eps"]])
    n = d.get("loss_samples") or x.shape[0]
    C.loss_reverse(yr[:n], ld[:n]).backward()
    out["gr"] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}
    out["y"], out["ld"] = yr.detach().cpu()[:n], ld.detach().cpu()[:n]
    torch.save(out, path + ".out")


def tensor_errors(got, ref):
    rows = {}
    num = den = 0.0
    for k, r in ref.items():
        g = torch.as_tensor(got[k]).double().reshape(-1)
        r = r.double().reshape(-1)
        e2, r2 = float(((g - r) ** 2).sum()), float((r ** 2).sum())
        num += e2
        den += r2
        sc = float(r.abs().max())
        rows[k] = {"rel_max": float((g - r).abs().max()) / sc if sc > 0 else 0.0, "rel_l2": (e2 / r2) ** 0.5 if r2 > 0 else 0.0,
                   "err2": e2, "numel": r.numel(), "scale": sc}
    glob = (num / max(den, 1e-300)) ** 0.5
    for v in rows.values():
        v["share_of_global_err2"] = v.pop("err2") / max(num, 1e-300)
    return glob, rows


def summarise(got, ref, yard_rows=None, top=5):
    glob, rows = tensor_errors(got, ref)
    worst = max(v["rel_max"] for v in rows.values())

    def line(k):
        v = dict(rows[k], name=k)
        if yard_rows is not None:
            v["oracle_fp32_rel_max"], v["oracle_fp32_rel_l2"] = yard_rows[k]["rel_max"], yard_rows[k]["rel_l2"]
        return v

    by_max = [line(k) for k in sorted(rows, key=lambda k: -rows[k]["rel_max"])[:top]]
    by_share = [line(k) for k in sorted(rows, key=lambda k: -rows[k]["share_of_global_err2"])[:top]]
    # by parameter kind (suffix after the layer name): where does the error live
    kinds = {}
    for k, v in rows.items():
        kind = ".".join(k.split(".")[-3:]) if "affine_layer" in k else k.split(".")[0] + ".*"
        a = kinds.setdefault(kind, {"share": 0.0, "worst_rel_max": 0.0, "n": 0})
        a["share"] += v["share_of_global_err2"]
        a["worst_rel_max"] = max(a["worst_rel_max"], v["rel_max"])
        a["n"] += 1
    kinds = dict(sorted(kinds.items(), key=lambda kv: -kv[1]["share"])[:8])
    return {"global_rel_l2": glob, "worst_rel_max": worst, "worst_by_rel_max": by_max, "worst_by_share": by_share, "by_kind": kinds}, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="M", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--seed", type=int, default=31)
    ap.add_argument("--variants", default="default,no_wino,no_wino_wgrad,no_fused_bwd")
    ap.add_argument("--forward", action="store_true", help="also the density direction")
    ap.add_argument("--loss-samples", type=int, default=0, help="loss on the first n samples of the batch only (oracle: encoder on the "
                                                                "whole batch, flow on n samples): stated-batch parity")
    ap.add_argument("--out", default=None)
    ap.add_argument("--child", default=None)
    args = ap.parse_args()
    if args.child:
        return hip_child(args.child)
    cfg = CONFIGS[args.config]
    m = build(cfg)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    del m
    x, y, seeds, hw = inputs(cfg, args.batch, args.seed)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    if args.loss_samples:
        # latents: seeded normal draws of the right shapes (taken from a batch-1 fp32 oracle forward)
        import tmglow_oracle as O
        with torch.no_grad():
            P = O.params_from_state_dict(sd, requires_grad=False)
            st1 = O.init_lstm_states(cfg, seeds[:1], list(hw))
            _, _, _, e1 = O.tmglow_forward(P, cfg, x[:1], y[:1], st1, return_eps=True, training=True)
        g = torch.Generator().manual_seed(args.seed + 1000)
        eps = [torch.randn((args.batch,) + tuple(e.shape[1:]), generator=g) for e in e1]
        r64 = oracle_subset_pass(cfg, sd, x, seeds, hw, torch.float64, eps, args.loss_samples)
        r32 = oracle_subset_pass(cfg, sd, x, seeds, hw, torch.float32, eps, args.loss_samples)
        r64["eps"] = eps
    else:
        r64 = oracle_pass(cfg, sd, x, y, seeds, torch.float64, want_forward=True)
        r32 = oracle_pass(cfg, sd, x, y, seeds, torch.float32, eps=r64["eps"], want_forward=args.forward)
    rep = {"config": args.config, "batch": args.batch, "input_seed": args.seed, "variants": {}}
    yard = {}
    for key in (("gf", "gr") if args.forward else ("gr",)):
        s, rows = summarise(r32[key], r64[key])
        yard[key] = rows
        rep.setdefault("oracle_fp32", {})[key] = s
    rep["oracle_fp32"]["y_maxabs"] = float((r32["y"].double() - r64["y"]).abs().max())
    tmp = tempfile.mkdtemp(prefix="tmg_parity_")
    for name in args.variants.split(","):
        path = os.path.join(tmp, name + ".pt")
        torch.save({"config": args.config, "x": x, "y": y, "seeds": seeds, "hw": hw, "eps": r64["eps"], "forward": args.forward,
                    "loss_samples": args.loss_samples}, path)
        env = dict(os.environ, **VARIANTS[name])
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            rep["variants"][name] = {"failed": r.stderr[-2000:]}
            continue
        out = torch.load(path + ".out")
        v = {"env": VARIANTS[name], "y_maxabs": float((out["y"].double() - r64["y"]).abs().max()),
             "logdet_rel": float(((out["ld"].double() - r64["ld"]).abs() / r64["ld"].abs().clamp_min(1.0)).max())}
        for key in (("gf", "gr") if args.forward else ("gr",)):
            v[key], _ = summarise(out[key], r64[key], yard[key])
            v[key]["global_rel_l2_vs_oracle_fp32"] = tensor_errors(out[key], r32[key])[0]
        rep["variants"][name] = v
        g = v["gr"]
        print("%-18s reverse grads: global rel-L2 %.3e  worst rel-max %.3e   (fp32 oracle %.3e / %.3e; hip vs fp32 oracle %.3e)  y %.2e" % (
            name, g["global_rel_l2"], g["worst_rel_max"], rep["oracle_fp32"]["gr"]["global_rel_l2"], rep["oracle_fp32"]["gr"]["worst_rel_max"],
            g["global_rel_l2_vs_oracle_fp32"], v["y_maxabs"]))
        for w in g["worst_by_share"][:int(os.environ.get("TMG_REPORT_TOP", 3))]:
            print("     share %.2f  rel_l2 %.2e rel_max %.2e (fp32 oracle %.2e)  %s" % (w["share_of_global_err2"], w["rel_l2"], w["rel_max"], w["oracle_fp32_rel_max"], w["name"]))
        sys.stdout.flush()
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rep, f, indent=1, default=float)


if __name__ == "__main__":
    main()
