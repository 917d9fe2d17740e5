#!/usr/bin/env python3
"""Which ReLU pre-activations of a parity case sit on a kink?  (CPU only; the oracle is the subject.)

A gradient check against an fp64 evaluation is only meaningful where fp32 and fp64 agree on every ReLU mask: one element of a
deep, small feature map on the other side of zero moves whole weight-gradient tensors by 1e-3 of their scale, whatever the
kernels are.  This scan runs the generative direction of the CPU oracle in fp32 and in fp64 on the same inputs and lists every
ReLU call whose masks differ, with the fp64 pre-activation of the flipped elements.

  python tools/kink_scan.py --config M --batch 1 --seeds 31,32,33
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-turbulence_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import parity_report as PR  # noqa: E402


def scan(cfg, sd, x, y, seeds):
    import tmglow_oracle as O
    H_, W_ = y.shape[2], y.shape[3]
    real_relu = F.relu
    masks, flips, state = [], [], {"i": 0, "mode": None}

    def relu(t, inplace=False):
        i = state["i"]
        state["i"] += 1
        if state["mode"] == "record":
            masks.append((t.detach() > 0))
        else:
            m = t.detach() > 0
            bad = (m != masks[i]).nonzero()
            for idx in bad[:8]:
                flips.append({"relu_call": i, "shape": list(t.shape), "index": idx.tolist(), "fp64_preactivation": float(t.detach()[tuple(idx)]),
                              "tensor_scale": float(t.detach().abs().max())})
            if len(bad) > 8:
                flips.append({"relu_call": i, "more": int(len(bad) - 8)})
        return real_relu(t)

    O.F.relu = relu
    try:
        with torch.no_grad():
            st64 = [(h.double(), c.double()) for h, c in O.init_lstm_states(cfg, seeds, [H_, W_])]
            P64 = O.params_from_state_dict(sd, dtype=torch.float64, requires_grad=False)
            O.F.relu = real_relu
            _, _, _, eps = O.tmglow_forward(P64, cfg, x.double(), y.double(), st64, return_eps=True, training=True)
            eps = [e.float() for e in eps]
            O.F.relu = relu
            state.update(i=0, mode="record")
            P32 = O.params_from_state_dict(sd, dtype=torch.float32, requires_grad=False)
            st32 = [(h.float(), c.float()) for h, c in st64]
            O.tmglow_reconstruct(P32, cfg, x, st32, eps, training=True)
            n_calls = state["i"]
            state.update(i=0, mode="compare")
            O.tmglow_reconstruct(P64, cfg, x.double(), st64, [e.double() for e in eps], training=True)
    finally:
        O.F.relu = real_relu
    return {"relu_calls": n_calls, "elements": int(sum(m.numel() for m in masks)), "flipped": flips}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="M", choices=sorted(PR.CONFIGS))
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--seeds", default="31")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    cfg = PR.CONFIGS[args.config]
    m = PR.build(cfg)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    del m
    rep = {"config": args.config, "batch": args.batch, "seeds": {}}
    for s in [int(v) for v in args.seeds.split(",")]:
        x, y, seeds, _ = PR.inputs(cfg, args.batch, s)
        r = scan(cfg, sd, x, y, seeds)
        rep["seeds"][s] = r
        print("input seed %d: %d ReLU calls, %d elements, %d flipped between fp32 and fp64" % (
            s, r["relu_calls"], r["elements"], sum(1 + f.get("more", 0) if "more" in f else 1 for f in r["flipped"])))
        for f in r["flipped"][:6]:
            print("   ", json.dumps(f))
        sys.stdout.flush()
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
