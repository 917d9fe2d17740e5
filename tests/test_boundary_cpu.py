"""CPU-side checks of the drop-in boundary: the C-ABI library exports what include/tmglow_hip.h
declares, the module API mirrors the reference's constructor / state_dict schema (fixtures recorded
from the reference), and the product path refuses to compute without a GPU (no silent fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import common as C


def test_library_exports_every_declared_symbol():
    import tmg_hip
    path = tmg_hip.build()
    lib = ctypes.CDLL(path)
    hdr = open(os.path.join(C.ROOT, "include", "tmglow_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(?:int|int64_t)\s+(tmg_\w+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    assert sorted(tmg_hip.EXPORTS) == declared
    for name in declared:
        assert hasattr(lib, name), name


def test_state_dict_schema_and_seeded_init_match_reference():
    from nn.tmGlow import TMGlow
    d = C.load_npz("cfg1_init_checksums.npz")
    C.seed_all(12345)
    m = TMGlow(**C.build_kwargs(C.CFG1))
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in d["keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in d["shapes"]]
    cs = C.tensor_checksums(sd)
    for k, v in zip(d["keys"], d["vals"]):
        assert abs(cs[str(k)] - v) <= 1e-6 * abs(v) + 1e-9, k


@pytest.mark.parametrize("name,cfg", [("tiny_model.npz", C.CFG_TINY), ("tiny3_model.npz", C.CFG_TINY3)])
def test_reference_state_dict_loads_strictly(name, cfg):
    from nn.tmGlow import TMGlow
    d = C.load_npz(name)
    m = TMGlow(**C.build_kwargs(cfg))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in C.sub(d, "sd.").items()}, strict=True)
    st = m.initLSTMStates(torch.arange(d["x"].shape[0]), list(d["y"].shape[2:]))
    for i, (h, c) in enumerate(st):
        assert np.array_equal(h.numpy(), d["h_in.%d.h" % i]) and np.array_equal(c.numpy(), d["h_in.%d.c" % i])


def test_seed_state_cache_is_bit_identical_and_draws_each_seed_once():
    """`initLSTMStates` keeps every distinct seed's states resident after the first host draw (the loaders draw seeds from
    random_(0, 1000), reference dataLoader.py:284, :422): later mini-batches - any order, repeated seeds - are gathers of the same
    bits the reference's per-call host draw (tmGlow.py:494-509, fixture `h_in.*` recorded from the reference) produces."""
    from nn.tmGlow import TMGlow
    d = C.load_npz("tiny_model.npz")
    cfg = C.CFG_TINY
    m = TMGlow(**C.build_kwargs(cfg))
    hw = list(d["y"].shape[2:])
    calls = []
    real = m._draw_seed_states
    m._draw_seed_states = lambda seeds, dim: (calls.append(list(seeds)), real(seeds, dim))[1]
    a = m.initLSTMStates(torch.tensor([0, 1]), hw)
    for i, (h, c) in enumerate(a):                     # the fixture's seeds are arange(B)
        assert np.array_equal(h.numpy(), d["h_in.%d.h" % i]) and np.array_equal(c.numpy(), d["h_in.%d.c" % i])
    b = m.initLSTMStates(torch.tensor([1, 7, 1, 0]), hw)
    assert calls == [[0, 1], [7]]                        # each distinct seed drawn once
    fresh = TMGlow(**C.build_kwargs(cfg))
    want = fresh.initLSTMStates(torch.tensor([1, 7, 1, 0]), hw)
    for (h, c), (h2, c2), (h0, c0) in zip(b, want, a):
        assert torch.equal(h, h2) and torch.equal(c, c2) and h.shape == h2.shape
        assert torch.equal(h[0], h0[1]) and torch.equal(h[2], h0[1]) and torch.equal(c[3], c0[0])
        assert h.permute(0, 2, 3, 1).is_contiguous()     # channels-last: the layout the flow works in
    # another field size is another cache; returned tensors are fresh (mutating them does not touch the cache)
    b[0][0].zero_()
    again = m.initLSTMStates(torch.tensor([1]), hw)
    assert torch.equal(again[0][0][0], want[0][0][0])
    other = m.initLSTMStates(torch.tensor([1]), [2 * hw[0], 2 * hw[1]])
    assert other[0][0].shape[2] == hw[0] and calls[-1] == [1]
    # the cache never enters the state_dict
    assert not any("seed" in k for k in m.state_dict())
    # seeds outside the loaders' range [0, 1000) (TrainFlow.test / modelPred draw from random_(0, 1e8) for every batch and sample) and
    # calls with cache=False are drawn per call and never become resident: a test epoch must not pin HBM
    held = m._seed_states["bytes"]
    n_rows = len(m._seed_states["rows"])
    big = m.initLSTMStates(torch.tensor([12345678, 1]), [2 * hw[0], 2 * hw[1]])
    nocache = m.initLSTMStates(torch.tensor([5]), [2 * hw[0], 2 * hw[1]], cache=False)
    assert m._seed_states["bytes"] == held and len(m._seed_states["rows"]) == n_rows
    assert calls[-2:] == [[12345678], [5]]
    ref_big = fresh.initLSTMStates(torch.tensor([12345678, 1]), [2 * hw[0], 2 * hw[1]], cache=False)
    assert all(torch.equal(h, h2) and torch.equal(c, c2) for (h, c), (h2, c2) in zip(big, ref_big)) and nocache[0][0].shape[0] == 1


def test_public_api_surface():
    import nn.tmGlow as T
    from nn.modules import actNorm, glowConv, flowAffine, flowLSTMBlock, flowUtils, denseBlock, convLSTM, misc
    for mod, names in [(T, ["TMGlow", "Encoder", "LSTMCFlowDecoder"]), (actNorm, ["ActNorm"]),
                       (glowConv, ["InvertibleConv1x1", "InvertibleConv1x1LU"]),
                       (flowAffine, ["AffineCouplingLayer", "LSTMAffineCouplingLayer"]),
                       (flowLSTMBlock, ["AffineCouplingBlock", "UnNormedAffineCouplingBlock", "LSTMCouplingBlock", "LSTMFLowBlock"]),
                       (flowUtils, ["Squeeze", "CheckerSqueeze", "GaussianDiag", "Conv2dZeros", "LatentEncoder", "Split"]),
                       (denseBlock, ["DenseBlock", "NoNormDenseBlock"]), (convLSTM, ["ConvLSTMCell", "ResidLSTMBlock"]),
                       (misc, ["UpsamplingLinear"])]:
        for n in names:
            assert hasattr(mod, n), n
    m = T.TMGlow(1, 1, [4, 4], [4, 4], cglow_upscale=2, rec_features=2)  # the reference's own self-test ctor (tmGlow.py:516)
    for attr in ("encoder", "glow", "glow_blocks", "rec_features", "in_mu", "in_std", "out_mu", "out_std"):
        assert hasattr(m, attr)
    for meth in ("forward", "sample", "reconstruct", "initLSTMStates"):
        assert callable(getattr(m, meth))


def test_no_cpu_fallback():
    from nn.tmGlow import TMGlow
    m = TMGlow(**C.build_kwargs(C.CFG_TINY))
    with pytest.raises(RuntimeError, match="no CPU path|HIP"):
        m.sample(torch.randn(1, 2, 8, 8))
    with pytest.raises(AssertionError):
        m.glow.forward(torch.zeros(1, 2, 16, 16), [None], None)


def test_zero_padded_channel_layout_is_exact():
    """Levels whose channel half is not a multiple of 4 (3-channel fields) run on [x1 | 0.. | x2 | 0..]: the padded mix applied to
    the padded activations must reproduce the un-padded mix on the real channels and keep the padding channels exactly zero
    (host-side parameter helper of LSTMFLowBlock; the activations are padded by tmg_pad_halves, which tests/test_hip_ops.py holds to
    the same concatenation on the GPU; the kernels only ever see the padded, float4-aligned problem)."""
    import torch
    from nn.modules.flowLSTMBlock import LSTMFLowBlock as B
    g = torch.Generator().manual_seed(0)

    def pad_x(x, ch, pad):        # the layout: [x1 | 0.. | x2 | 0..]
        z = torch.zeros(x.shape[:-1] + (pad,))
        return torch.cat([x[..., :ch], z, x[..., ch:], z], -1)

    def unpad_x(xp, ch, pad):
        return torch.cat([xp[..., :ch], xp[..., ch + pad:2 * ch + pad]], -1)

    for C_ in (12, 20, 6):
        ch = C_ // 2
        pad = (-ch) % 4
        W, b = torch.randn(3, C_, C_, generator=g), torch.randn(3, C_, generator=g)
        x = torch.randn(2, 5, 7, C_, generator=g)
        xp = pad_x(x, ch, pad)
        assert xp.shape[-1] == 2 * (ch + pad) and (ch + pad) % 4 == 0
        assert torch.equal(unpad_x(xp, ch, pad), x)
        Wp, bp = B._pad_mix(W, b, ch, pad)
        for k in range(3):
            yp = xp @ Wp[k].t() + bp[k]
            y = x @ W[k].t() + b[k]
            assert torch.allclose(unpad_x(yp, ch, pad), y, atol=1e-6)
            assert float(yp[..., ch:ch + pad].abs().max()) == 0.0 and float(yp[..., 2 * ch + pad:].abs().max()) == 0.0


def test_hip_adam_falls_back_to_torch_off_device():
    """tmg_optim.HipAdam on tensors the one-launch kernel does not take (here: CPU tensors) runs torch's own single-tensor Adam on
    the same state - same numbers as torch.optim.Adam, same state-dict schema."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deep-turbulence_amd"))
    from tmg_optim import HipAdam
    g = torch.Generator().manual_seed(3)
    pa = [torch.randn(s, generator=g).requires_grad_(True) for s in [(5,), (3, 4), (2, 2, 3, 3)]]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = HipAdam(pa, lr=1e-2, weight_decay=1e-8, amsgrad=True)
    ob = torch.optim.Adam(pb, lr=1e-2, weight_decay=1e-8, amsgrad=True, foreach=False)
    for _ in range(3):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
        assert set(oa.state[a]) == set(ob.state[b]) and float(oa.state[a]["step"]) == 3
    ob.load_state_dict(oa.state_dict())


def test_leading_channels_node_matches_plain_slicing():
    """tmg_ops.LeadingChannelsFn (x -> (x, x[..., :n]) with the slice's gradient added in place into the whole tensor's) gives the
    gradients of plain slicing - pure tensor bookkeeping, no kernel involved."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deep-turbulence_amd"))
    import tmg_ops
    g = torch.Generator().manual_seed(2)
    x0 = torch.randn(2, 3, 4, 6, generator=g)
    w_all, w_lead = torch.randn(2, 3, 4, 6, generator=g), torch.randn(2, 3, 4, 2, generator=g)
    res = []
    for mode in range(2):
        x = x0.clone().requires_grad_(True)
        h = x * 1.5
        xa, x1 = tmg_ops.LeadingChannelsFn.apply(h, 2) if mode else (h, h[..., :2])
        ((xa.tanh() * w_all).sum() + (x1 * x1 * w_lead).sum()).backward()
        res.append(x.grad)
    assert torch.allclose(res[0], res[1], rtol=1e-6, atol=1e-7)
    # only the slice takes part in the loss
    x = x0.clone().requires_grad_(True)
    _, x1 = tmg_ops.LeadingChannelsFn.apply(x * 1.0, 2)
    (x1 * w_lead).sum().backward()
    assert torch.equal(x.grad[..., :2], w_lead) and not bool(x.grad[..., 2:].any())


def test_derived_cache_validity_rules():
    """tmg_ops.DerivedCache (padded weights / folded mixes evaluated once per BPTT window): a value is reused only while its source
    parameters are unchanged (torch version counter, data pointer, PARAM_GENERATION for writers torch does not see), in the same grad
    mode, and until a backward pass has consumed its graph; a rebuilt value carries a fresh graph."""
    import torch
    import tmg_ops as ops
    p = torch.nn.Parameter(torch.arange(6.0).reshape(2, 3))
    builds = []

    def build():
        builds.append(1)
        return [torch.cat([p, torch.zeros(2, 1)], 1), p.sum(0)]

    c = ops.DerivedCache()
    # grad mode OUTSIDE a BPTT window: every call builds its own value (two forward passes may be followed by two separate backward
    # passes: a shared graph would be freed by the first)
    assert c.get("x", [p], (1,), build) is not c.get("x", [p], (1,), build) and len(builds) == 2
    builds.clear()
    with ops.bptt_window():
        a = c.get("x", [p], (1,), build)
        b = c.get("x", [p], (1,), build)
        assert a is b and len(builds) == 1                       # T time-steps of a window share one evaluation
        assert c.get("x", [p], (2,), build) is not a and len(builds) == 2      # another `extra` (shape key) is another value
        a = c.get("x", [p], (1,), build)
        (a[0].sum() * 2.0 + a[1].sum()).backward()               # a backward pass goes through the cached graph: stale
        assert torch.equal(p.grad, torch.full((2, 3), 3.0))
        a2 = c.get("x", [p], (1,), build)
        assert a2 is not a
        a2[0].sum().backward()                                    # the rebuilt value has a graph of its own
    assert ops.DerivedCache.window_depth == 0
    with torch.no_grad():
        a3 = c.get("x", [p], (1,), build)                     # grad mode is part of the key
        assert a3 is not a2 and not a3[0].requires_grad
        assert c.get("x", [p], (1,), build) is a3
        p.add_(1.0)                                            # an in-place update moves the version counter
        a4 = c.get("x", [p], (1,), build)
        assert a4 is not a3 and float(a4[0][0, 0]) == 1.0
        ops.PARAM_GENERATION[0] += 1                           # a writer torch does not see (tmg_optim.HipAdam's kernel)
        assert c.get("x", [p], (1,), build) is not a4


def test_fused_grad_accumulation_sums_like_autograd_on_cpu():
    """tmg_ops.fused_grad_accumulation with a plain-torch node that routes its parameter gradient through the sink (as the HIP nodes
    do): three uses of two parameters in one backward pass give autograd's own sums, on top of an existing .grad as well; a derived
    (DerivedCache proxy) tensor's summed gradient goes back through the graph that built it, once."""
    import torch
    import tmg_ops as ops

    class Lin(torch.autograd.Function):
        @staticmethod
        def forward(ctx, w, x):
            ctx.save_for_backward(w, x)
            return x @ w.t()

        @staticmethod
        def backward(ctx, g):
            w, x = ctx.saved_tensors
            return ops._defer((w,), (g.t() @ x,)) + (g @ w,)

    g = torch.Generator().manual_seed(1)
    w1, w2 = (torch.nn.Parameter(torch.randn(4, 3, generator=g)) for _ in range(2))
    xs = [torch.randn(5, 3, generator=g) for _ in range(3)]

    def loss(pad=None):
        tot = 0.0
        for t, x in enumerate(xs):
            wa = w1 if pad is None else pad
            xx = x if pad is None else torch.cat([x, torch.zeros(5, 1)], 1)
            tot = tot + (Lin.apply(wa, xx) ** 2).sum() * (t + 1) + (Lin.apply(w2, x) ** 3).sum()
        return tot

    loss().backward()
    ref = [w1.grad.clone(), w2.grad.clone()]
    w1.grad = w2.grad = None
    with ops.fused_grad_accumulation():
        loss().backward()
        assert w1.grad is None                                 # nothing reaches .grad before the context closes
    assert torch.allclose(w1.grad, ref[0], rtol=1e-6, atol=1e-6) and torch.allclose(w2.grad, ref[1], rtol=1e-6, atol=1e-6)
    with ops.fused_grad_accumulation():                        # on top of existing gradients
        loss().backward()
    assert torch.allclose(w1.grad, 2 * ref[0], rtol=1e-6, atol=1e-5)
    # a derived tensor (zero-padded copy of w1, shared by the three uses) as a sink proxy
    w1.grad = w2.grad = None
    cache = ops.DerivedCache()
    with ops.bptt_window() as win:
        pad = cache.get("pad", [w1], (), lambda: [torch.cat([w1, torch.zeros(4, 1)], 1)], proxies=True)[0]
        assert ops._GradSink.is_proxy(pad)
        win.backward(loss(pad))
        assert torch.allclose(w1.grad, ref[0], rtol=1e-6, atol=1e-6) and torch.allclose(w2.grad, ref[1], rtol=1e-6, atol=1e-6)
        assert cache.get("pad", [w1], (), lambda: [torch.cat([w1, torch.zeros(4, 1)], 1)], proxies=True)[0] is not pad    # consumed: rebuilt
    # a proxy registration dies with its tensor: the TensorImpl address of a dead cache entry may be handed to any later tensor, which
    # must not become a sink target by that accident (round 4: "backward through the graph a second time" with several models in one
    # process) - simulated by registering a weak reference to a tensor that is gone under a live tensor's id
    import weakref
    dead = torch.zeros(3, requires_grad=True) * 2.0
    live = torch.ones(3, requires_grad=True) * 2.0
    ops._GradSink.proxy_ids[live._cdata] = weakref.ref(dead)
    del dead
    assert not ops._GradSink.is_proxy(live)
    del ops._GradSink.proxy_ids[live._cdata]


def test_captured_window_argument_structures():
    """tmg_dist.CapturedWindow copies nested list / tuple structures of tensors into the graph's own inputs: the flattening order and the
    structure-preserving map it relies on (CPU: no graph is recorded here)."""
    import tmg_dist
    a, b, c = torch.zeros(2), torch.ones(3), torch.full((1,), 2.0)
    args = ([a, (b, None)], [(c, a)])
    flat = tmg_dist._flat_tensors(args)
    assert [t.data_ptr() for t in flat] == [a.data_ptr(), b.data_ptr(), c.data_ptr(), a.data_ptr()]
    m = tmg_dist._map_tensors(args, lambda t: t + 1)
    assert isinstance(m, tuple) and isinstance(m[0], list) and isinstance(m[0][1], tuple) and m[0][1][1] is None
    assert torch.equal(m[0][0], a + 1) and torch.equal(m[1][0][0], c + 1)
    with pytest.raises(TypeError, match="tensors or nested lists"):
        tmg_dist._flat_tensors([a, 3.0])
