"""ctypes binding of libtmglow_hip.so (C ABI: include/tmglow_hip.h) for PyTorch-ROCm tensors.

PyTorch is plumbing here: device memory (caching allocator), the current HIP stream and autograd
bookkeeping.  Every numerical operation of the hot path is a kernel of the shared library.  There is
no CPU or eager fallback: calling an op without the library, or with a tensor that is not a
contiguous-NHWC fp32 CUDA/HIP tensor, raises.

Internal tensor convention: activations are torch tensors of shape [B, H, W, C] (NHWC, contiguous)
or channel-slice views of such tensors.  `seg(t)` turns one into the (pointer, stride, offset, n)
descriptor the library expects.
"""
import ctypes
import os
import weakref
import subprocess
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtmglow_hip.so")
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["tmg_conv.hip", "tmg_pointwise.hip", "tmg_physics.hip", "tmg_mix16.hip", "tmg_coupling.hip", "tmg_wino.hip", "tmg_thin.hip", "tmg_glue.hip"]
# Sources compiled WITHOUT the packed-fp32 vector instructions (v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32): beside MFMAs a packed
# f32 instruction costs ~13 cycles more than the two scalar ones it replaces (MI355X_MICROARCH.md, cycle constants, 'price of one
# filler beside MFMAs'), and the compiler SLP-packs adjacent scalar adds / multiplies by itself under -O3.  The matrix-core kernels'
# files are listed; the vector-ALU kernels (tmg_pointwise.hip, tmg_physics.hip) keep the packed forms, which double their arithmetic
# rate.  Round 3 measured the switch inside the noise and left the list empty; round 5 on one box, alternating three times: packed
# everywhere 43.62 / 43.98 / 43.76 ms per step, this list 43.51 / 43.60 / 43.37 (wino + conv + coupling alone 43.64 / 43.52 / 43.58).
# TMG_NOPK=wino,conv (env, build time; TMG_NOPK= for none) overrides the list for A/B measurements.
NO_PACKED_F32 = ["tmg_wino.hip", "tmg_conv.hip", "tmg_coupling.hip", "tmg_thin.hip", "tmg_mix16.hip"]
_lib = None

c_i64 = ctypes.c_int64
c_vp = ctypes.c_void_p

EXPORTS = [
    "tmg_conv_pack", "tmg_conv_pack_map", "tmg_conv_fwd", "tmg_conv_fwd_add", "tmg_affine_bwd_scaled", "tmg_c1_fwd_add", "tmg_conv_wgrad", "tmg_conv_wgrad_ws_floats", "tmg_conv_rep_border_fix", "tmg_conv_dgrad_direct",
    "tmg_affine_apply", "tmg_affine_apply_pass", "tmg_bn_finalize", "tmg_affine_bwd", "tmg_lstm_pointwise_fwd", "tmg_lstm_pointwise_bwd", "tmg_gauss_fwd",
    "tmg_gauss_bwd", "tmg_checker", "tmg_upsample_fwd", "tmg_upsample_bwd", "tmg_chan_reduce", "tmg_bn_bwd_apply",
    "tmg_phys_fwd", "tmg_phys_rms", "tmg_phys_bwd", "tmg_conv_wgrad_grouped", "tmg_conv_wgrad_grouped_ws_floats", "tmg_conv_pack_batched", "tmg_masked_add", "tmg_c1x2_fwd", "tmg_c1_fwd", "tmg_c1_bwd", "tmg_dense2_bwd", "tmg_dkappa", "tmg_prof_enable", "tmg_prof_collect", "tmg_mix_f16", "tmg_phys_bwd_dev", "tmg_coupling_fwd", "tmg_coupling_bwd",
    "tmg_conv_wino_pack", "tmg_conv_wino_fwd", "tmg_conv_wino_narrow", "tmg_conv_wino_wgrad", "tmg_conv_wino_wgrad_ws_floats", "tmg_mix_f32", "tmg_lu_fold_fwd", "tmg_lu_fold_bwd", "tmg_lu_fold_bwd_split", "tmg_level_finish", "tmg_conv_wgrad_thin_grouped", "tmg_mix_wgrad_grouped", "tmg_layer_planes", "tmg_conv_wino_wgrad_grouped", "tmg_conv_wino_wgrad_grouped_ws_floats", "tmg_adam_step", "tmg_chan_moments", "tmg_bn_finalize64", "tmg_mix_f32_affine_fwd", "tmg_mix_f32_affine_bwd", "tmg_conv_pack_many", "tmg_pad_halves", "tmg_coupling_fwd_halves", "tmg_coupling_bwd_halves", "tmg_fill_i64", "tmg_conv_wino_pack3", "tmg_conv_wino_fwd3", "tmg_mat_inverse", "tmg_gauss_sample", "tmg_reverse_loss_fwd", "tmg_reverse_loss_bwd", "tmg_sum_terms", "tmg_vec_sum", "tmg_level_pack", "tmg_spread2", "tmg_phys_fields",
]


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 into libtmglow_hip.so (in-tree).  One object per source (recompiled only when the
    source or a header changed, all stale ones in parallel), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    inc = os.path.join(os.path.dirname(_HERE), "include")
    headers = [os.path.join(CSRC, "tmg_common.h"), os.path.join(inc, "tmglow_hip.h")]
    hmt = max(os.path.getmtime(h) for h in headers)
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-I", inc]
    nopk = os.environ.get("TMG_NOPK")
    nopk = NO_PACKED_F32 if nopk is None else [("tmg_%s.hip" % n) for n in nopk.split(",") if n]
    jobs, objs = [], []
    stamps = []
    for src in SOURCES:
        sp, ob = os.path.join(CSRC, src), os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(ob)
        extra = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"] if src in nopk else []
        extra += os.environ.get("TMG_EXTRA_DEFS", "").split()      # diagnostic builds (e.g. -DTMG_WINO_STAMP), never the product's
        # an object is stale when its source / a header is newer OR when it was compiled with other flags (a diagnostic build
        # leaves objects that are newer than their sources: the next plain build must not link them into the product library)
        line = " ".join(flags + extra)
        stamp = ob + ".flags"
        old = open(stamp).read() if os.path.exists(stamp) else None
        if (force or not os.path.exists(ob) or os.path.getmtime(ob) < max(os.path.getmtime(sp), hmt)
                or (old is not None and old != line) or (old is None and extra)):
            jobs.append([hipcc] + flags + extra + ["-c", sp, "-o", ob])
        stamps.append((stamp, line))
    if not jobs and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
        for stamp, line in stamps:
            if not os.path.exists(stamp):
                open(stamp, "w").write(line)
        return LIB_PATH

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        # (hipcc hands -target-feature to its HOST pass as well, which prints one "not a recognized feature" line per kernel: dropped)
        err = "\n".join(ln for ln in r.stderr.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in ln)
        if err.strip():
            print(err, file=sys.stderr, flush=True)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    for stamp, line in stamps:
        open(stamp, "w").write(line)
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libtmglow_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`. "
                               "There is no fallback path." % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        for name in EXPORTS:
            getattr(_lib, name).restype = ctypes.c_int
        _lib.tmg_conv_wgrad_ws_floats.restype = ctypes.c_int64
        _lib.tmg_conv_wgrad_grouped_ws_floats.restype = ctypes.c_int64
    return _lib


def prof_enable(on):
    """False / 0: off; True / 1: time the matrix-core kernels; 2: also the bandwidth-bound kernel classes."""
    lib().tmg_prof_enable(c_i64(int(on)))


def prof_kernel_id(name):
    """Kernel id of a name returned by prof_collect (for prof_enable(100 + id): time only that kernel)."""
    l = lib()
    l.tmg_prof_name.restype = ctypes.c_char_p
    for k in range(64):
        if l.tmg_prof_name(c_i64(k)).decode() == name:
            return k
    raise KeyError(name)


def prof_collect():
    """{kernel name: (launches, total ms, total algorithmic work)} of the event-timed kernels: work = flops for the matrix-core
    kernels, HBM bytes for the classes whose name starts with "hbm:"."""
    l = lib()
    l.tmg_prof_name.restype = ctypes.c_char_p
    nk = 64
    buf = (ctypes.c_double * (3 * nk))()
    n = l.tmg_prof_collect(buf, c_i64(nk))
    out = {}
    for k in range(n):
        if buf[3 * k] > 0:
            out[l.tmg_prof_name(c_i64(k)).decode()] = (int(buf[3 * k]), buf[3 * k + 1], buf[3 * k + 2])
    return out


def _stream():
    """Raw handle of torch's CURRENT stream on the current device.  (torch.cuda.current_stream() builds a Stream object through three
    layers of Python per call - 8 us, once per kernel launch: 3 ms of host time per step; the raw getter is a single C call.)"""
    return c_vp(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _chk(code, name):
    if code != 0:
        raise RuntimeError("%s failed with code %d" % (name, code))


def _ptr(t):
    return c_vp(t.data_ptr()) if t is not None else c_vp(0)


def check_act(t):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError("TM-Glow HIP ops need fp32 tensors on a HIP device (got %s on %s); there is no CPU path"
                           % (t.dtype, t.device))


def seg(t):
    """(ptr, pixel stride, channel offset folded into ptr => 0, n) of an NHWC tensor or channel-slice view."""
    check_act(t)
    B, H, W, C = t.shape
    s = t.stride()
    ps = s[2] if W > 1 else (s[1] // W if H > 1 else (s[0] // (H * W) if B > 1 else C))
    ok = s[3] == 1 or C == 1
    if W > 1:
        ok = ok and (H == 1 or s[1] == W * ps) and (B == 1 or s[0] == H * W * ps)
    if not ok or ps < C:
        raise RuntimeError("tensor is not an NHWC channel-slice: shape %s strides %s" % (tuple(t.shape), s))
    return (t.data_ptr(), int(ps), 0, int(C))


def check_device(t):
    """Kernels launch on the CURRENT device's current stream: a tensor that lives on another GPU would be read through a
    foreign pointer (fault, or silent peer traffic).  The model entry points (TMGlow.forward / sample / reconstruct) switch to
    their input's device themselves; stand-alone modules must be called with their device current."""
    if t.is_cuda and t.device.index != torch.cuda.current_device():
        raise RuntimeError("TM-Glow HIP op called with a tensor on %s while the current device is cuda:%d: wrap the call in "
                           "`with torch.cuda.device(t.device):` (or torch.cuda.set_device once per process)"
                           % (t.device, torch.cuda.current_device()))


def nhwc(x):
    """API tensor [B,C,H,W] (any strides) -> internal [B,H,W,C] contiguous (free if channels_last)."""
    check_act(x)
    check_device(x)
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    """internal [B,H,W,C] -> API-shaped [B,C,H,W] view (channels_last strides)."""
    return t.permute(0, 3, 1, 2)


def _segs(tensors):
    d = [seg(t) for t in tensors]
    ptrs = (c_vp * len(d))(*[x[0] for x in d])
    desc = (c_i64 * (3 * len(d)))(*[v for x in d for v in (x[1], x[2], x[3])])
    return ptrs, desc, len(d)


def _d2(t):
    s = seg(t)
    return (c_i64 * 2)(s[1], 0)


def _i64(*v):
    return (c_i64 * len(v))(*[int(x) for x in v])


# ------------------------------------------------------------------------------------------------
# dense contractions
# ------------------------------------------------------------------------------------------------
# Bumped by code that rewrites parameter memory behind torch's version counters (tmg_optim.HipAdam, tmg_dist.broadcast_parameters,
# tmg_ops.invalidate_derived); tmg_ops.PARAM_GENERATION is this list.
PARAM_GENERATION = [0]


class _PackPlan:
    """The packed MFMA operands of the model's PARAMETERS, all re-packed by a few launches per parameter update instead of one
    launch per conv and pass (round 5: 67 tmg_conv_pack launches of ~4.6 us per training step, 670 per eager BPTT window).  The plan
    learns which (parameter, mode, cin_eff, cmap) operands the model asks for; the first request that finds its operand stale - the
    parameter's memory, torch version counter or PARAM_GENERATION moved - re-packs EVERY known operand whose parameter moved
    (tmg_conv_pack_many, 48 jobs per launch), later requests of the same parameter state are dictionary look-ups.  Only nn.Parameter
    objects take part (identity + weak reference): temporaries derived from parameters have no version to check and are packed per call
    as before.  While a hipGraph is being recorded the plan is bypassed (the recorded pass must re-pack at every replay)."""

    def __init__(self):
        self.jobs = {}

    @staticmethod
    def _state(w):
        return (w.data_ptr(), w._version, PARAM_GENERATION[0], w.device)

    def get(self, w, mode, cin_eff, cmap):
        key = (id(w), int(mode), int(cin_eff), cmap)
        e = self.jobs.get(key)
        st = self._state(w)
        if e is not None and e["ref"]() is w and e["state"] == st:
            return e["out"]
        self.jobs[key] = {"ref": weakref.ref(w), "state": None, "out": None, "spec": (int(mode), int(cin_eff), cmap)}
        self._repack(w.device)
        return self.jobs[key]["out"]

    def _repack(self, device):
        todo = []
        for k in list(self.jobs):
            e = self.jobs[k]
            w = e["ref"]()
            if w is None:
                del self.jobs[k]
            elif w.device == device and e["state"] != self._state(w):
                todo.append((e, w))
        outs = conv_pack_many([(w,) + e["spec"] for e, w in todo])
        for (e, w), o in zip(todo, outs):
            e["out"], e["state"] = o, self._state(w)


_PACK_PLAN = _PackPlan()


def conv_pack(w, mode, cin_eff=0, cmap=None):
    """w: torch layout [Cout, Cin, k, k] -> packed MFMA operand (see tmg_conv_pack); cin_eff > Cin zero-extends the
    input-channel dimension.  cmap = (cvalid, csplit, cgap) selects / re-maps source channels (tmg_conv_pack_map).  Parameters go
    through the pack plan (_PackPlan: the returned tensor is shared by every request of the same parameter state - read-only)."""
    check_act(w)
    if (isinstance(w, torch.nn.Parameter) and w.is_contiguous() and os.environ.get("TMG_NO_PACK_PLAN") is None
            and not torch.cuda.is_current_stream_capturing()):
        return _PACK_PLAN.get(w, mode, cin_eff, tuple(int(v) for v in cmap) if cmap is not None else None)
    w = w.contiguous()
    Cout, Cin, k, _ = w.shape
    if cmap is not None:
        ce = int(cin_eff)
        K, N = (ce, Cout) if mode == 0 else (Cout, ce)
        Kp, Np = (K + 15) // 16 * 16, (N + 15) // 16 * 16
        wpk = torch.empty(k * k * Kp * Np, device=w.device, dtype=torch.float32)
        _chk(lib().tmg_conv_pack_map(_ptr(w), _ptr(wpk), c_i64(Cout), c_i64(Cin), c_i64(ce), c_i64(k), c_i64(mode), _i64(*cmap), _stream()),
             "tmg_conv_pack_map")
        return wpk
    ce = max(int(cin_eff), Cin)
    K, N = (ce, Cout) if mode == 0 else (Cout, ce)
    Kp, Np = (K + 15) // 16 * 16, (N + 15) // 16 * 16
    wpk = torch.empty(k * k * Kp * Np, device=w.device, dtype=torch.float32)
    _chk(lib().tmg_conv_pack(_ptr(w), _ptr(wpk), c_i64(Cout), c_i64(Cin), c_i64(ce), c_i64(k), c_i64(mode), _stream()), "tmg_conv_pack")
    return wpk


def conv_pack_many(jobs):
    """jobs: list of (w, mode) or (w, mode, cin_eff, cmap) as conv_pack takes them -> list of packed operands, <= 48 jobs per launch
    (tmg_conv_pack_many).  All operands of a launch live in one allocation."""
    outs = []
    for g0 in range(0, len(jobs), 48):
        grp = [tuple(j) + (0, None)[len(j) - 2:] for j in jobs[g0:g0 + 48]]
        desc, sizes, ws = [], [], []
        for w, mode, cin_eff, cmap in grp:
            check_act(w)
            assert w.is_contiguous()
            Cout, Cin, k, _ = w.shape
            ce = int(cin_eff) if cmap is not None else max(int(cin_eff), Cin)
            K, N = (ce, Cout) if mode == 0 else (Cout, ce)
            Kp, Np = (K + 15) // 16 * 16, (N + 15) // 16 * 16
            m = cmap if cmap is not None else (Cin, 0x7fffffff, 0)
            desc += [Cout, Cin, ce, k, mode, m[0], m[1], m[2]]
            sizes.append(k * k * Kp * Np)
            ws.append(w)
        flat = torch.empty(sum(sizes), device=ws[0].device, dtype=torch.float32)
        views, o = [], 0
        for n_ in sizes:
            views.append(flat[o:o + n_])
            o += n_
        wp = (c_vp * len(ws))(*[w.data_ptr() for w in ws])
        op = (c_vp * len(ws))(*[v.data_ptr() for v in views])
        _chk(lib().tmg_conv_pack_many(wp, op, _i64(*desc), c_i64(len(ws)), _stream()), "tmg_conv_pack_many")
        outs += views
    return outs


def conv_pack_batched(w, mode, cin_eff=0, cmap=None):
    """w: [N, Cout, Cin, k, k] -> [N, packed] (one launch); arguments as conv_pack."""
    check_act(w)
    w = w.contiguous()
    N_, Cout, Cin, k, _ = w.shape
    ce = int(cin_eff) if cmap is not None else max(int(cin_eff), Cin)
    K, N = (ce, Cout) if mode == 0 else (Cout, ce)
    Kp, Np = (K + 15) // 16 * 16, (N + 15) // 16 * 16
    wpk = torch.empty((N_, k * k * Kp * Np), device=w.device, dtype=torch.float32)
    m = cmap if cmap is not None else (Cin, 0x7fffffff, 0)
    _chk(lib().tmg_conv_pack_batched(_ptr(w), _ptr(wpk), c_i64(N_), c_i64(Cout), c_i64(Cin), c_i64(ce), c_i64(k), c_i64(mode), _i64(*m),
                                     _stream()), "tmg_conv_pack_batched")
    return wpk


def conv_fwd(inputs, wpk, Cout, ksize, stride, outs, bias=None, kappa=None, in_scale=None, in_shift=None, relu_in=False,
             pad_rep=False, relu_out=False, accumulate=False, add=None):
    """inputs / outs: lists of NHWC tensors (or channel-slice views) forming the channel concatenation.
    add: optional NHWC tensor / channel-slice view with Cout channels summed in before bias and scale."""
    B, Hin, Win, _ = inputs[0].shape
    Hout, Wout = outs[0].shape[1], outs[0].shape[2]
    ip, idesc, n_in = _segs(inputs)
    op, odesc, n_out = _segs(outs)
    Cin = sum(t.shape[3] for t in inputs)
    assert sum(t.shape[3] for t in outs) == Cout
    dims = _i64(B, Hin, Win, Hout, Wout, ksize, stride, Cin, Cout, relu_in, pad_rep, relu_out, accumulate)
    if add is not None:
        _chk(lib().tmg_conv_fwd_add(ip, idesc, c_i64(n_in), _ptr(wpk), _ptr(bias), _ptr(kappa), _ptr(in_scale), _ptr(in_shift), _ptr(add),
                                    _d2(add), op, odesc, c_i64(n_out), dims, _stream()), "tmg_conv_fwd_add")
        return
    _chk(lib().tmg_conv_fwd(ip, idesc, c_i64(n_in), _ptr(wpk), _ptr(bias), _ptr(kappa), _ptr(in_scale), _ptr(in_shift), op, odesc,
                            c_i64(n_out), dims, _stream()), "tmg_conv_fwd")


def wino_eligible(Cin, Cout, ksize, stride):
    """Shapes the Winograd F(2x2, 3x3) kernel takes over from the direct implicit GEMM: 3x3 / stride 1 with many output
    channels (the matrix-pipe-bound contractions).  TMG_NO_WINOGRAD=1 keeps everything on the direct kernel."""
    return (ksize == 3 and stride == 1 and Cout >= 64 and Cout % 4 == 0 and Cin % 4 == 0 and Cin >= 16
            and os.environ.get("TMG_NO_WINOGRAD") is None)


def wino_narrow_eligible(Cin, Cout):
    """Shapes of the few-output-channel Winograd kernel (the input gradients of the wide contractions, the ConvLSTM block's
    narrow convs): 3x3 / stride 1, Cout <= 48, Cin >= 64."""
    return Cout <= 48 and Cout % 4 == 0 and Cin % 4 == 0 and Cin >= 64 and os.environ.get("TMG_NO_WINOGRAD") is None


def conv_wino_pack(w, mode=0, nvalid=0):
    """Winograd operand U = G g G^T of a [Cout, Cin, 3, 3] weight: [16][K_pad/16][N_pad][16] floats.  mode 0: forward (K = Cin,
    N = Cout); mode 1: input gradient w.r.t. the first `nvalid` input channels (K = Cout, N = nvalid or Cin), taps flipped."""
    Cout, Cin = w.shape[0], w.shape[1]
    assert w.shape[2] == 3 and w.shape[3] == 3
    w = w.contiguous()
    K, N = (Cin, Cout) if mode == 0 else (Cout, nvalid if 0 < nvalid < Cin else Cin)
    U = torch.empty(16 * ((K + 15) // 16 * 16) * ((N + 15) // 16 * 16), device=w.device, dtype=torch.float32)
    _chk(lib().tmg_conv_wino_pack(_ptr(w), _ptr(U), c_i64(Cout), c_i64(Cin), c_i64(mode), c_i64(nvalid), _stream()), "tmg_conv_wino_pack")
    return U


# Arithmetic of the WIDE Winograd contractions (conv3x3_auto): "f32" = fp32 MFMA (the default, the headline's); "bf16x3" = the bf16
# matrix pipe at fp32 accuracy (three-way exact split of both operands, six part products: tmg_conv_wino_fwd3) - opt-in
_WINO_PRECISION = ["bf16x3" if os.environ.get("TMG_WINO_BF3") else "f32"]


def set_winograd_precision(kind):
    if kind not in ("f32", "bf16x3"):
        raise ValueError("Winograd precision must be 'f32' or 'bf16x3', got %r" % (kind,))
    _WINO_PRECISION[0] = kind


def winograd_precision():
    return _WINO_PRECISION[0]


def conv_wino_pack3(w, mode=0, nvalid=0):
    """bf16x3 Winograd operand of a [Cout, Cin, 3, 3] weight (tmg_conv_wino_pack3): U = G g G^T in fp32, split into three bf16 parts,
    in A-fragment order of mfma_f32_16x16x32_bf16.  Modes as conv_wino_pack."""
    Cout, Cin = w.shape[0], w.shape[1]
    assert w.shape[2] == 3 and w.shape[3] == 3
    w = w.contiguous()
    K, N = (Cin, Cout) if mode == 0 else (Cout, nvalid if 0 < nvalid < Cin else Cin)
    U = torch.empty(16 * ((K + 31) // 32 * 32) * ((N + 15) // 16 * 16) * 3, device=w.device, dtype=torch.int16)
    _chk(lib().tmg_conv_wino_pack3(_ptr(w), _ptr(U), c_i64(Cout), c_i64(Cin), c_i64(mode), c_i64(nvalid), _stream()), "tmg_conv_wino_pack3")
    return U


def conv_wino_fwd3(inputs, U, Cout, outs, bias=None, relu_in=False, pad_rep=False):
    """conv_wino_fwd with the bf16x3 operand of conv_wino_pack3; False outside the envelope (nothing was launched)."""
    if isinstance(outs, torch.Tensor):
        outs = [outs]
    B, Hin, Win, _ = inputs[0].shape
    ip, idesc, n_in = _segs(inputs)
    op, odesc, n_out = _segs(outs)
    Cin = sum(t.shape[3] for t in inputs)
    assert sum(t.shape[3] for t in outs) == Cout and outs[0].shape[1] == Hin and outs[0].shape[2] == Win
    rc = lib().tmg_conv_wino_fwd3(ip, idesc, c_i64(n_in), _ptr(U), _ptr(bias), op, odesc, c_i64(n_out),
                                  _i64(B, Hin, Win, Cin, Cout, relu_in, pad_rep), _stream())
    if rc == -100:
        return False
    _chk(rc, "tmg_conv_wino_fwd3")
    return True


def conv_wino_narrow(inputs, U, Cout, outs, bias=None, relu_in=False, pad_rep=False, relu_out=False):
    """Few output channels: outs = list of <= 3 NHWC tensors / channel-slice views forming the Cout channels.  False when the
    shape is outside the kernel's envelope (nothing was launched)."""
    B, Hin, Win, _ = inputs[0].shape
    ip, idesc, n_in = _segs(inputs)
    op, odesc, n_out = _segs(outs)
    Cin = sum(t.shape[3] for t in inputs)
    assert sum(t.shape[3] for t in outs) == Cout
    rc = lib().tmg_conv_wino_narrow(ip, idesc, c_i64(n_in), _ptr(U), _ptr(bias), op, odesc, c_i64(n_out),
                                    _i64(B, Hin, Win, Cin, Cout, relu_in, pad_rep, relu_out), _stream())
    if rc == -100:
        return False
    _chk(rc, "tmg_conv_wino_narrow")
    return True


def conv_wino_fwd(inputs, U, Cout, outs, bias=None, relu_in=False, pad_rep=False):
    """outs = conv3x3(pad(act(cat(inputs)))) + bias through the Winograd kernel (outs: tensor or list of <= 3 segments); False when
    the shape is outside its envelope (nothing was launched: the caller runs conv_fwd with the direct operand)."""
    if isinstance(outs, torch.Tensor):
        outs = [outs]
    B, Hin, Win, _ = inputs[0].shape
    ip, idesc, n_in = _segs(inputs)
    op, odesc, n_out = _segs(outs)
    Cin = sum(t.shape[3] for t in inputs)
    assert sum(t.shape[3] for t in outs) == Cout and outs[0].shape[1] == Hin and outs[0].shape[2] == Win
    rc = lib().tmg_conv_wino_fwd(ip, idesc, c_i64(n_in), _ptr(U), _ptr(bias), op, odesc, c_i64(n_out),
                                 _i64(B, Hin, Win, Cin, Cout, relu_in, pad_rep), _stream())
    if rc == -100:
        return False
    _chk(rc, "tmg_conv_wino_fwd")
    return True


def conv3x3_auto(inputs, weight, Cout, outs, bias=None, relu_in=False, pad_rep=False, relu_out=False, dgrad=False, nvalid=0):
    """A plain 3x3 / stride-1 contraction (no kappa / add / accumulate) through the fastest kernel that takes the shape: Winograd
    (wide or narrow) when eligible, the direct implicit GEMM otherwise.  dgrad: the input gradient of a conv with this weight
    ([K = weight.shape[0] dy channels] -> the first `nvalid` (0: all) input channels).  Returns the direct mode-1 operand when it
    had to pack one (callers of replicate-padded convs reuse it for the border fold), else None."""
    Cin = sum(t.shape[3] for t in inputs)
    mode = 1 if dgrad else 0
    if not relu_out and wino_eligible(Cin, Cout, 3, 1):
        # bf16x3 (opt-in): when its kernel declines the shape, the fp32 wide Winograd kernel is next - not the narrow / direct ones
        # (a silent slowdown)
        if (_WINO_PRECISION[0] == "bf16x3"
                and conv_wino_fwd3(inputs, conv_wino_pack3(weight, mode, nvalid), Cout, outs, bias=bias, relu_in=relu_in, pad_rep=pad_rep)):
            return None
        if conv_wino_fwd(inputs, conv_wino_pack(weight, mode, nvalid), Cout, outs, bias=bias, relu_in=relu_in, pad_rep=pad_rep):
            return None
    if wino_narrow_eligible(Cin, Cout):
        if conv_wino_narrow(inputs, conv_wino_pack(weight, mode, nvalid), Cout, outs, bias=bias, relu_in=relu_in, pad_rep=pad_rep,
                            relu_out=relu_out):
            return None
    wpk = conv_pack(weight, mode, nvalid, (nvalid, 1 << 30, 0)) if (dgrad and nvalid) else conv_pack(weight, mode)
    conv_fwd(inputs, wpk, Cout, 3, 1, outs, bias=bias, relu_in=relu_in, pad_rep=pad_rep, relu_out=relu_out)
    return wpk


_WS = {}


def workspace(nfloats, device):
    """Grow-only scratch buffer per (device, stream): kernels on one stream run in order, so sharing it is safe."""
    if torch.cuda.is_current_stream_capturing():
        # a captured launch must not write into a buffer that a later eager call may outgrow and free: scratch of the graph's own pool
        return torch.empty(max(int(nfloats), 1), device=device, dtype=torch.float32)
    key = (device.index, torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    buf = _WS.get(key)
    if buf is None or buf.numel() < nfloats:
        buf = torch.empty(max(int(nfloats), 1 << 20), device=device, dtype=torch.float32)
        _WS[key] = buf
    return buf


def conv_wgrad(inputs, dy, dW, dbias, ksize, stride, kappa=None, in_scale=None, in_shift=None, relu_in=False, pad_rep=False,
               use_ws=True, cin_dst=0, cin_valid=0, ci_split=0, ci_off0=0, ci_off1=0):
    B, Hin, Win, _ = inputs[0].shape
    _, Hout, Wout, Cout = dy.shape
    ip, idesc, n_in = _segs(inputs)
    Cin = sum(t.shape[3] for t in inputs)
    if (ksize == 3 and stride == 1 and kappa is None and in_scale is None and use_ws and Cin >= 32
            and (os.environ.get("TMG_WINO_WGRAD_ALL") is not None or Cout >= 128 or (Cout >= 32 and Cin >= int(os.environ.get("TMG_WW_CIN_MIN", 32)) and B * Hin * Win >= int(os.environ.get("TMG_WW_PIX_MIN", 1 << 20))))
            and os.environ.get("TMG_NO_WINOGRAD") is None and os.environ.get("TMG_NO_WINOGRAD_WGRAD") is None
            # Winograd F(3x3, 2x2), 2.25x fewer matrix-core operations.  Measured per call site at config M: it wins wherever the
            # contraction is matrix-pipe bound (many output channels, or many pixels x input channels) and loses to the direct kernel
            # on the small ones (its 16-position slabs make the reduce step the larger part)
            and conv_wino_wgrad(inputs, dy, dW, dbias, relu_in, pad_rep, cin_dst, cin_valid, ci_split, ci_off0, ci_off1)):
        return
    dims = _i64(B, Hin, Win, Hout, Wout, ksize, stride, Cin, Cout, relu_in, pad_rep, cin_dst, cin_valid, ci_split, ci_off0, ci_off1)
    ws = workspace(lib().tmg_conv_wgrad_ws_floats(dims), dy.device) if use_ws else None
    _chk(lib().tmg_conv_wgrad(ip, idesc, c_i64(n_in), _ptr(in_scale), _ptr(in_shift), _ptr(dy), _d2(dy), _ptr(dW), _ptr(dbias),
                              _ptr(kappa), _ptr(ws), c_i64(ws.numel() if ws is not None else 0), dims, _stream()), "tmg_conv_wgrad")


def wino_wgrad_eligible(Cin, Cout):
    """conv_wgrad's routing rule for the Winograd weight-gradient kernel at large pixel counts (see conv_wgrad)."""
    return (Cin >= 32 and Cout >= 128 and os.environ.get("TMG_NO_WINOGRAD") is None and os.environ.get("TMG_NO_WINOGRAD_WGRAD") is None)


def conv_wino_wgrad(inputs, dy, dW, dbias, relu_in=False, pad_rep=False, cin_dst=0, cin_valid=0, ci_split=0, ci_off0=0, ci_off1=0):
    """3x3 / stride-1 weight gradient as Winograd F(3x3, 2x2) (tmg_conv_wino_wgrad); False when the shape is outside the kernel's
    envelope (nothing was launched).  conv_wgrad routes the large contractions here."""
    B, Hin, Win, _ = inputs[0].shape
    Cout = dy.shape[3]
    ip, idesc, n_in = _segs(inputs)
    Cin = sum(t.shape[3] for t in inputs)
    wd = _i64(B, Hin, Win, Cin, Cout, relu_in, pad_rep, cin_dst, cin_valid, ci_split, ci_off0, ci_off1)
    need = lib().tmg_conv_wino_wgrad_ws_floats(wd)
    if need > 0:
        ws = workspace(need, dy.device)
        rc = lib().tmg_conv_wino_wgrad(ip, idesc, c_i64(n_in), _ptr(dy), _d2(dy), _ptr(dW), _ptr(dbias), _ptr(ws), c_i64(ws.numel()), wd, _stream())
        if rc != -100:
            _chk(rc, "tmg_conv_wino_wgrad")
            return True
    return False


_GTAB = {}


def _segment_table(rows, device):
    """Device copy of a pointer table (the grouped launches' segment tables, tmg_level_pack's parameter table).  The values travel as
    the ARGUMENTS of a fill kernel (tmg_fill_i64, 256 int64 per launch): asynchronous, in stream order, legal during hipGraph capture.
    Round 5 copied the table from pageable host memory on a cache miss - a synchronising copy: one per grouped launch whenever the
    activations' addresses changed, i.e. on every time-step of an eager BPTT window (its T sets of activations are all alive), and with
    round 6's parameter tables in the FORWARD pass it cost the host-bound eager window its run-ahead (0.49 -> 0.53 s per window).
    In a steady single-step loop the caching allocator hands out the same blocks step after step: the table recurs and its device copy
    is reused (no launch at all).  The cache is bounded; a miss is one or two 3-us launches."""
    flat = [int(v) for r in rows for v in r]
    if torch.cuda.is_current_stream_capturing():
        # not cached: the table lives in the graph's pool like every other tensor of the capture
        t = torch.empty((len(rows), len(rows[0])), dtype=torch.int64, device=device)
        _chk(lib().tmg_fill_i64(_ptr(t), _i64(*flat), c_i64(len(flat)), _stream()), "tmg_fill_i64")
        return t
    key = (device.index, len(rows), tuple(flat))
    t = _GTAB.get(key)
    if t is None:
        if len(_GTAB) > 1024:
            _GTAB.clear()
        t = torch.empty((len(rows), len(rows[0])), dtype=torch.int64, device=device)
        _chk(lib().tmg_fill_i64(_ptr(t), _i64(*flat), c_i64(len(flat)), _stream()), "tmg_fill_i64")
        _GTAB[key] = t
    return t


def layer_planes(src):
    """[B,H,W,2K] contiguous -> [K,B,H,W,2]: one pixel-contiguous float2 plane per layer (tmg_layer_planes)."""
    B, Hh, Ww, CP = src.shape
    assert src.is_contiguous() and CP % 4 == 0
    dst = torch.empty((CP // 2, B, Hh, Ww, 2), device=src.device, dtype=torch.float32)
    _chk(lib().tmg_layer_planes(_ptr(src), _ptr(dst), c_i64(B * Hh * Ww), c_i64(CP), _stream()), "tmg_layer_planes")
    return dst


def _pixel_linear(t):
    """[B,H,W,c] view whose pixels are equally spaced in memory (address = base + pixel index * stride(2) + channel)."""
    B, Hh, Ww, _ = t.shape
    return t.stride(3) == 1 and (Hh == 1 or t.stride(1) == t.stride(2) * Ww) and (B == 1 or t.stride(0) == t.stride(2) * Ww * Hh)


def conv_wgrad_grouped(group_inputs, dy, dy_group_channels, dW, dbias, ksize, stride, relu_in=False, pad_rep=False, cin_dst=0,
                       cin_valid=0, ci_split=0, ci_off0=0, ci_off1=0, group_dy=None):
    """One launch for len(group_inputs) identically shaped weight gradients.  group_inputs[g]: list of <= 3 NHWC segments;
    dy: [B,H,W,>= G*Cg] with group g at channels [g*Cg, (g+1)*Cg), or None with group_dy = list of G tensors [B,H,W,Cg]
    of identical layout; dW: [G, Cg, cin_dst, k, k] contiguous; dbias: [G, Cg] or None.  Returns False when the library
    cannot group this shape (nothing was launched)."""
    dy_pairs = group_dy is not None and isinstance(group_dy[0], (tuple, list))     # every group's dy as (half 1, half 2) tensors
    if dy_pairs:
        # only the streaming 1x1 kernel below reads an upstream gradient in two halves
        if not (ksize == 1 and stride == 1 and int(dy_group_channels) in (16, 32) and os.environ.get("TMG_NO_MIX_WGRAD_KERNEL") is None):
            group_dy = [torch.cat(list(t), 3) for t in group_dy]
            dy_pairs = False
    if dy is None:
        dy = group_dy[0][0] if dy_pairs else group_dy[0]
        if dy_pairs:
            assert all(a.shape == dy.shape and b.shape == dy.shape for a, b in group_dy)     # (strides per group: from the table)
        else:
            assert all(t.shape == dy.shape and t.stride() == dy.stride() for t in group_dy)
    G = len(group_inputs)
    first = group_inputs[0]
    B, Hin, Win, _ = first[0].shape
    _, Hout, Wout, _ = dy.shape
    Cg = int(dy_group_channels)
    ip, idesc, n_in = _segs(first)
    Cin = sum(t.shape[3] for t in first)
    rows = []
    for segs in group_inputs:
        # same shapes in every group; the pixel strides may differ (a group's row of the device table carries its own: the first and
        # last layer of a level's node address channel-slice views of [.., C] tensors, the others [.., C/2] tensors of their own)
        assert len(segs) == n_in and all(a.shape == b.shape and a.stride(2) % 4 == b.stride(2) % 4 for a, b in zip(segs, first))
        row = []
        for t in segs:
            row += list(seg(t))  # (pointer incl. the view's channel offset, pixel stride, 0, channels)
        row += [0, 0, 0, 0] * (3 - n_in)
        if group_dy is None:
            row += [0, 0, 0, 0]
        elif dy_pairs:
            a, b = group_dy[len(rows)]
            row += [seg(a)[0], seg(a)[1], seg(b)[0], seg(b)[1]]
        else:
            row += [seg(group_dy[len(rows)])[0], seg(group_dy[len(rows)])[1], 0, 0]
        rows.append(row)
    gtab = _segment_table(rows, dy.device)
    if (group_dy is None and ksize == 3 and stride == 1 and Cin >= 20 and Cg >= 32 and os.environ.get("TMG_NO_WINOGRAD") is None
            and os.environ.get("TMG_NO_WINOGRAD_WGRAD") is None):
        # the wide levels' per-layer zero-conv weight gradients: Winograd F(3x3, 2x2), all layers of the level in one launch
        wd = _i64(B, Hin, Win, Cin, Cg, relu_in, pad_rep, cin_dst, cin_valid, ci_split, ci_off0, ci_off1)
        need = lib().tmg_conv_wino_wgrad_grouped_ws_floats(wd, c_i64(G))
        if need > 0:
            ws = workspace(need, dy.device)
            gd = _i64(Cg, dW[0].numel(), dbias[0].numel() if dbias is not None else 0)
            rc = lib().tmg_conv_wino_wgrad_grouped(ip, idesc, c_i64(n_in), _ptr(gtab), c_i64(G), gd, _ptr(dy), _d2(dy), _ptr(dW), _ptr(dbias),
                                                   _ptr(ws), c_i64(ws.numel()), wd, _stream())
            if rc != -100:
                _chk(rc, "tmg_conv_wino_wgrad_grouped")
                return True
    if (group_dy is not None and ksize == 1 and stride == 1 and Cg == Cin and Cin in (16, 32) and cin_dst in (0, Cin)
            and cin_valid in (0, Cin) and ci_split == 0 and os.environ.get("TMG_NO_MIX_WGRAD_KERNEL") is None
            and all(_pixel_linear(t) for segs in group_inputs for t in segs)
            and all(_pixel_linear(t) for e in group_dy for t in (e if dy_pairs else (e,)))):
        # the 1x1 mixes' weight gradients: streaming GEMM over the pixels, operands straight from global memory
        rc = lib().tmg_mix_wgrad_grouped(_ptr(gtab), c_i64(G), _ptr(dW), _ptr(dbias), _i64(B * Hin * Win, Cin), _stream())
        if rc != -100:
            _chk(rc, "tmg_mix_wgrad_grouped")
            return True
    if dy_pairs:    # outside the streaming kernel's envelope after all: the general kernels read one tensor per group
        return conv_wgrad_grouped(group_inputs, None, dy_group_channels, dW, dbias, ksize, stride, relu_in=relu_in, pad_rep=pad_rep,
                                  cin_dst=cin_dst, cin_valid=cin_valid, ci_split=ci_split, ci_off0=ci_off0, ci_off1=ci_off1,
                                  group_dy=[torch.cat(list(t), 3) for t in group_dy])
    if (group_dy is None and ksize == 3 and stride == 1 and Cg in (2, 4) and dW.shape[1] == 4 and not pad_rep and dbias is None and cin_dst in (0, Cin)
            and cin_valid in (0, Cin) and ci_split == 0 and Cin in (12, 20, 36, 68) and os.environ.get("TMG_NO_THIN_WGRAD") is None
            and all(t.stride(2) % 4 == 0 and t.data_ptr() % 16 == 0 for segs in group_inputs for t in segs)):
        # four output rows per group (the growth-1 layers; dy: (dd1, dd2, 0, 0) quads or compact (dd1, dd2) pairs): 4x4x1 MFMA blocks
        # instead of 16x16 tiles that would be 2/16 used
        rc = lib().tmg_conv_wgrad_thin_grouped(_ptr(gtab), c_i64(G), _i64(*[t.shape[3] for t in first]), c_i64(n_in), _ptr(dy),
                                               c_i64(dy.stride(2)), _ptr(dW), _i64(B, Hin, Win, Cin, relu_in, Cg), _stream())
        if rc != -100:
            _chk(rc, "tmg_conv_wgrad_thin_grouped")
            return True
    dims = _i64(B, Hin, Win, Hout, Wout, ksize, stride, Cin, Cg, relu_in, pad_rep, cin_dst, cin_valid, ci_split, ci_off0, ci_off1)
    ws = workspace(lib().tmg_conv_wgrad_grouped_ws_floats(dims, c_i64(G)), dy.device)
    gd = _i64(Cg, dW[0].numel(), dbias[0].numel() if dbias is not None else 0)
    rc = lib().tmg_conv_wgrad_grouped(ip, idesc, c_i64(n_in), _ptr(gtab), c_i64(G), gd, _ptr(dy), _d2(dy), _ptr(dW), _ptr(dbias),
                                      _ptr(ws), c_i64(ws.numel()), dims, _stream())
    if rc == -100:
        return False
    _chk(rc, "tmg_conv_wgrad_grouped")
    return True


def _halves(t):
    """(half 1, half 2) of an activation given as ONE [B,H,W,C] tensor (channel-slice views of it) or as a pair of [B,H,W,C/2]
    tensors (the split layout of the narrow levels)."""
    if isinstance(t, (tuple, list)):
        a, b = t
        assert a.shape == b.shape
        return a, b
    ch = t.shape[3] // 2
    return t[..., :ch], t[..., ch:]


def coupling_fwd(x, out, rsave, y2save, D, hc, wz, bz, kappa, Wm, bm, logdet, reverse, wz_d1col):
    """Zero conv + affine coupling + log-det (+ trailing channel mix) of one coupling layer in one launch (tmg_coupling_fwd_halves).
    x / out: [B,H,W,C] NHWC (or channel-slice views), or PAIRS of [B,H,W,C/2] tensors (channel halves in tensors of their own);
    D: [B,H,W,4] from c1x2_fwd; hc: [B,H,W,C] view of the level's conditioning contribution.  Returns False when the shape is
    outside the kernel's envelope (nothing was launched)."""
    x1, x2 = _halves(x)
    o1, o2 = _halves(out)
    B, Hh, Ww, ch = x1.shape
    s1, s2, t1, t2, sh = seg(x1), seg(x2), seg(o1), seg(o2), seg(hc)
    assert rsave.is_contiguous() and D.is_contiguous() and (y2save is None or y2save.is_contiguous()) and wz.is_contiguous()
    dims = _i64(B, Hh, Ww, 2 * ch, 1 if reverse else 0, s1[1], t1[1], sh[1], wz.shape[1], wz_d1col, s2[1], t2[1])
    rc = lib().tmg_coupling_fwd_halves(c_vp(s1[0]), c_vp(s2[0]), c_vp(t1[0]), c_vp(t2[0]), _ptr(rsave), _ptr(y2save), _ptr(D), c_vp(sh[0]),
                                       _ptr(wz), _ptr(bz), _ptr(kappa), _ptr(Wm), _ptr(bm), _ptr(logdet), dims, _stream())
    if rc == -100:
        return False
    _chk(rc, "tmg_coupling_fwd_halves")
    return True


def coupling_bwd(dout, x2, r, g, Wm, wz, kappa, DH, dtin, G0, GD, wz_d1col, fwd=False):
    """Mix input gradient + affine-coupling backward + zero-conv input gradient (exact replicate adjoint) of one generative-
    direction coupling layer in one launch (tmg_coupling_bwd_halves).  dout / dtin: [B,H,W,C] tensors or pairs of [B,H,W,C/2] halves;
    x2: the SECOND half of the layer input [B,H,W,C/2] (a channel-slice view or a tensor of its own); DH: [B,H,W,C] channel-slice
    view of the level's stash.  fwd=True: the density direction's layer (mix -> coupling): dout = gradient w.r.t. the coupling
    output, x2 = second half of the coupling OUTPUT, no mix in front (Wm is not read)."""
    d1, d2 = _halves(dout)
    t1, t2 = _halves(dtin)
    B, Hh, Ww, ch = d1.shape
    assert x2.shape[3] == ch
    sd1, sd2, sx, sh, st1, st2 = seg(d1), seg(d2), seg(x2), seg(DH), seg(t1), seg(t2)
    assert r.is_contiguous() and G0.is_contiguous() and GD.is_contiguous() and Wm.is_contiguous() and wz.is_contiguous()
    dims = _i64(B, Hh, Ww, 2 * ch, sd1[1], sx[1], sh[1], st1[1], wz.shape[1], wz_d1col, sd2[1], st2[1], 1 if fwd else 0)
    rc = lib().tmg_coupling_bwd_halves(c_vp(sd1[0]), c_vp(sd2[0]), c_vp(sx[0]), _ptr(r), _ptr(g), _ptr(Wm), _ptr(wz), _ptr(kappa), c_vp(sh[0]),
                                       c_vp(st1[0]), c_vp(st2[0]), _ptr(G0), _ptr(GD), dims, _stream())
    if rc == -100:
        return False
    _chk(rc, "tmg_coupling_bwd_halves")
    return True


def mix_f16(x, W, bias, y, transposed=False):
    """y = fp16(W) . fp16(x) + bias per pixel, fp32 accumulation (tmg_mix_f16); W: fp32 [C, C]; transposed: apply W^T."""
    B, Hh, Ww, C = x.shape
    check_act(W)
    assert W.is_contiguous() and W.shape == (C, C) and y.shape == x.shape
    _chk(lib().tmg_mix_f16(_ptr(x), _d2(x), _ptr(W), _ptr(bias), _ptr(y), _d2(y), _i64(B * Hh * Ww, C, 1 if transposed else 0), _stream()),
         "tmg_mix_f16")


def mat_inverse(W, b=None):
    """(W^-1, -W^-1 b) of K channel mixes [K, C, C] / [K, C] (fp64 arithmetic on the device, rounded once; C <= 64)."""
    K, C = W.shape[0], W.shape[1]
    W = W.contiguous()
    Winv = torch.empty_like(W)
    binv = torch.empty((K, C), device=W.device, dtype=torch.float32) if b is not None else None
    _chk(lib().tmg_mat_inverse(_ptr(W), _ptr(b.contiguous() if b is not None else None), _ptr(Winv), _ptr(binv), _i64(K, C), _stream()),
         "tmg_mat_inverse")
    return Winv, binv


def lu_fold_fwd(tab, sign_s, perm, iperm, W, Wm, bm, ld, reverse, sgn, hw):
    K, C = sign_s.shape
    assert W.dtype == torch.float64 and C <= 256
    _chk(lib().tmg_lu_fold_fwd(_ptr(tab), _ptr(sign_s), _ptr(perm), _ptr(iperm), _ptr(W), _ptr(Wm), _ptr(bm), _ptr(ld), _i64(K, C, reverse),
                               _flts([sgn, hw]), _stream()), "tmg_lu_fold_fwd")


def lu_fold_bwd(tab, sign_s, perm, iperm, W, dWm, dbm, dld, dl, du, dlogs, da, db, reverse, sgn, hw, dWm_tail=None, dbm_tail=None):
    """dWm_tail / dbm_tail: upstream gradients of the LAST layer in tensors of their own (dWm / dbm then cover layers 0..K-2)."""
    K, C = sign_s.shape
    _chk(lib().tmg_lu_fold_bwd_split(_ptr(tab), _ptr(sign_s), _ptr(perm), _ptr(iperm), _ptr(W), _ptr(dWm), _ptr(dbm), _ptr(dWm_tail),
                                     _ptr(dbm_tail), _ptr(dld), _ptr(dl), _ptr(du), _ptr(dlogs), _ptr(da), _ptr(db), _i64(K, C, reverse),
                                     _flts([sgn, hw]), _stream()), "tmg_lu_fold_bwd_split")


def level_finish(Wz, dWz, Bz, dBz, Kp, tmpX, tmpC, dW1, dW2, dK, ws, ch, Cc):
    """Parameter-gradient epilogue of a level's fused coupling node in one launch (tmg_level_finish): d(kappa) of all NL zero convs
    (fp64 inner products) and the scatter of the grouped weight-gradient rows tmpX / tmpC into dW1 / dW2.  ws: 4 * NL zeroed floats."""
    NL, C = dBz.shape
    for t in (Wz, dWz, Bz, dBz, Kp, tmpX, tmpC, dW1, dW2, dK, ws):
        assert t is None or t.is_contiguous()
    assert ws.numel() >= 4 * NL
    _chk(lib().tmg_level_finish(_ptr(Wz), _ptr(dWz), _ptr(Bz), _ptr(dBz), _ptr(Kp), _ptr(tmpX), _ptr(tmpC), _ptr(dW1), _ptr(dW2), _ptr(dK),
                                _ptr(ws), _i64(NL, C, ch, Cc, tmpC.shape[1] if tmpC is not None else 4), _stream()), "tmg_level_finish")


def adam_step(tab, chunks, nchunks, lr, b1, b2, eps, wd, bc1, bc2s, amsgrad):
    """One launch for the Adam / AMSGrad update of every parameter (tmg_adam_step): tab int64 [n][5] pointers, chunks int32 [nchunks][3]."""
    _chk(lib().tmg_adam_step(_ptr(tab), _ptr(chunks), _i64(nchunks, 1 if amsgrad else 0), _flts([lr, b1, b2, eps, wd, bc1, bc2s, 1.0 - b1, 1.0 - b2]), _stream()),
         "tmg_adam_step")


def mix_f32(x, W, bias, y, transposed=False):
    """y = W x + bias per pixel (W^T when `transposed`: the input gradient of the same mix) in fp32 on the matrix cores, for
    C <= 128 (tmg_mix_f32): the stand-alone 1x1 mixes of the wide levels / ConvLSTM blocks.  False when C is outside the envelope."""
    B, Hh, Ww, C = x.shape
    if C > 128 or C % 4:
        return False
    _chk(lib().tmg_mix_f32(_ptr(x), _d2(x), _ptr(W), _ptr(bias), _ptr(y), _d2(y), _i64(B * Hh * Ww, C, 1 if transposed else 0), _stream()),
         "tmg_mix_f32")
    return True


def _d2o(t):
    """{pixel stride, 0}: the channel offset of a slice view is already folded into its data pointer."""
    return _d2(t)


def mix_affine_fwd(x, hh, W, bias, y, r, y2, logdet):
    """Coupling (generative direction) + trailing mix in one launch (tmg_mix_f32_affine_fwd).  False outside the envelope."""
    B, Hh, Ww, C = x.shape
    rc = lib().tmg_mix_f32_affine_fwd(_ptr(x), _d2o(x), _ptr(hh), _d2o(hh), _ptr(W), _ptr(bias), _ptr(y), _d2o(y), _ptr(r), _ptr(y2),
                                      _ptr(logdet), _i64(B * Hh * Ww, C, Hh * Ww), _stream())
    if rc == -100:
        return False
    _chk(rc, "tmg_mix_f32_affine_fwd")
    return True


def mix_affine_bwd(dy, W, r, t2, g, kappa, dto1, dtin2, dhh):
    """Input gradient of the mix + the coupling's backward in one launch (tmg_mix_f32_affine_bwd).  False outside the envelope."""
    B, Hh, Ww, C = dy.shape
    rc = lib().tmg_mix_f32_affine_bwd(_ptr(dy), _d2o(dy), _ptr(W), _ptr(r), _ptr(t2), _d2o(t2), _ptr(g), _ptr(kappa), _ptr(dto1), _ptr(dtin2),
                                      _d2o(dtin2), _ptr(dhh), _d2o(dhh), _i64(B * Hh * Ww, C, Hh * Ww), _stream())
    if rc == -100:
        return False
    _chk(rc, "tmg_mix_f32_affine_bwd")
    return True


def conv_rep_border_fix(dy, w, outs, kappa=None):
    B, H, W, Cdy = dy.shape
    op, odesc, n_out = _segs(outs)
    Cx = sum(t.shape[3] for t in outs)
    _chk(lib().tmg_conv_rep_border_fix(_ptr(dy), _d2(dy), _ptr(w), _ptr(kappa), op, odesc, c_i64(n_out), _i64(B, H, W, Cdy, Cx),
                                       _stream()), "tmg_conv_rep_border_fix")


def conv_dgrad_direct(dy, w, dx, ksize, stride, accumulate=False):
    B, Hin, Win, Cin = dx.shape
    _, Hout, Wout, Cout = dy.shape
    _chk(lib().tmg_conv_dgrad_direct(_ptr(dy), _d2(dy), _ptr(w), _ptr(dx), _d2(dx),
                                     _i64(B, Hin, Win, Hout, Wout, Cin, Cout, ksize, stride, accumulate), _stream()),
         "tmg_conv_dgrad_direct")


# ------------------------------------------------------------------------------------------------
# bandwidth-bound ops
# ------------------------------------------------------------------------------------------------
def affine_apply(hh, x2, y2, rsave, logdet, reverse, x1=None, y1=None):
    """x1 / y1 (optional, Ch channels each): pass-through half copied in the same launch."""
    B, H, W, Ch = x2.shape
    if x1 is None:
        _chk(lib().tmg_affine_apply(_ptr(hh), _d2(hh), _ptr(x2), _d2(x2), _ptr(y2), _d2(y2), _ptr(rsave), _ptr(logdet),
                                    _i64(B, H * W, Ch, reverse), _stream()), "tmg_affine_apply")
    else:
        assert x1.shape == x2.shape and y1.shape == x2.shape
        _chk(lib().tmg_affine_apply_pass(_ptr(hh), _d2(hh), _ptr(x2), _d2(x2), _ptr(y2), _d2(y2), _ptr(rsave), _ptr(logdet), _ptr(x1),
                                         _d2(x1), _ptr(y1), _d2(y1), _i64(B, H * W, Ch, reverse), _stream()), "tmg_affine_apply_pass")


def affine_bwd(gout, yref, rsave, g, gin, dhh, reverse, kappa=None):
    """kappa given: dhh is written pre-multiplied by exp(clamp(kappa))."""
    B, H, W, Ch = gout.shape
    _chk(lib().tmg_affine_bwd_scaled(_ptr(gout), _d2(gout), _ptr(yref), _d2(yref), _ptr(rsave), _ptr(g), _ptr(gin), _d2(gin), _ptr(dhh),
                                     _d2(dhh), _ptr(kappa), _i64(B, H * W, Ch, reverse), _stream()), "tmg_affine_bwd_scaled")


def lstm_pointwise_fwd(gates, c_prev, c_next, h_next):
    B, H, W, R4 = gates.shape
    cd = _d2(c_prev) if c_prev is not None else _i64(0, 0)
    _chk(lib().tmg_lstm_pointwise_fwd(_ptr(gates), _ptr(c_prev), cd, _ptr(c_next), _ptr(h_next), _i64(B * H * W, R4 // 4), _stream()),
         "tmg_lstm_pointwise_fwd")


def lstm_pointwise_bwd(acts, c_prev, c_next, dh, dc_in, dc_prev):
    B, H, W, R4 = acts.shape
    cd = _d2(c_prev) if c_prev is not None else _i64(0, 0)
    _chk(lib().tmg_lstm_pointwise_bwd(_ptr(acts), _ptr(c_prev), cd, _ptr(c_next), _ptr(dh), _ptr(dc_in), _ptr(dc_prev),
                                      _i64(B * H * W, R4 // 4), _stream()), "tmg_lstm_pointwise_bwd")


def _fl(vals):
    return (ctypes.c_float * 4)(*vals)


def gauss_fwd(hz, zin, zout, logp, mode, clip_mean, limits):
    B, H, W, Ch = zin.shape
    zo = _d2(zout) if zout is not None else _i64(0, 0)
    _chk(lib().tmg_gauss_fwd(_ptr(hz), _d2(hz), _ptr(zin), _d2(zin), _ptr(zout), zo, _ptr(logp), _i64(B, H * W, Ch, mode, clip_mean),
                             _fl(limits), _stream()), "tmg_gauss_fwd")


def gauss_bwd(hz, zin, dzin, g, dzout, dhz, mode, clip_mean, limits):
    B, H, W, Ch = zin.shape
    di = _d2(dzin) if dzin is not None else _i64(0, 0)
    do = _d2(dzout) if dzout is not None else _i64(0, 0)
    _chk(lib().tmg_gauss_bwd(_ptr(hz), _d2(hz), _ptr(zin), _d2(zin), _ptr(dzin), di, _ptr(g), _ptr(dzout), do, _ptr(dhz), _d2(dhz),
                             _i64(B, H * W, Ch, mode, clip_mean), _fl(limits), _stream()), "tmg_gauss_bwd")


def gauss_sample(hz, eps, z1, out, logp, clip_mean, limits, eps_out=None, nonce=None, site=0):
    """out[..., C - Ch:] = mean + exp(log-std) eps, out[..., :Ch] = z1 (when given): the Gaussian sample written straight into the
    level's activation (tmg_gauss_sample).  eps None: drawn in the kernel from (nonce, site) and stored to eps_out."""
    B, Hh, Ww, Co = out.shape
    Ch = hz.shape[3] // 2
    off = Co - Ch
    zd = lambda t: _d2(t) if t is not None else _i64(0, 0)  # noqa: E731
    od = seg(out)
    _chk(lib().tmg_gauss_sample(_ptr(hz), _d2(hz), _ptr(eps), zd(eps), _ptr(z1), zd(z1), _ptr(out), _i64(od[1], off), c_i64(0), _ptr(eps_out),
                                _ptr(logp), _ptr(nonce), _i64(B, Hh * Ww, Ch, clip_mean, site), _fl(limits), _stream()), "tmg_gauss_sample")


def reverse_loss_fwd(y, ld, loss, s1, s2):
    _chk(lib().tmg_reverse_loss_fwd(_ptr(y), _ptr(ld), _ptr(loss), _i64(y.numel(), ld.numel()), _flts([s1, s2]), _stream()), "tmg_reverse_loss_fwd")


def reverse_loss_bwd(y, g, dy, dld, s1, s2):
    _chk(lib().tmg_reverse_loss_bwd(_ptr(y), _ptr(g), _ptr(dy), _ptr(dld), _i64(y.numel(), dld.numel()), _flts([s1, s2]), _stream()), "tmg_reverse_loss_bwd")


SUM_TERMS_MAX = 8


def sum_terms(terms, out):
    """out[b] = sum of the terms (fp32 vectors of out's length, or one-element tensors: broadcast) in one launch (tmg_sum_terms)."""
    n = len(terms)
    ptrs = (c_vp * n)(*[t.data_ptr() for t in terms])
    _chk(lib().tmg_sum_terms(ptrs, _i64(*[t.numel() for t in terms]), c_i64(n), _ptr(out), c_i64(out.numel()), _stream()), "tmg_sum_terms")


def vec_sum(g, out):
    _chk(lib().tmg_vec_sum(_ptr(g), c_i64(g.numel()), _ptr(out), _stream()), "tmg_vec_sum")


def spread2(dy, up):
    """up (contiguous [B,H,W,C]) = dy on the even positions, zeros elsewhere, in one launch (tmg_spread2); False outside the envelope."""
    B, Hh, Ww, C = up.shape
    rc = lib().tmg_spread2(_ptr(dy), _d2(dy), _ptr(up), _i64(B, Hh, Ww, dy.shape[1], dy.shape[2], C), _stream())
    if rc == -100:
        return False
    _chk(rc, "tmg_spread2")
    return True


def level_pack(tab, Wz, Wcat, Bz, Kp, NL, NLp, C, ch, Cc):
    """Stacked / sliced parameter operands of a level node from the modules' own tensors (tmg_level_pack; tab from _segment_table)."""
    _chk(lib().tmg_level_pack(_ptr(tab), _ptr(Wz), _ptr(Wcat), _ptr(Bz), _ptr(Kp), _i64(NL, NLp, C, ch, Cc), _stream()), "tmg_level_pack")


def checker(src, dst, to_small):
    if to_small:
        B, h, w, _ = dst.shape
        C = src.shape[3]
    else:
        B, h, w, _ = src.shape
        C = dst.shape[3]
    _chk(lib().tmg_checker(_ptr(src), _d2(src), _ptr(dst), _d2(dst), _i64(B, h, w, C, to_small), _stream()), "tmg_checker")


def pad_halves(src, dst, ch, pad, to_padded):
    """compact [B,H,W,2 ch] <-> zero-padded [B,H,W,2 (ch + pad)] = [x1 | 0.. | x2 | 0..] (one launch, zeros included)."""
    B, Hh, Ww, _ = src.shape
    assert src.shape[3] == (2 * ch if to_padded else 2 * (ch + pad)) and dst.shape[3] == (2 * (ch + pad) if to_padded else 2 * ch)
    _chk(lib().tmg_pad_halves(_ptr(src), _d2(src), _ptr(dst), _d2(dst), _i64(B * Hh * Ww, ch, pad, 1 if to_padded else 0), _stream()),
         "tmg_pad_halves")


def upsample_fwd(src, dst):
    B, hi, wi, C = src.shape
    _, ho, wo, _ = dst.shape
    assert src.is_contiguous() and dst.is_contiguous()
    _chk(lib().tmg_upsample_fwd(_ptr(src), _ptr(dst), _i64(B, hi, wi, ho, wo, C), _stream()), "tmg_upsample_fwd")


def upsample_bwd(dout, din):
    B, hi, wi, C = din.shape
    _, ho, wo, _ = dout.shape
    assert dout.is_contiguous() and din.is_contiguous()
    _chk(lib().tmg_upsample_bwd(_ptr(dout), _ptr(din), _i64(B, hi, wi, ho, wo, C), _stream()), "tmg_upsample_bwd")


def chan_reduce(x, g, v0, v1, v2, v3, s0, s1, mode, divisor=0):
    """mode 0 with divisor > 0: v0 holds channel SUMS and v0/divisor is subtracted (centred second pass)."""
    B, H, W, C = x.shape
    gd = _d2(g) if g is not None else _i64(0, 0)
    _chk(lib().tmg_chan_reduce(_ptr(x), _d2(x), _ptr(g), gd, _ptr(v0), _ptr(v1), _ptr(v2), _ptr(v3), _ptr(s0), _ptr(s1),
                               _i64(B * H * W, C, mode, divisor), _stream()), "tmg_chan_reduce")


def bn_bwd_apply(x, g, a, bsh, mean, rstd, gamma, m0, m1, dx, accumulate, divisor=0):
    """divisor > 0: m0, m1 are sums and are divided by it inside the kernel."""
    B, H, W, C = x.shape
    _chk(lib().tmg_bn_bwd_apply(_ptr(x), _d2(x), _ptr(g), _d2(g), _ptr(a), _ptr(bsh), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(m0),
                                _ptr(m1), _ptr(dx), _d2(dx), _i64(B * H * W, C, accumulate, divisor), _stream()), "tmg_bn_bwd_apply")


def chan_moments(x, acc64):
    """One-pass per-channel sum / sum of squares of an NHWC tensor or channel-slice view into acc64 (double [2, C], zeroed)."""
    B, H, W, C = x.shape
    assert acc64.dtype == torch.float64 and acc64.numel() == 2 * C
    _chk(lib().tmg_chan_moments(_ptr(x), _d2(x), _ptr(acc64), _i64(B * H * W, C), _stream()), "tmg_chan_moments")


def bn_finalize64(acc64, gamma, beta, running_mean, running_var, out, n, eps, momentum, counter=None):
    """counter: the module's int64 num_batches_tracked tensor (device scalar), incremented by the same launch."""
    C = gamma.numel()
    if counter is not None:
        assert counter.dtype == torch.int64 and counter.numel() == 1 and counter.device == gamma.device
    _chk(lib().tmg_bn_finalize64(_ptr(acc64), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), _ptr(out),
                                 _i64(C, n, counter.data_ptr() if counter is not None else 0), _flts([eps, momentum]), _stream()),
         "tmg_bn_finalize64")


def bn_finalize(sums, csq, gamma, beta, running_mean, running_var, out, n, eps, momentum):
    """out: [5, C] = mean, var, rstd, a, bsh; running_* (or None) updated in place."""
    C = gamma.numel()
    _chk(lib().tmg_bn_finalize(_ptr(sums), _ptr(csq), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), _ptr(out), _i64(C, n),
                               _flts([eps, momentum]), _stream()), "tmg_bn_finalize")


def masked_add(dst, src=None, ref=None, add=None, accumulate=False):
    B, H, W, n = dst.shape
    z = _i64(0, 0)
    _chk(lib().tmg_masked_add(_ptr(src), _d2(src) if src is not None else z, _ptr(ref), _d2(ref) if ref is not None else z, _ptr(add),
                              _d2(add) if add is not None else z, _ptr(dst), _d2(dst), _i64(B * H * W, n, accumulate), _stream()),
         "tmg_masked_add")


def c1_fwd(inputs, w, out, relu_in=True, w_rows=0, fill4=False, add=None, w_split=0, w_gap=0):
    B, H, W, _ = inputs[0].shape
    ip, idesc, n_in = _segs(inputs)
    Cin = sum(t.shape[3] for t in inputs)
    ad = _d2(add) if add is not None else _i64(0, 0)
    _chk(lib().tmg_c1_fwd_add(ip, idesc, c_i64(n_in), _ptr(w), _ptr(add), ad, _ptr(out), _d2(out),
                              _i64(B, H, W, Cin, relu_in, w_rows, fill4, w_split, w_gap), _stream()), "tmg_c1_fwd_add")


def c1x2_fwd(inputs, w1, w2, out, w_rows, w2_d1_row, add1=None, add2=None, w_split=0, w_gap=0, relu_in=True):
    """Both growth-1 layers in one launch; out: [B,H,W,4] receives (d1, d2, 0, 0)."""
    B, H, W, _ = inputs[0].shape
    ip, idesc, n_in = _segs(inputs)
    Cin = sum(t.shape[3] for t in inputs)
    z = _i64(0, 0)
    _chk(lib().tmg_c1x2_fwd(ip, idesc, c_i64(n_in), _ptr(w1), _ptr(w2), _ptr(add1), _d2(add1) if add1 is not None else z, _ptr(add2),
                            _d2(add2) if add2 is not None else z, _ptr(out), _d2(out),
                            _i64(B, H, W, Cin, relu_in, w_rows, w_split, w_gap, w2_d1_row), _stream()), "tmg_c1x2_fwd")


def c1_bwd(inputs, w, dW, dd, dref, gsegs, relu_in=True):
    B, H, W, _ = inputs[0].shape
    ip, idesc, n_in = _segs(inputs)
    gp, gdesc, ng = _segs(gsegs)
    Cin = sum(t.shape[3] for t in inputs)
    rd = _d2(dref) if dref is not None else _i64(0, 0)
    _chk(lib().tmg_c1_bwd(ip, idesc, c_i64(n_in), _ptr(w), _ptr(dW), _ptr(dd), _d2(dd), _ptr(dref), rd, gp, gdesc, c_i64(ng),
                          _i64(B, H, W, Cin, relu_in), _stream()), "tmg_c1_bwd")


def dkappa(w, dw, b, db, kappa, dk):
    _chk(lib().tmg_dkappa(_ptr(w), _ptr(dw), c_i64(w.numel()), _ptr(b), _ptr(db), c_i64(b.numel()), _ptr(kappa), _ptr(dk), _stream()),
         "tmg_dkappa")


def dense2_bwd(inputs, w1p, w2p, dW1p, dW2p, GD, D, g0, outs, cin_nn, add0=None, rows1=0, rows2=0, dd1=None, dd2=None, split2=0, gap2=0,
               dd_quad=False):
    """Fused backward of the two growth-1 layers; `inputs` = nn inputs + [D]; g0 / outs: lists (<= 2) of NHWC tensors."""
    B, H, W, _ = inputs[0].shape
    ip, idesc, n_in = _segs(inputs)
    gp, gdesc, ng = _segs(g0)
    op, odesc, _ = _segs(outs)
    Cin = sum(t.shape[3] for t in inputs)
    a_stride = seg(add0)[1] if add0 is not None else 0
    _chk(lib().tmg_dense2_bwd(ip, idesc, c_i64(n_in), _ptr(w1p), _ptr(w2p), _ptr(dW1p), _ptr(dW2p), _ptr(GD), c_i64(seg(GD)[1]), _ptr(D),
                              c_i64(seg(D)[1]), gp, gdesc, op, odesc, c_i64(ng), _ptr(add0), c_i64(a_stride),
                              _i64(B, H, W, Cin, cin_nn, rows1 or Cin, rows2 or Cin, dd1.data_ptr() if dd1 is not None else 0,
                                   dd2.data_ptr() if dd2 is not None else 0, seg(dd1)[1] if dd1 is not None else 0, split2, gap2, 1 if dd_quad else 0),
                              _stream()),
         "tmg_dense2_bwd")


# ------------------------------------------------------------------------------------------------
# physics-constrained loss (row F1)
# ------------------------------------------------------------------------------------------------
def _flts(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def phys_fields(u, p, ustar, pstar, dx, dy, rho, k1, k2, scale):
    """Divergence / pressure-Poisson residual fields for 3x3 or 5x5 stencils, scaled or not (tmg_phys_fields).  u: contiguous
    [N,2,H,W]; p: contiguous [N,1,H,W] or None; ustar [N,1,H,W+2] / pstar [N,1,H,W] (either may be None)."""
    check_act(u)
    N, _, Hh, Ww = u.shape
    rc = lib().tmg_phys_fields(_ptr(u), _ptr(p), _ptr(ustar), _ptr(pstar), _i64(N, Hh, Ww, k1, k2, 1 if scale else 0), _flts([dx, dy, rho]), _stream())
    if rc == -100:
        raise ValueError('kernel_size size {:d} is not supported!'.format(k1 if k1 not in (3, 5) else k2))
    _chk(rc, "tmg_phys_fields")


def phys_fwd(y, target, sums, sd, mu, dx, dy, rho=1.0, pstar=None, ustar=None):
    """y / target: contiguous [N,3,H,W]; sums: zero-filled [3] (sum pstar^2, sum ustar^2, sum (y-target)^2)."""
    check_act(y)
    N, _, Hh, Ww = y.shape
    _chk(lib().tmg_phys_fwd(_ptr(y), _ptr(target), _ptr(sums), _ptr(pstar), _ptr(ustar), _i64(N, Hh, Ww),
                            _flts(list(sd) + list(mu) + [dx, dy, rho]), _stream()), "tmg_phys_fwd")


def phys_rms(y, trms, mean_out, coef_out, sum_out):
    """y: contiguous [B,T,3,H,W]; trms: [B,3,H,W]; sum_out: zero-filled [1]."""
    B, T = y.shape[0], y.shape[1]
    chw = y.shape[2] * y.shape[3] * y.shape[4]
    _chk(lib().tmg_phys_rms(_ptr(y), _ptr(trms), _ptr(mean_out), _ptr(coef_out), _ptr(sum_out), _i64(B, T, chw), _stream()), "tmg_phys_rms")


def phys_bwd(y, target, mean, coef, dyo, T, sd, mu, dx, dy, rho, cp, cd, cl, cr, upstream=None):
    """upstream: optional device scalar (fp32, 1 element) multiplied onto cp..cr inside the kernel."""
    N, _, Hh, Ww = y.shape
    _chk(lib().tmg_phys_bwd_dev(_ptr(y), _ptr(target), _ptr(mean), _ptr(coef), _ptr(dyo), _ptr(upstream), _i64(N, T, Hh, Ww),
                                _flts(list(sd) + list(mu) + [dx, dy, rho, cp, cd, cl, cr]), _stream()), "tmg_phys_bwd_dev")
