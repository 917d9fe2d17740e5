// Shared device helpers for the TM-Glow gfx950 kernels.  fp32, NHWC activations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16 bytes through a buffer descriptor: address = descriptor base + scalar byte offset `soff` + per-lane 32-bit byte offset `voff`.
// The hot loops read their L2-resident packed operands (weights, Winograd U) this way: a flat global load needs a 64-bit VECTOR add
// per load (v_lshl_add_u64) for the same address, and the fp32 MFMA shares the vector ALUs - measured on wino_fwd_kernel<2>, round 6:
// 2.31 -> 2.13 ms for the ConvLSTM gate conv.  The descriptor is built from kernel arguments only (wave-uniform: no waterfall loop).
// (The result must be taken as a whole vector: element-wise __builtin_bit_cast of the builtin's result made hipcc 7.2 load one dword.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tmg_make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 tmg_bload4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0));
    return make_float4(v[0], v[1], v[2], v[3]);
}

#define TMG_MAX_IN_SEG 3
#define TMG_MAX_OUT_SEG 3

// A "segment" is a channel range [off, off+n) of an NHWC tensor whose pixels are `stride` floats
// apart.  Several segments side by side stand for the channel concatenation the reference builds
// with torch.cat (flowAffine.py:74, convLSTM.py:72,151, denseBlock.py:152) without materialising it.
struct TmgSeg {
    const float* p;
    int stride;
    int off;
    int n;
};
struct TmgOSeg {
    float* p;
    int stride;
    int off;
    int n;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in thread 0
__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats of LDS */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float out_scale_of(const float* kappa) {
    if (!kappa) return 1.0f;
    float k = *kappa;
    k = fminf(fmaxf(k, -4.0f), 1.3862943611198906f);
    return expf(k);
}


// 4 consecutive (concatenated) input channels c..c+3 of pixel (b,iy,ix), after padding rule,
// optional affine and optional ReLU.
template <typename P>
__device__ __forceinline__ float4 load_in4(const P& p, int b, int iy, int ix, int c) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.pad_rep) {
        iy = min(max(iy, 0), p.Hin - 1);
        ix = min(max(ix, 0), p.Win - 1);
    } else if (iy < 0 || iy >= p.Hin || ix < 0 || ix >= p.Win) {
        return v;
    }
    if (c >= p.Cin) return v;
    const size_t pix = ((size_t)b * p.Hin + iy) * p.Win + ix;
    // segment lookup by value (no pointer into the kernel-argument struct: keeps it out of scratch)
    const int n0 = p.in[0].n, n1 = p.in[1].n;
    if (p.vec4) {
        int cl = c;
        const float* sp = p.in[0].p;
        int ss = p.in[0].stride, so = p.in[0].off;
        if (cl >= n0) {
            cl -= n0;
            sp = p.in[1].p; ss = p.in[1].stride; so = p.in[1].off;
            if (cl >= n1) {
                cl -= n1;
                sp = p.in[2].p; ss = p.in[2].stride; so = p.in[2].off;
            }
        }
        v = *reinterpret_cast<const float4*>(sp + pix * ss + so + cl);
    } else {
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int cl = c + e;
            float x = 0.f;
            if (cl < p.Cin) {
                const float* sp = p.in[0].p;
                int ss = p.in[0].stride, so = p.in[0].off;
                if (cl >= n0) {
                    cl -= n0;
                    sp = p.in[1].p; ss = p.in[1].stride; so = p.in[1].off;
                    if (cl >= n1) {
                        cl -= n1;
                        sp = p.in[2].p; ss = p.in[2].stride; so = p.in[2].off;
                    }
                }
                x = sp[pix * ss + so + cl];
            }
            t[e] = x;
        }
        v = make_float4(t[0], t[1], t[2], t[3]);
    }
    if (p.in_scale) {
        float* f = reinterpret_cast<float*>(&v);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < p.Cin) f[e] = f[e] * p.in_scale[c + e] + p.in_shift[c + e];
    }
    if (p.relu_in) {
        v.x = fmaxf(v.x, 0.f);
        v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f);
        v.w = fmaxf(v.w, 0.f);
    }
    return v;
}

// Stage channels [c0, c0+kch) of the (PH x PW) input patch whose top-left input pixel is (iy0, ix0).
// U loads are issued back to back before any is written to LDS: a thread that waits for each load before issuing
// the next one is bound by U x the memory latency (measured: 5x slower staging at U = 1).
template <int U = 8, typename P>
__device__ __forceinline__ void stage_patch(const P& p, float* lds, int b, int iy0, int ix0, int PH, int PW, int c0,
                                            int kch, int CS, int swz = 0) {
    const int k4 = kch >> 2;
    const int items = PH * PW * k4;
    const int nt = blockDim.x;
    // floor(n / d) == umulhi(n, ceil(2^32 / d)) for n * d < 2^32 (d >= 2): two multiplies instead of two divisions per item
    const unsigned mk = k4 > 1 ? 0xFFFFFFFFu / (unsigned)k4 + 1u : 0u;
    const unsigned mp = 0xFFFFFFFFu / (unsigned)PW + 1u;
    for (int base = threadIdx.x; base < items; base += nt * U) {
        float4 v[U];
        int dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = base + u * nt;
            dst[u] = -1;
            if (it < items) {
                const int pix = k4 > 1 ? (int)__umulhi((unsigned)it, mk) : it;
                const int c4 = it - pix * k4;
                const int py = (int)__umulhi((unsigned)pix, mp);
                const int px = pix - py * PW;
                v[u] = load_in4(p, b, iy0 + py, ix0 + px, c0 + 4 * c4);
                dst[u] = pix * CS + ((4 * c4) ^ ((px & 1) ? swz : 0));  // swz: XOR swizzle of odd patch columns (0 = linear)
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (dst[u] >= 0) *reinterpret_cast<float4*>(lds + dst[u]) = v[u];
    }
}

// Zero-filled global memory: lanes of a load batch that have nothing to read point here, so the batch has no control flow.
static __device__ float tmg_zero_page[64];

// The matrix-core source files are built without the packed-fp32 vector instructions (tmg_hip.NO_PACKED_F32: beside MFMAs a
// v_pk_add_f32 costs more than the two scalar additions it replaces); a kernel in such a file that is vector-ALU work, or that was
// tuned with them, keeps them with this attribute.
#if defined(__HIP_DEVICE_COMPILE__)
#define TMG_PACKED_F32 __attribute__((target("packed-fp32-ops")))
#else
#define TMG_PACKED_F32
#endif

// Loads through pointers the compiler cannot prove global - pointers that arrive as integers in a device table (the grouped launches'
// segment tables), and everything selected against them - are emitted as flat_load: those tick the LDS counter as well as the memory
// counter, and since LDS and memory return out of order every later wait for an LDS read becomes lgkmcnt(0) INCLUDING the outstanding
// flat loads - a prefetched tile is then waited for at the first fragment read of the current one (round 5: found in every
// wino_wgrad_kernel and conv_wgrad_kernel<..,true> instance).  These helpers state the address space.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifdef TMG_FLAT_LOADS      // (A/B builds only: the generic loads of rounds 1-4)
#define TMG_GAS
#else
#define TMG_GAS __attribute__((address_space(1)))
#endif
__device__ __forceinline__ float4 tmg_ldg4(const float* p) {
    const f32x4 v = *(const TMG_GAS f32x4*)p;
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float2 tmg_ldg2(const float* p) {
    const f32x2 v = *(const TMG_GAS f32x2*)p;
    return make_float2(v[0], v[1]);
}
__device__ __forceinline__ float tmg_ldg1(const float* p) { return *(const TMG_GAS float*)p; }

// ---- XCD-aware tile order (speed only, never correctness) --------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2: MI355X_MICROARCH.md, Workgroup dispatch), so with
// tile = blockIdx.x neighbouring tiles of an image always sit on DIFFERENT L2s and every halo line is fetched from the fabric
// by each of them.  The tile kernels below walk a LOGICAL block order instead in which an XCD owns one contiguous eighth of the
// tiles (whole images at the large levels): halo lines and the half-used cache lines of neighbouring tiles are then shared in
// one L2.  TMG_NO_XCD_MAP=1 (read once per process by the launchers) restores the plain order for A/B measurements.
static inline int tmg_xcd_map_on() {
    static const int on = getenv("TMG_NO_XCD_MAP") ? 0 : 1;
    return on;
}
// logical id of physical block b of a grid of G: XCD x = b % 8 owns the contiguous logical ids [x G/8 + min(x, G%8), ...)
__device__ __forceinline__ int tmg_xcd_block(int b, int G, int on) {
    if (!on || G < 16) return b;
    const int qn = G >> 3, rn = G & 7, x = b & 7;
    return x * qn + min(x, rn) + (b >> 3);
}
// grid-stride tile loop of block b: for (t = first; t < end; t += step).  XCD x walks the tiles [n x / 8, n (x + 1) / 8) with
// the blocks it holds.
struct TmgTileRange { int first, end, step; };
__device__ __forceinline__ TmgTileRange tmg_xcd_tiles(int ntiles, int b, int G, int on) {
    if (!on || G < 16 || ntiles < 64) return TmgTileRange{b, ntiles, G};
    const int x = b & 7;
    const int nb = (G >> 3) + (x < (G & 7) ? 1 : 0);
    const int lo = (int)(((long long)ntiles * x) >> 3), hi = (int)(((long long)ntiles * (x + 1)) >> 3);
    return TmgTileRange{lo + (b >> 3), hi, nb};
}

// Address of 4 consecutive (concatenated) input channels c..c+3 of pixel (b,iy,ix) under the padding rule, for
// float4-addressable segment lists (p.vec4); the zero page when there is nothing to read.
template <typename P>
__device__ __forceinline__ const float* in4_addr(const P& p, int b, int iy, int ix, int c, bool& oob) {
    const int iyc = min(max(iy, 0), p.Hin - 1), ixc = min(max(ix, 0), p.Win - 1);
    oob = !p.pad_rep && (iy != iyc || ix != ixc);
    const size_t pix = ((size_t)b * p.Hin + iyc) * p.Win + ixc;
    int cl = c;
    const float* sp = p.in[0].p;
    int ss = p.in[0].stride, so = p.in[0].off;
    if (cl >= p.in[0].n) {
        cl -= p.in[0].n;
        sp = p.in[1].p; ss = p.in[1].stride; so = p.in[1].off;
        if (cl >= p.in[1].n) {
            cl -= p.in[1].n;
            sp = p.in[2].p; ss = p.in[2].stride; so = p.in[2].off;
        }
    }
    const float* a = sp + pix * ss + so + cl;
    return (oob || c >= p.Cin) ? tmg_zero_page : a;
}

// stage_patch with the U loads of a batch issued back to back WITHOUT control flow between them (a load inside a
// divergent block makes the compiler drain vmcnt before the next load, serialising the batch); the optional affine /
// ReLU run after the whole batch has been issued.  Falls back to stage_patch for segment lists that are not
// float4-addressable.
template <int U = 4, typename P>
__device__ __forceinline__ void stage_patch_bf(const P& p, float* lds, int b, int iy0, int ix0, int PH, int PW, int c0,
                                               int kch, int CS) {
    if (!p.vec4) {
        stage_patch<U>(p, lds, b, iy0, ix0, PH, PW, c0, kch, CS);
        return;
    }
    const int k4 = kch >> 2;
    const int items = PH * PW * k4;
    const int nt = blockDim.x;
    const unsigned mk = k4 > 1 ? 0xFFFFFFFFu / (unsigned)k4 + 1u : 0u;
    const unsigned mp = 0xFFFFFFFFu / (unsigned)PW + 1u;
    for (int base = threadIdx.x; base < items; base += nt * U) {
        float4 v[U];
        const float* a[U];
        int dst[U];
        unsigned oobm = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = min(base + u * nt, items - 1);
            const int pix = k4 > 1 ? (int)__umulhi((unsigned)it, mk) : it;
            const int c4 = it - pix * k4;
            const int py = (int)__umulhi((unsigned)pix, mp);
            const int px = pix - py * PW;
            bool oob;
            a[u] = in4_addr(p, b, iy0 + py, ix0 + px, c0 + 4 * c4, oob);
            oobm |= (oob ? 1u : 0u) << u;
            dst[u] = (base + u * nt < items) ? pix * CS + 4 * c4 : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(a[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p.in_scale && !((oobm >> u) & 1u)) {
                const int it = min(base + u * nt, items - 1);
                const int pix = k4 > 1 ? (int)__umulhi((unsigned)it, mk) : it;
                const int c = c0 + 4 * (it - pix * k4);
                float* f = reinterpret_cast<float*>(&v[u]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < p.Cin) f[e] = f[e] * p.in_scale[c + e] + p.in_shift[c + e];
            }
            if (p.relu_in) {
                v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f);
                v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f);
            }
            if (dst[u] >= 0) *reinterpret_cast<float4*>(lds + dst[u]) = v[u];
        }
    }
}

// Split staging for software pipelining (async-STAGE split): stage_issue() puts up to U loads per thread in flight into
// registers, stage_commit() writes them to LDS after the compute they overlap; items beyond U per thread are staged
// directly at commit time.
template <int U>
struct StageRegs {
    float4 v[U];
    int dst[U];
};

template <int U, typename P>
__device__ __forceinline__ void stage_issue(const P& p, StageRegs<U>& R, int b, int iy0, int ix0, int PH, int PW, int c0, int kch,
                                            int CS) {
    const int k4 = kch >> 2;
    const int items = PH * PW * k4;
    const int nt = blockDim.x;
    const unsigned mk = k4 > 1 ? 0xFFFFFFFFu / (unsigned)k4 + 1u : 0u;
    const unsigned mp = 0xFFFFFFFFu / (unsigned)PW + 1u;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int it = threadIdx.x + u * nt;
        R.dst[u] = -1;
        if (it < items) {
            const int pix = k4 > 1 ? (int)__umulhi((unsigned)it, mk) : it;
            const int c4 = it - pix * k4;
            const int py = (int)__umulhi((unsigned)pix, mp);
            const int px = pix - py * PW;
            R.v[u] = load_in4(p, b, iy0 + py, ix0 + px, c0 + 4 * c4);
            R.dst[u] = pix * CS + 4 * c4;
        }
    }
}

template <int U, typename P>
__device__ __forceinline__ void stage_commit(const P& p, StageRegs<U>& R, float* lds, int b, int iy0, int ix0, int PH, int PW,
                                             int c0, int kch, int CS) {
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (R.dst[u] >= 0) *reinterpret_cast<float4*>(lds + R.dst[u]) = R.v[u];
    const int k4 = kch >> 2;
    const int items = PH * PW * k4;
    const unsigned mk = k4 > 1 ? 0xFFFFFFFFu / (unsigned)k4 + 1u : 0u;
    const unsigned mp = 0xFFFFFFFFu / (unsigned)PW + 1u;
    for (int it = threadIdx.x + U * blockDim.x; it < items; it += blockDim.x) {
        const int pix = k4 > 1 ? (int)__umulhi((unsigned)it, mk) : it;
        const int c4 = it - pix * k4;
        const int py = (int)__umulhi((unsigned)pix, mp);
        const int px = pix - py * PW;
        *reinterpret_cast<float4*>(lds + pix * CS + 4 * c4) = load_in4(p, b, iy0 + py, ix0 + px, c0 + 4 * c4);
    }
}

// Select the output segment holding concatenated channel `nl` BY VALUE (taking a pointer into the
// kernel-argument struct would push the whole struct into scratch memory).
#define TMG_PICK_OSEG(ARR, NL_, PTR_, STRIDE_, OFF_)                                                      \
    float* PTR_ = ARR[0].p;                                                                               \
    int STRIDE_ = ARR[0].stride, OFF_ = ARR[0].off;                                                       \
    if (NL_ >= ARR[0].n) {                                                                                \
        NL_ -= ARR[0].n; PTR_ = ARR[1].p; STRIDE_ = ARR[1].stride; OFF_ = ARR[1].off;                     \
        if (NL_ >= ARR[1].n) { NL_ -= ARR[1].n; PTR_ = ARR[2].p; STRIDE_ = ARR[2].stride; OFF_ = ARR[2].off; } \
    }

// One-time opt-in of a kernel to > 64 KB of dynamic LDS, per DEVICE: the attribute belongs to the (function, device) pair, so a
// per-process flag would leave the second GPU of a multi-device process (the reference's thread-per-GPU replicas,
// utils/parallel.py:222-231) without it.  hipGetDevice is a thread-local read.
static inline void tmg_lds_optin(const void* fn, std::atomic<uint64_t>& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_relaxed) & bit) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done.fetch_or(bit, std::memory_order_relaxed);
}
#define TMG_LDS_OPTIN(FN)                                              \
    do {                                                               \
        static std::atomic<uint64_t> done__{0};                        \
        tmg_lds_optin(reinterpret_cast<const void*>(FN), done__);      \
    } while (0)

// Per-launch HIP-event timing (bench.py's roofline / bandwidth lines); the registry lives in tmg_conv.hip.  work = algorithmic
// flops for matrix-core kernels, algorithmic HBM bytes for the bandwidth-bound classes (ids in g_prof_names there).
extern "C" int tmg_prof_open(int kid, double work, hipStream_t st);
extern "C" void tmg_prof_close(int slot, hipStream_t st);
struct TmgProf {
    int slot; hipStream_t st;
    TmgProf(int kid, double work, hipStream_t s) : slot(tmg_prof_open(kid, work, s)), st(s) {}
    ~TmgProf() { tmg_prof_close(slot, st); }
};
enum { TMG_PROF_CPL = 32, TMG_PROF_C1X2 = 33, TMG_PROF_D2B = 34, TMG_PROF_AFF = 35, TMG_PROF_AFFB = 36, TMG_PROF_LSTMF = 37,
       TMG_PROF_LSTMB = 38, TMG_PROF_GAUSS = 39, TMG_PROF_RESAMPLE = 40, TMG_PROF_MIX16 = 41, TMG_PROF_CPLB = 42, TMG_PROF_WINO = 43, TMG_PROF_WINO_WG = 44 };

// Compute units of the current device (queried once per process: one process per GPU).  The persistent kernels size their grids as
// multiples of it - one 512-thread block per CU, or the measured 2x / 3x / 4x of the narrow-level tile kernels - instead of the
// literal 256 of an MI355X (a silent mis-tuning on any other part).
static inline int tmg_num_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

#define TMG_CHECK_LAUNCH()                          \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)
