// tmg_mix16.hip -- the invertible 1x1 channel mix (ActNorm folded in) with fp16 operands and fp32 accumulation on the
// gfx950 matrix cores (v_mfma_f32_16x16x16_f16).  The reduced-precision variant BASELINE.json's configs[4] names
// ("fp16 MFMA 1x1 convs"); replaces F.conv2d(x, W[C,C,1,1]) at glowConv.py:193-194 / :219-220 and its input gradient.
// Opt-in only (tmg_ops.set_mix_precision / TMG_MIX_F16): the default path is the fp32 MFMA kernel of tmg_conv.hip.
//
//   y[p, co] = sum_ci  fp16(W[co, ci]) * fp16(x[p, ci])  (fp32 accumulate)  + bias[co]
//
// Mapping.  D = A x B with A = W tile (16 output channels x 16 input channels, fp16 in LDS, layout [ci/4][co][4]),
// B = x^T (16 input channels x 16 pixels): a lane's B fragment is 4 consecutive channels of one pixel = ONE float4 global
// load (converted to fp16 in registers, activations stay fp32 in HBM), and its 4 accumulators are 4 consecutive output
// channels of one pixel = ONE float4 store.  No LDS round trip for activations.  The op is HBM-bound (8C bytes per pixel
// against 2C^2 flop): a wave keeps NP*NT float4 loads in flight per iteration.
#include "tmg_common.h"

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

struct Mix16P {
    const float* x; int xs;
    const float* W; const float* bias;
    float* y; int ys;
    long npix; int C; int transposed;
    // mix32_kernel with the affine coupling fused in (AFF = 1 / 2, the 64- and 128-channel levels' generative direction):
    int ppi;                       // pixels per image (multiple of a wave's pixel group)
    const float* hh; int hs;       // AFF 1: zero-conv output [npix][C], (shift, r) interleaved
    float* r_out; float* y2_out;   // AFF 1: softsign argument and transformed half, [npix][C/2] each (saved for backward)
    float* logdet;                 // AFF 1: [images] accumulators
    const float* r_in;             // AFF 2: the saved r
    const float* t2; int t2s;      // AFF 2: the coupling's input half x2 (pointer at its first channel), pixel stride
    const float* g;                // AFF 2: upstream gradient on the per-image log-det, or null
    const float* kappa;            // AFF 2: log-scale of the zero conv (dhh is written pre-multiplied by exp(clamp(kappa)))
    float* dtin2; int dts;         // AFF 2: gradient w.r.t. x2 (pointer at its first channel), pixel stride
    float* dhh; int dhs;           // AFF 2: exp(kappa) x gradient w.r.t. the zero-conv output [npix][C], (da, dr) interleaved
};

template <int NT, int NP>
__global__ __launch_bounds__(256) void mix16_kernel(Mix16P p) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    h16x4* Wl = reinterpret_cast<h16x4*>(smem_raw);  // [NT*4][CP]: (ci quad, co) -> 4 halfs
    constexpr int CP = NT * 16;
    const int C = p.C;
    // W -> fp16 tiles in LDS.  Eight items per thread are loaded before any is converted (round 6: one item at a time was a chain of
    // 64 exposed L2 latencies per thread at C = 256 - 30 us of a 60-us launch on the 16 x 16 maps of cfg5's deepest level); the plain
    // orientation reads its four input channels as ONE float4 (C % 4 == 0: the entry point checks it)
    constexpr int WU = 8;
    for (int i0 = threadIdx.x; i0 < NT * 4 * CP; i0 += 256 * WU) {
        float4 wv[WU];
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int i = i0 + u * 256;
            const int kq = i / CP, co = i - kq * CP;
            const int ci = 4 * kq;
            wv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < NT * 4 * CP && co < C && ci < C) {
                if (!p.transposed) {
                    wv[u] = *reinterpret_cast<const float4*>(p.W + (size_t)co * C + ci);
                } else {
                    wv[u].x = p.W[(size_t)ci * C + co];
                    wv[u].y = p.W[(size_t)(ci + 1) * C + co];
                    wv[u].z = p.W[(size_t)(ci + 2) * C + co];
                    wv[u].w = p.W[(size_t)(ci + 3) * C + co];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int i = i0 + u * 256;
            if (i < NT * 4 * CP) {
                h16x4 v;
                v[0] = (_Float16)wv[u].x; v[1] = (_Float16)wv[u].y; v[2] = (_Float16)wv[u].z; v[3] = (_Float16)wv[u].w;
                Wl[i] = v;
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, lq = lane >> 4;
    constexpr bool AREG = NT <= 4;           // the whole operand fits in registers for C <= 64
    h16x4 Areg[AREG ? NT * NT : 1];
    if (AREG) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int ks = 0; ks < NT; ++ks) Areg[mt * NT + ks] = Wl[(ks * 4 + lq) * CP + mt * 16 + l16];
    }
    const long group = 16L * NP;
    if constexpr (NT > 8) {
        // Wide mixes (C > 128: the 256-channel level of BASELINE configs[4]).  Round 5's form kept NT bias quads, NT fp32 B quads, their
        // fp16 conversions and - hoisted by the scheduler - most of a tile row's A fragments live at once: 564 B of scratch per lane at
        // NT = 16.  Here the B quads are converted as they arrive (the fp32 copies die at once), the bias quad is read per output tile
        // (L1-resident) and a scheduling fence per output tile keeps the A-fragment reads of later tiles from being hoisted.
        for (long g0 = ((long)blockIdx.x * 4 + wave) * group; g0 < p.npix; g0 += (long)gridDim.x * 4 * group) {
#pragma unroll
            for (int np = 0; np < NP; ++np) {
                const long px = g0 + np * 16 + l16;
                h16x4 Bf[NT];
                {
                    float4 xv[NT];
#pragma unroll
                    for (int ks = 0; ks < NT; ++ks) {
                        const int c = ks * 16 + 4 * lq;
                        const float* a = (px < p.npix && c < C) ? p.x + (size_t)px * p.xs + c : tmg_zero_page;
                        xv[ks] = *reinterpret_cast<const float4*>(a);
                    }
#pragma unroll
                    for (int ks = 0; ks < NT; ++ks) {
                        Bf[ks][0] = (_Float16)xv[ks].x; Bf[ks][1] = (_Float16)xv[ks].y;
                        Bf[ks][2] = (_Float16)xv[ks].z; Bf[ks][3] = (_Float16)xv[ks].w;
                    }
                }
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) {
                    __builtin_amdgcn_sched_barrier(0);
                    const int c = mt * 16 + 4 * lq;
                    const float4 bq = (p.bias && c < C) ? *reinterpret_cast<const float4*>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                    f32x4 acc = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                    for (int ks = 0; ks < NT; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(Wl[(ks * 4 + lq) * CP + mt * 16 + l16], Bf[ks], acc, 0, 0, 0);
                    if (px < p.npix && c < C)
                        *reinterpret_cast<float4*>(p.y + (size_t)px * p.ys + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                }
            }
        }
        return;
    }
    float4 bv[NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        const int c = mt * 16 + 4 * lq;
        bv[mt] = (p.bias && c < C) ? *reinterpret_cast<const float4*>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long g0 = ((long)blockIdx.x * 4 + wave) * group; g0 < p.npix; g0 += (long)gridDim.x * 4 * group) {
        float4 xv[NP][NT];
#pragma unroll
        for (int np = 0; np < NP; ++np)
#pragma unroll
            for (int ks = 0; ks < NT; ++ks) {
                const long px = g0 + np * 16 + l16;
                const int c = ks * 16 + 4 * lq;
                const float* a = (px < p.npix && c < C) ? p.x + (size_t)px * p.xs + c : tmg_zero_page;
                xv[np][ks] = *reinterpret_cast<const float4*>(a);
            }
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            h16x4 Bf[NT];
#pragma unroll
            for (int ks = 0; ks < NT; ++ks) {
                Bf[ks][0] = (_Float16)xv[np][ks].x; Bf[ks][1] = (_Float16)xv[np][ks].y;
                Bf[ks][2] = (_Float16)xv[np][ks].z; Bf[ks][3] = (_Float16)xv[np][ks].w;
            }
            const long px = g0 + np * 16 + l16;
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                f32x4 acc = {bv[mt].x, bv[mt].y, bv[mt].z, bv[mt].w};
#pragma unroll
                for (int ks = 0; ks < NT; ++ks) {
                    const h16x4 a = AREG ? Areg[mt * NT + ks] : Wl[(ks * 4 + lq) * CP + mt * 16 + l16];
                    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, Bf[ks], acc, 0, 0, 0);
                }
                const int c = mt * 16 + 4 * lq;
                if (px < p.npix && c < C)
                    *reinterpret_cast<float4*>(p.y + (size_t)px * p.ys + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            }
        }
    }
}

template <int NT, int NP>
static int launch_mix16(const Mix16P& p, hipStream_t st) {
    const size_t lds = (size_t)NT * 4 * NT * 16 * sizeof(h16x4);
    if (lds > 64 * 1024) TMG_LDS_OPTIN((&mix16_kernel<NT, NP>));
    const long groups = (p.npix + 64L * NP - 1) / (64L * NP);
    const int grid = (int)(groups < 2048 ? (groups < 1 ? 1 : groups) : 2048);
    TmgProf prof(TMG_PROF_MIX16, 8.0 * (double)p.npix * p.C, st);   // x read, y written
    hipLaunchKernelGGL((mix16_kernel<NT, NP>), dim3(grid), dim3(256), lds, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

// y = fp16(W) . fp16(x) + bias per pixel, fp32 accumulation.  x_d / y_d = {pixel stride, channel offset};
// dims = {npix, C, transposed}: transposed != 0 uses W^T (the input gradient of the same mix).  C % 4 == 0, C <= 256.
extern "C" int tmg_mix_f16(const void* x, const int64_t* x_d, const void* W, const void* bias, void* y, const int64_t* y_d,
                           const int64_t* dims, hipStream_t st) {
    Mix16P p;
    p.x = static_cast<const float*>(x) + x_d[1]; p.xs = (int)x_d[0];
    p.y = static_cast<float*>(y) + y_d[1]; p.ys = (int)y_d[0];
    p.W = static_cast<const float*>(W); p.bias = static_cast<const float*>(bias);
    p.npix = (long)dims[0]; p.C = (int)dims[1]; p.transposed = (int)dims[2];
    if (p.C < 4 || p.C % 4 || p.C > 256 || p.xs % 4 || p.ys % 4 || x_d[1] % 4 || y_d[1] % 4) return -1;
    if (p.npix <= 0) return 0;
    switch ((p.C + 15) / 16) {
        case 1: return launch_mix16<1, 8>(p, st);
        case 2: return launch_mix16<2, 8>(p, st);
        case 3: return launch_mix16<3, 4>(p, st);
        case 4: return launch_mix16<4, 4>(p, st);
        case 5: case 6: return launch_mix16<6, 2>(p, st);
        case 7: case 8: return launch_mix16<8, 2>(p, st);
        case 9: case 10: case 11: case 12: return launch_mix16<12, 1>(p, st);
        default: return launch_mix16<16, 1>(p, st);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same mix in full fp32 (v_mfma_f32_16x16x4_f32), for the stand-alone mixes of the wide levels and the ConvLSTM blocks: tensors
// of 8-64 MB with K = C <= 256, i.e. a bandwidth / latency kernel.  The general conv kernel spent ~20 us per launch on them (512-
// thread persistent blocks, LDS-staged patch, packed-operand launch beforehand); here a wave streams 16-pixel groups straight
// from global memory (B fragment = one float4 per lane and 16 channels) against the weight held in LDS ([ci/4][co][4] floats, read
// as float4 = four k-steps), no packing launch.
// ---------------------------------------------------------------------------------------------------------------------------------
// AFF (the per-op chain of the wide levels, generative direction: coupling -> mix, flowAffine.py:102-109 + glowConv.py:207-222):
//   0: y = W x + b.
//   1: the affine coupling evaluated on the way IN: the second channel half of the mix input is y2 = x2 exp(-2 softsign(r)) - shift
//      with (shift, r) read from the zero-conv output; r and y2 are stored for backward, the log-det is summed per wave and image.
//   2: (transposed mix = its input gradient) the coupling's backward on the way OUT: the second half of W^T dy is turned into the
//      gradient w.r.t. x2 and exp(kappa) x the gradient w.r.t. the zero-conv output (tmg_affine_bwd_scaled's arithmetic); the first
//      half is stored as is.  One launch and one [npix][C] round trip less per layer and direction.
template <int NT, int NP, int AFF>
__global__ __launch_bounds__(256) void mix32_kernel(Mix16P p) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float4* Wl = reinterpret_cast<float4*>(smem_raw);   // [NT*4][CP]: (ci quad, co) -> 4 floats
    constexpr int CP = NT * 16;
    const int C = p.C;
    // Weight tile -> LDS.  On the 64- / 128-channel levels a block works on 64 pixels only, so this staging IS the kernel's time: the
    // loads of UB items are issued together (a one-item loop is a chain of exposed L2 latencies: 64 iterations at 128 channels), the
    // W[co][ci] orientation as one float4 per item, out-of-range items from the zero page (no divergent loads).
    constexpr int TOT = NT * 4 * CP, UB = 8;
    for (int i0 = threadIdx.x; i0 < TOT; i0 += 256 * UB) {
        float4 v[UB];
        if (!p.transposed && (C & 3) == 0) {
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = i0 + u * 256;
                const int kq = i / CP, co = i - kq * CP;
                const bool ok = i < TOT && co < C && 4 * kq < C;
                v[u] = *reinterpret_cast<const float4*>(ok ? p.W + (size_t)co * C + 4 * kq : tmg_zero_page);
            }
        } else {
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = i0 + u * 256;
                const int kq = i / CP, co = i - kq * CP;
                float f[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ci = 4 * kq + e;
                    const bool ok = i < TOT && co < C && ci < C;
                    f[e] = *(ok ? (p.transposed ? p.W + (size_t)ci * C + co : p.W + (size_t)co * C + ci) : tmg_zero_page);
                }
                v[u] = make_float4(f[0], f[1], f[2], f[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u)
            if (i0 + u * 256 < TOT) Wl[i0 + u * 256] = v[u];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, lq = lane >> 4;
    float4 bv[NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        const int c = mt * 16 + 4 * lq;
        bv[mt] = (p.bias && c < C) ? *reinterpret_cast<const float4*>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const long group = 16L * NP;
    constexpr int HT = NT / 2;               // channel tiles per half (AFF != 0: C == NT * 16)
    const float hsc = AFF == 2 ? out_scale_of(p.kappa) : 1.f;
    for (long g0 = ((long)blockIdx.x * 4 + wave) * group; g0 < p.npix; g0 += (long)gridDim.x * 4 * group) {
        float4 xv[NP][NT];
        float ldsum = 0.f;
#pragma unroll
        for (int np = 0; np < NP; ++np)
#pragma unroll
            for (int ks = 0; ks < NT; ++ks) {
                const long px = g0 + np * 16 + l16;
                const int c = ks * 16 + 4 * lq;
                const float* a = (px < p.npix && c < C) ? p.x + (size_t)px * p.xs + c : tmg_zero_page;
                xv[np][ks] = *reinterpret_cast<const float4*>(a);
                if (AFF == 1 && ks >= HT) {
                    const int c2 = c - HT * 16;          // channel inside the second half
                    const bool ok = px < p.npix;
                    const float* hp = ok ? p.hh + (size_t)px * p.hs + 2 * c2 : tmg_zero_page;
                    const float4 ha = *reinterpret_cast<const float4*>(hp), hb = *reinterpret_cast<const float4*>(hp + (ok ? 4 : 0));
                    const float sh[4] = {ha.x, ha.z, hb.x, hb.z}, rr[4] = {ha.y, ha.w, hb.y, hb.w};
                    const float xx[4] = {xv[np][ks].x, xv[np][ks].y, xv[np][ks].z, xv[np][ks].w};
                    float oo[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sg = 2.f * rr[e] / (1.f + fabsf(rr[e]));
                        oo[e] = xx[e] * expf(-sg) - sh[e];
                        ldsum += ok ? sg : 0.f;
                    }
                    xv[np][ks] = make_float4(oo[0], oo[1], oo[2], oo[3]);
                    if (ok) {
                        *reinterpret_cast<float4*>(p.r_out + (size_t)px * (HT * 16) + c2) = make_float4(rr[0], rr[1], rr[2], rr[3]);
                        *reinterpret_cast<float4*>(p.y2_out + (size_t)px * (HT * 16) + c2) = xv[np][ks];
                    }
                }
            }
        if (AFF == 1) {
            // one atomic per wave and group: the group's pixels lie in one image (ppi is a multiple of the group, launcher)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ldsum += __shfl_xor(ldsum, o);
            if (lane == 0) atomicAdd(p.logdet + g0 / p.ppi, ldsum);
        }
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            const long px = g0 + np * 16 + l16;
            f32x4 acc[NT];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) acc[mt] = (f32x4){bv[mt].x, bv[mt].y, bv[mt].z, bv[mt].w};
#pragma unroll
            for (int ks = 0; ks < NT; ++ks) {
                // step e of channel block ks contracts channels 16 ks + 4 lq + e on both operands: the weight fragment is component e
                // of the lane's LDS quad (co = l16, quad lq), the pixel fragment component e of its global quad (pixel l16, quad lq);
                // the output tiles are the inner loop so that consecutive MFMAs hit different accumulators
                float4 a[NT];
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) a[mt] = Wl[(ks * 4 + lq) * CP + mt * 16 + l16];
                const float4 b = xv[np][ks];
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b.x, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b.y, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, b.z, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, b.w, acc[mt], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                const int c = mt * 16 + 4 * lq;
                if (AFF == 2 && mt >= HT) {
                    if (px < p.npix) {
                        const int c2 = c - HT * 16;
                        const float4 rq = *reinterpret_cast<const float4*>(p.r_in + (size_t)px * (HT * 16) + c2);
                        const float4 yq = *reinterpret_cast<const float4*>(p.t2 + (size_t)px * p.t2s + c2);
                        const float gb = p.g ? p.g[px / p.ppi] : 0.f;
                        const float rr[4] = {rq.x, rq.y, rq.z, rq.w}, yy[4] = {yq.x, yq.y, yq.z, yq.w};
                        float gi[4], da[4], dr[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float den = 1.f + fabsf(rr[e]);
                            const float sg = 2.f * rr[e] / den;
                            const float inv = expf(-sg);
                            const float go_ = acc[mt][e];
                            gi[e] = go_ * inv;
                            da[e] = -go_ * hsc;
                            dr[e] = hsc * (-2.f * go_ * (yy[e] * inv) + 2.f * gb) / (den * den);
                        }
                        *reinterpret_cast<float4*>(p.dtin2 + (size_t)px * p.dts + c2) = make_float4(gi[0], gi[1], gi[2], gi[3]);
                        float* dp = p.dhh + (size_t)px * p.dhs + 2 * c2;
                        *reinterpret_cast<float4*>(dp) = make_float4(da[0], dr[0], da[1], dr[1]);
                        *reinterpret_cast<float4*>(dp + 4) = make_float4(da[2], dr[2], da[3], dr[3]);
                    }
                } else if (px < p.npix && c < C) {
                    *reinterpret_cast<float4*>(p.y + (size_t)px * p.ys + c) = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
                }
            }
        }
    }
}

template <int NT, int NP, int AFF = 0>
static int launch_mix32(const Mix16P& p, hipStream_t st) {
    const size_t lds = (size_t)NT * 4 * NT * 16 * sizeof(float4);
    if (lds > 64 * 1024) TMG_LDS_OPTIN((&mix32_kernel<NT, NP, AFF>));
    const long groups = (p.npix + 64L * NP - 1) / (64L * NP);
    const int grid = (int)(groups < 2048 ? (groups < 1 ? 1 : groups) : 2048);
    TmgProf prof(31, 2.0 * (double)p.npix * p.C * p.C, st);   // the "conv 1x1 (invertible channel mix, fp32 MFMA)" class: flops
    hipLaunchKernelGGL((mix32_kernel<NT, NP, AFF>), dim3(grid), dim3(256), lds, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

// The mix of a coupling layer's generative direction with the affine coupling fused in (see mix32_kernel, AFF = 1): C = 64 or 128.
// x: the layer input [npix][C] (x1 | x2), hh: zero-conv output; y = W [x1 ; y2] + bias, r / y2: [npix][C/2] dense, logdet[image] +=.
// dims = {npix, C, pixels per image}; x_d / hh_d / y_d = {pixel stride, channel offset}.  -100 outside the envelope.
extern "C" int tmg_mix_f32_affine_fwd(const void* x, const int64_t* x_d, const void* hh, const int64_t* hh_d, const void* W, const void* bias,
                                      void* y, const int64_t* y_d, void* r, void* y2, void* logdet, const int64_t* dims, hipStream_t st) {
    Mix16P p = {};
    p.x = static_cast<const float*>(x) + x_d[1]; p.xs = (int)x_d[0];
    p.hh = static_cast<const float*>(hh) + hh_d[1]; p.hs = (int)hh_d[0];
    p.y = static_cast<float*>(y) + y_d[1]; p.ys = (int)y_d[0];
    p.W = static_cast<const float*>(W); p.bias = static_cast<const float*>(bias);
    p.r_out = static_cast<float*>(r); p.y2_out = static_cast<float*>(y2); p.logdet = static_cast<float*>(logdet);
    p.npix = (long)dims[0]; p.C = (int)dims[1]; p.ppi = (int)dims[2]; p.transposed = 0;
    if ((p.C != 64 && p.C != 128) || p.xs % 4 || p.ys % 4 || p.hs % 4 || x_d[1] % 4 || y_d[1] % 4 || hh_d[1] % 4 || p.ppi <= 0 ||
        p.npix % p.ppi || p.ppi % (p.C == 64 ? 32 : 16))
        return -100;
    if (((((uintptr_t)p.x) | ((uintptr_t)p.hh) | ((uintptr_t)p.y) | ((uintptr_t)p.r_out) | ((uintptr_t)p.y2_out)) & 15)) return -100;
    if (p.npix <= 0) return 0;
    return p.C == 64 ? launch_mix32<4, 2, 1>(p, st) : launch_mix32<8, 1, 1>(p, st);
}

// Its input gradient with the coupling's backward fused in (AFF = 2): dy [npix][C] -> dto1 = (W^T dy)[:C/2] ([npix][C/2] dense),
// dtin2 = gradient w.r.t. x2 (pointer at the first x2 channel, stride dtin2_d[0]), dhh [npix][C] = exp(clamp(kappa)) x the gradient
// w.r.t. the zero-conv output.  r: saved softsign argument, t2: the coupling's input half x2, g: per-image log-det gradient or null.
extern "C" int tmg_mix_f32_affine_bwd(const void* dy, const int64_t* dy_d, const void* W, const void* r, const void* t2, const int64_t* t2_d,
                                      const void* g, const void* kappa, void* dto1, void* dtin2, const int64_t* dtin2_d, void* dhh,
                                      const int64_t* dhh_d, const int64_t* dims, hipStream_t st) {
    Mix16P p = {};
    p.x = static_cast<const float*>(dy) + dy_d[1]; p.xs = (int)dy_d[0];
    p.W = static_cast<const float*>(W); p.bias = nullptr;
    p.npix = (long)dims[0]; p.C = (int)dims[1]; p.ppi = (int)dims[2]; p.transposed = 1;
    p.y = static_cast<float*>(dto1); p.ys = p.C / 2;
    p.r_in = static_cast<const float*>(r);
    p.t2 = static_cast<const float*>(t2) + t2_d[1]; p.t2s = (int)t2_d[0];
    p.g = static_cast<const float*>(g); p.kappa = static_cast<const float*>(kappa);
    p.dtin2 = static_cast<float*>(dtin2) + dtin2_d[1]; p.dts = (int)dtin2_d[0];
    p.dhh = static_cast<float*>(dhh) + dhh_d[1]; p.dhs = (int)dhh_d[0];
    if ((p.C != 64 && p.C != 128) || p.xs % 4 || p.t2s % 4 || p.dts % 4 || p.dhs % 4 || dy_d[1] % 4 || t2_d[1] % 4 || dtin2_d[1] % 4 ||
        dhh_d[1] % 4 || p.ppi <= 0 || p.npix % p.ppi)
        return -100;
    if (((((uintptr_t)p.x) | ((uintptr_t)p.y) | ((uintptr_t)p.r_in) | ((uintptr_t)p.t2) | ((uintptr_t)p.dtin2) | ((uintptr_t)p.dhh)) & 15)) return -100;
    if (p.npix <= 0) return 0;
    return p.C == 64 ? launch_mix32<4, 2, 2>(p, st) : launch_mix32<8, 1, 2>(p, st);
}

// y = W x + bias per pixel in fp32 on the matrix cores; arguments as tmg_mix_f16.  C % 4 == 0, C <= 128 (the weight tile lives in LDS).
extern "C" int tmg_mix_f32(const void* x, const int64_t* x_d, const void* W, const void* bias, void* y, const int64_t* y_d,
                           const int64_t* dims, hipStream_t st) {
    Mix16P p;
    p.x = static_cast<const float*>(x) + x_d[1]; p.xs = (int)x_d[0];
    p.y = static_cast<float*>(y) + y_d[1]; p.ys = (int)y_d[0];
    p.W = static_cast<const float*>(W); p.bias = static_cast<const float*>(bias);
    p.npix = (long)dims[0]; p.C = (int)dims[1]; p.transposed = (int)dims[2];
    if (p.C < 4 || p.C % 4 || p.C > 128 || p.xs % 4 || p.ys % 4 || x_d[1] % 4 || y_d[1] % 4) return -1;
    if (p.npix <= 0) return 0;
    switch ((p.C + 15) / 16) {
        case 1: return launch_mix32<1, 8>(p, st);
        case 2: return launch_mix32<2, 4>(p, st);
        case 3: return launch_mix32<3, 2>(p, st);
        case 4: return launch_mix32<4, 2>(p, st);
        case 5: case 6: return launch_mix32<6, 1>(p, st);
        default: return launch_mix32<8, 1>(p, st);
    }
}
