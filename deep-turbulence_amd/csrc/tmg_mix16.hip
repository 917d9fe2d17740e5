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
};

template <int NT, int NP>
__global__ __launch_bounds__(256) void mix16_kernel(Mix16P p) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    h16x4* Wl = reinterpret_cast<h16x4*>(smem_raw);  // [NT*4][CP]: (ci quad, co) -> 4 halfs
    constexpr int CP = NT * 16;
    const int C = p.C;
    for (int i = threadIdx.x; i < NT * 4 * CP; i += 256) {
        const int kq = i / CP, co = i - kq * CP;
        h16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ci = 4 * kq + e;
            float f = 0.f;
            if (co < C && ci < C) f = p.transposed ? p.W[(size_t)ci * C + co] : p.W[(size_t)co * C + ci];
            v[e] = (_Float16)f;
        }
        Wl[i] = v;
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
    float4 bv[NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        const int c = mt * 16 + 4 * lq;
        bv[mt] = (p.bias && c < C) ? *reinterpret_cast<const float4*>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const long group = 16L * NP;
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
template <int NT, int NP>
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
                if (px < p.npix && c < C)
                    *reinterpret_cast<float4*>(p.y + (size_t)px * p.ys + c) = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
            }
        }
    }
}

template <int NT, int NP>
static int launch_mix32(const Mix16P& p, hipStream_t st) {
    const size_t lds = (size_t)NT * 4 * NT * 16 * sizeof(float4);
    if (lds > 64 * 1024) TMG_LDS_OPTIN((&mix32_kernel<NT, NP>));
    const long groups = (p.npix + 64L * NP - 1) / (64L * NP);
    const int grid = (int)(groups < 2048 ? (groups < 1 ? 1 : groups) : 2048);
    TmgProf prof(31, 2.0 * (double)p.npix * p.C * p.C, st);   // the "conv 1x1 (invertible channel mix, fp32 MFMA)" class: flops
    hipLaunchKernelGGL((mix32_kernel<NT, NP>), dim3(grid), dim3(256), lds, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
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
