// Round 6: the small launches that used to be torch-native glue around the flow kernels (rocprofv3 of the step, round 5:
// 149 at::native launches, 1.2 ms of 43) as kernels of this library -
//   gauss_sample_kernel   z2 = mean + e^{lsd} eps written INTO the second half of the level's [.., C] activation, the pass-through
//                         half copied beside it (Split.reverse's torch.cat, flowUtils.py:331-334), eps either given (reconstruct) or
//                         drawn here (Philox4x32-10 + Box-Muller, keyed by a per-call nonce that torch's own generator produced:
//                         graph-safe and reproducible under torch.manual_seed without a randn launch and an eps round trip per level)
//   reverse_loss_*        the benchmark loss mean(y^2) + mean(logdet) / (noc H W) of SURVEY 8-D and its gradient: two launches
//                         instead of the ~15 of pow / mean / div / add and their autograd nodes
//   sum_terms_kernel      log-det bookkeeping: the sum of up to 8 per-sample vectors / scalars (tmGlow.py:438-440,
//                         flowLSTMBlock.py:314-318 add them one `+` at a time: ~25 one-block launches per step)
//   level_pack_kernel     the parameter-side operands of a level node (stacked zero-conv weights, the conditioning columns of all
//                         layers, biases, scales) gathered through a device pointer table: one launch instead of ~10 stack / cat /
//                         slice-copy launches per level and direction
#include "tmg_common.h"
#include "tmglow_hip.h"

static inline int glue_grid(size_t n, int cap = 4096) {
    size_t g = (n + 255) / 256;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

#define LOG2PI_G 1.8378770664093453f

// ---------------------------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter = (quad index lo, quad index hi, site, 0), key = the call's 64-bit nonce
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        const unsigned hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0;
        k.y += W1;
    }
    return c;
}

__device__ __forceinline__ float u01(unsigned x) {      // (0, 1]: never 0, so the logarithm below is finite
    return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}

__device__ __forceinline__ void normal4(uint4 r, float* n) {
    const float r0 = sqrtf(-2.f * logf(u01(r.x))), r1 = sqrtf(-2.f * logf(u01(r.z)));
    float s0, c0, s1, c1;
    sincospif(2.f * u01(r.y), &s0, &c0);
    sincospif(2.f * u01(r.w), &s1, &c1);
    n[0] = r0 * c0;
    n[1] = r0 * s0;
    n[2] = r1 * c1;
    n[3] = r1 * s1;
}

// hz: [npix][2 Ch] = (mean | log-std), clipped as in gauss_fwd_kernel (tmg_pointwise.hip).  One block row per image.
// eps_in != null: the latents are given; else they are drawn (nonce: two int64 on the device, site: which draw of the call) and,
// when eps_out != null, stored for the backward pass.  z2 -> out (stride os, offset oo); pass != null: out[.., po + j] = pass[.., j].
template <bool VEC>
__global__ __launch_bounds__(256) void gauss_sample_kernel(const float* __restrict__ hz, int hs, const float* __restrict__ eps_in, int es,
                                                           const float* __restrict__ pass, int ps, float* __restrict__ out, int os, int oo,
                                                           int po, float* __restrict__ eps_out, float* __restrict__ logp,
                                                           const unsigned long long* __restrict__ nonce, unsigned site, int pix_per_img,
                                                           int Ch, int clip_mean, float mlo, float mhi, float slo, float shi) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const size_t base = (size_t)b * pix_per_img;
    const unsigned total = (unsigned)pix_per_img * (unsigned)Ch;
    const unsigned nquad = (total + 3u) >> 2;
    uint2 key = make_uint2(0u, 0u);
    if (!eps_in) {
        const unsigned long long k0 = nonce[0], k1 = nonce[1];
        key = make_uint2((unsigned)k0 ^ (unsigned)(k1 >> 32), (unsigned)(k0 >> 32) ^ (unsigned)k1);
    }
    float lp = 0.f;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < nquad; q += gridDim.x * 256u) {
        float nrm[4];
        if (!eps_in) {
            const unsigned long long gq = (unsigned long long)b * nquad + q;
            normal4(philox4x32_10(make_uint4((unsigned)gq, (unsigned)(gq >> 32), site, 0x7467u), key), nrm);
        }
        if constexpr (VEC) {
            // Ch % 4 == 0 and every stride / offset a multiple of 4 floats: the quad is four consecutive channels of ONE pixel - one
            // float4 per operand (the scalar form issued four 4-byte requests per operand and lane: 131 us at the first level of the
            // metric configuration against 31 us for the round-5 kernel that neither drew nor copied)
            const unsigned i = 4u * q;
            const unsigned pl = i / (unsigned)Ch;
            const size_t pix = base + pl;
            const int j = (int)(i - pl * (unsigned)Ch);
            float4 mean = *reinterpret_cast<const float4*>(hz + pix * hs + j);
            float4 lsd = *reinterpret_cast<const float4*>(hz + pix * hs + Ch + j);
            if (clip_mean) {
                mean.x = fminf(fmaxf(mean.x, mlo), mhi); mean.y = fminf(fmaxf(mean.y, mlo), mhi);
                mean.z = fminf(fmaxf(mean.z, mlo), mhi); mean.w = fminf(fmaxf(mean.w, mlo), mhi);
            }
            lsd.x = fminf(fmaxf(lsd.x, slo), shi); lsd.y = fminf(fmaxf(lsd.y, slo), shi);
            lsd.z = fminf(fmaxf(lsd.z, slo), shi); lsd.w = fminf(fmaxf(lsd.w, slo), shi);
            float4 v = make_float4(nrm[0], nrm[1], nrm[2], nrm[3]);
            if (eps_in) v = *reinterpret_cast<const float4*>(eps_in + pix * es + j);
            lp += -0.5f * (4.f * LOG2PI_G + 2.f * (lsd.x + lsd.y + lsd.z + lsd.w) + v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
            *reinterpret_cast<float4*>(out + pix * os + oo + j) =
                make_float4(mean.x + expf(lsd.x) * v.x, mean.y + expf(lsd.y) * v.y, mean.z + expf(lsd.z) * v.z, mean.w + expf(lsd.w) * v.w);
            if (eps_out) *reinterpret_cast<float4*>(eps_out + pix * Ch + j) = v;
            if (pass) *reinterpret_cast<float4*>(out + pix * os + po + j) = *reinterpret_cast<const float4*>(pass + pix * ps + j);
            continue;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned i = 4u * q + e;
            if (i >= total) break;
            const unsigned pl = i / (unsigned)Ch;
            const size_t pix = base + pl;
            const int j = (int)(i - pl * (unsigned)Ch);
            float mean = hz[pix * hs + j];
            float lsd = hz[pix * hs + Ch + j];
            if (clip_mean) mean = fminf(fmaxf(mean, mlo), mhi);
            lsd = fminf(fmaxf(lsd, slo), shi);
            const float v = eps_in ? eps_in[pix * es + j] : nrm[e];
            lp += -0.5f * (LOG2PI_G + 2.f * lsd + v * v);
            out[pix * os + oo + j] = mean + expf(lsd) * v;
            if (eps_out) eps_out[pix * Ch + j] = v;
            if (pass) out[pix * os + po + j] = pass[pix * ps + j];
        }
    }
    const float tot = block_sum_256(lp, red);
    if (threadIdx.x == 0) atomicAdd(logp + b, tot);
}

// dims = {B, pixels per image, Ch, clip_mean, site}; d-arrays = {pixel stride, channel offset}; fl = {mean_lo, mean_hi, lsd_lo, lsd_hi}
extern "C" int tmg_gauss_sample(const void* hz, const int64_t* hz_d, const void* eps_in, const int64_t* ei_d, const void* pass,
                                const int64_t* p_d, void* out, const int64_t* o_d, int64_t pass_off, void* eps_out, void* logp,
                                const void* nonce, const int64_t* dims, const float* fl, hipStream_t st) {
    const int B = (int)dims[0], ppi = (int)dims[1], Ch = (int)dims[2];
    const size_t per = (size_t)ppi * Ch;
    if (per >= (1ull << 31)) return -2;
    if (!eps_in && !nonce) return -3;
    int gx = (int)((per / 4 + 255) / 256);      // as tmg_gauss_fwd: few blocks per image (one atomic each), enough to fill the chip
    int cap = (2048 + B - 1) / B;
    if (cap < 4) cap = 4;
    if (cap > 256) cap = 256;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    const float* hzp = (const float*)hz + hz_d[1];
    const float* ep = eps_in ? (const float*)eps_in + ei_d[1] : nullptr;
    const float* pp = pass ? (const float*)pass + p_d[1] : nullptr;
    bool vec = (Ch & 3) == 0 && (hz_d[0] & 3) == 0 && (o_d[0] & 3) == 0 && (o_d[1] & 3) == 0 && (pass_off & 3) == 0 &&
               ((((uintptr_t)hzp) | ((uintptr_t)out) | ((uintptr_t)eps_out)) & 15) == 0;
    if (ep) vec = vec && (ei_d[0] & 3) == 0 && (((uintptr_t)ep) & 15) == 0;
    if (pp) vec = vec && (p_d[0] & 3) == 0 && (((uintptr_t)pp) & 15) == 0;
#define TMG_GS_ARGS dim3(gx, B), dim3(256), 0, st, hzp, (int)hz_d[0], ep, eps_in ? (int)ei_d[0] : 0, pp, pass ? (int)p_d[0] : 0, (float*)out, \
                    (int)o_d[0], (int)o_d[1], (int)pass_off, (float*)eps_out, (float*)logp, (const unsigned long long*)nonce, (unsigned)dims[4], \
                    ppi, Ch, (int)dims[3], fl[0], fl[1], fl[2], fl[3]
    if (vec) hipLaunchKernelGGL(gauss_sample_kernel<true>, TMG_GS_ARGS);
    else hipLaunchKernelGGL(gauss_sample_kernel<false>, TMG_GS_ARGS);
#undef TMG_GS_ARGS
    TMG_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Benchmark loss of SURVEY 8-D (generative direction): loss = s1 sum(y^2) + s2 sum(logdet), s1 = 1 / numel(y), s2 = 1 / (B noc H W)
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reverse_loss_fwd_kernel(const float* __restrict__ y, size_t n4, size_t n, const float* __restrict__ ld,
                                                               int B, float s1, float s2, float* __restrict__ loss) {
    __shared__ float red[4];
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = ((const float4*)y)[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0) {
        for (size_t i = 4 * n4 + threadIdx.x; i < n; i += 256) acc += y[i] * y[i];      // tail of a size that is not a multiple of 4
        acc *= s1;
        for (int b = threadIdx.x; b < B; b += 256) acc += s2 * ld[b];
    } else {
        acc *= s1;
    }
    const float tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) atomicAdd(loss, tot);
}

__global__ __launch_bounds__(256) void reverse_loss_bwd_kernel(const float* __restrict__ y, size_t n4, size_t n, const float* __restrict__ g,
                                                               float s1, float s2, float* __restrict__ dy, float* __restrict__ dld, int B) {
    const float gv = g[0];
    const float a = 2.f * s1 * gv;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = ((const float4*)y)[i];
        v.x *= a; v.y *= a; v.z *= a; v.w *= a;
        ((float4*)dy)[i] = v;
    }
    if (blockIdx.x == 0) {
        for (size_t i = 4 * n4 + threadIdx.x; i < n; i += 256) dy[i] = a * y[i];
        for (int b = threadIdx.x; b < B; b += 256) dld[b] = s2 * gv;
    }
}

// y: n contiguous floats (16-byte aligned); loss: one float, zero on entry; dims = {n, B}; fl = {s1, s2}
extern "C" int tmg_reverse_loss_fwd(const void* y, const void* ld, void* loss, const int64_t* dims, const float* fl, hipStream_t st) {
    const size_t n = (size_t)dims[0];
    if (((uintptr_t)y & 15) != 0) return -2;
    hipLaunchKernelGGL(reverse_loss_fwd_kernel, dim3(glue_grid(n / 4, 1024)), dim3(256), 0, st, (const float*)y, n / 4, n, (const float*)ld,
                       (int)dims[1], fl[0], fl[1], (float*)loss);
    TMG_CHECK_LAUNCH();
    return 0;
}

extern "C" int tmg_reverse_loss_bwd(const void* y, const void* g, void* dy, void* dld, const int64_t* dims, const float* fl, hipStream_t st) {
    const size_t n = (size_t)dims[0];
    if ((((uintptr_t)y | (uintptr_t)dy) & 15) != 0) return -2;
    hipLaunchKernelGGL(reverse_loss_bwd_kernel, dim3(glue_grid(n / 4)), dim3(256), 0, st, (const float*)y, n / 4, n, (const float*)g, fl[0], fl[1],
                       (float*)dy, (float*)dld, (int)dims[1]);
    TMG_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// out[b] = sum_k term_k[b]   (len_k == 1: a scalar term, broadcast);  gsum[k] (optional) = nothing here - see sum_terms_bwd
// ---------------------------------------------------------------------------------------------------------------------------
struct SumTermsP {
    const float* t[TMG_SUM_TERMS_MAX];
    int len[TMG_SUM_TERMS_MAX];
    int n;
};

__global__ void sum_terms_kernel(SumTermsP p, float* __restrict__ out, int B) {
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < TMG_SUM_TERMS_MAX; ++k)
            if (k < p.n) acc += p.t[k][p.len[k] == 1 ? 0 : b];
        out[b] = acc;
    }
}

// out[0] = sum_b g[b]  (the gradient of a broadcast scalar term)
__global__ __launch_bounds__(256) void vec_sum_kernel(const float* __restrict__ g, int B, float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) acc += g[b];
    const float tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) out[0] = tot;
}

extern "C" int tmg_sum_terms(const void* const* terms, const int64_t* lens, int64_t n, void* out, int64_t B, hipStream_t st) {
    if (n < 1 || n > TMG_SUM_TERMS_MAX) return -2;
    SumTermsP p;
    p.n = (int)n;
    for (int k = 0; k < TMG_SUM_TERMS_MAX; ++k) {
        p.t[k] = k < n ? (const float*)terms[k] : nullptr;
        p.len[k] = k < n ? (int)lens[k] : 0;
        if (k < n && lens[k] != 1 && lens[k] != B) return -3;
    }
    hipLaunchKernelGGL(sum_terms_kernel, dim3((int)((B + 255) / 256)), dim3(256), 0, st, p, (float*)out, (int)B);
    TMG_CHECK_LAUNCH();
    return 0;
}

extern "C" int tmg_vec_sum(const void* g, int64_t B, void* out, hipStream_t st) {
    hipLaunchKernelGGL(vec_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)g, (int)B, (float*)out);
    TMG_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Parameter-side operands of a level node (LevelCouplingFn): per layer k the module's own tensors
//   w1 [1][cin][3][3], w2 [1][cin + 1][3][3], wz [C][cin + 2][3][3], bz [C], kappa [1]      (cin = ch + Cc)
// are gathered through a device pointer table tab[k][5] into
//   Wz   [NL][C][cin + 2][3][3]         the stacked zero-conv weights (operand packing, level_finish)
//   Wcat [NL C + 2 NLp][Cc][3][3]       rows k C + o: the conditioning columns of wz_k (Wzc); rows NL C + 2 k, + 2 k + 1: the
//                                       conditioning columns of w1_k / w2_k (Wdc; the rows of the NLp - NL padding layers are zero)
//   Bz   [NL][C],  Kp [NL]
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void level_pack_kernel(const long long* __restrict__ tab, float* __restrict__ Wz, float* __restrict__ Wcat, float* __restrict__ Bz,
                                  float* __restrict__ Kp, int NL, int NLp, int C, int ch, int Cc) {
    const int cin = ch + Cc;
    const size_t nWz = (size_t)NL * C * (cin + 2) * 9;
    const size_t nWzc = (size_t)NL * C * Cc * 9;
    const size_t nWdc = (size_t)2 * NLp * Cc * 9;
    const size_t nB = (size_t)NL * C;
    const size_t total = nWz + nWzc + nWdc + nB + NL;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        if (i < nWz) {
            const size_t per = (size_t)C * (cin + 2) * 9;
            const int k = (int)(i / per);
            const float* wz = (const float*)tab[5 * k + 2];
            Wz[i] = wz[i - (size_t)k * per];
        } else if (i < nWz + nWzc) {
            const size_t r = i - nWz;                       // [k][o][c][tap]
            const int tap = (int)(r % 9);
            size_t q = r / 9;
            const int c = (int)(q % Cc);
            q /= Cc;
            const int o = (int)(q % C);
            const int k = (int)(q / C);
            const float* wz = (const float*)tab[5 * k + 2];
            Wcat[r] = wz[((size_t)o * (cin + 2) + ch + c) * 9 + tap];
        } else if (i < nWz + nWzc + nWdc) {
            const size_t r = i - nWz - nWzc;                // [kp][which][c][tap]
            const int tap = (int)(r % 9);
            size_t q = r / 9;
            const int c = (int)(q % Cc);
            q /= Cc;
            const int which = (int)(q & 1);
            const int k = (int)(q >> 1);
            float v = 0.f;
            if (k < NL) {
                const float* w = (const float*)tab[5 * k + which];        // w1: [cin][9], w2: [cin + 1][9]; conditioning columns at ch..
                v = w[(size_t)(ch + c) * 9 + tap];
            }
            Wcat[nWzc + r] = v;
        } else if (i < nWz + nWzc + nWdc + nB) {
            const size_t r = i - nWz - nWzc - nWdc;
            const int k = (int)(r / C);
            Bz[r] = ((const float*)tab[5 * k + 3])[r - (size_t)k * C];
        } else {
            const int k = (int)(i - nWz - nWzc - nWdc - nB);
            Kp[k] = ((const float*)tab[5 * k + 4])[0];
        }
    }
}

// dims = {NL, NLp, C, ch, Cc}
extern "C" int tmg_level_pack(const void* tab, void* Wz, void* Wcat, void* Bz, void* Kp, const int64_t* dims, hipStream_t st) {
    const int NL = (int)dims[0], NLp = (int)dims[1], C = (int)dims[2], ch = (int)dims[3], Cc = (int)dims[4];
    const size_t total = (size_t)NL * C * (ch + Cc + 2) * 9 + (size_t)NL * C * Cc * 9 + (size_t)2 * NLp * Cc * 9 + (size_t)NL * C + NL;
    hipLaunchKernelGGL(level_pack_kernel, dim3(glue_grid(total, 2048)), dim3(256), 0, st, (const long long*)tab, (float*)Wz, (float*)Wcat,
                       (float*)Bz, (float*)Kp, NL, NLp, C, ch, Cc);
    TMG_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// up[b][y][x][c] = (y, x both even) ? dy[b][y / 2][x / 2][c] : 0 - the operand of the stride-2 input gradient on the matrix cores
// (tmg_ops.ConvFn.backward: dx = stride-1 correlation of the flipped taps with dy spread onto the even positions of a zero grid).
// One launch writes the whole grid (round 5: a zero fill of the grid + a strided torch copy).  float4 over channels (C % 4 == 0).
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void spread2_kernel(const float* __restrict__ dy, int dys, float* __restrict__ up, int B, int H, int W, int h, int w, int C4) {
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t r = i / C4;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(x & 1) && !(y & 1) && (y >> 1) < h && (x >> 1) < w)
            v = *reinterpret_cast<const float4*>(dy + (((size_t)b * h + (y >> 1)) * w + (x >> 1)) * dys + 4 * c4);
        reinterpret_cast<float4*>(up)[i] = v;
    }
}

// dims = {B, H, W (of the grid), h, w (of dy), C}; dy_d = {pixel stride, channel offset}; up: contiguous [B][H][W][C]
extern "C" int tmg_spread2(const void* dy, const int64_t* dy_d, void* up, const int64_t* dims, hipStream_t st) {
    const int C = (int)dims[5];
    if ((C & 3) || (dy_d[0] & 3) || (dy_d[1] & 3) || ((((uintptr_t)dy) | ((uintptr_t)up)) & 15)) return -100;
    const size_t total = (size_t)dims[0] * dims[1] * dims[2] * (C / 4);
    hipLaunchKernelGGL(spread2_kernel, dim3(glue_grid(total)), dim3(256), 0, st, (const float*)dy + dy_d[1], (int)dy_d[0], (float*)up, (int)dims[0],
                       (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4], C / 4);
    TMG_CHECK_LAUNCH();
    return 0;
}
