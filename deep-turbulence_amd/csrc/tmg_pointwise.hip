// Bandwidth-bound kernels of the TM-Glow hot path (gfx950, fp32, NHWC):
//   affine coupling apply + per-sample log-det and its backward   (reference flowAffine.py:76-83, :102-109)
//   ConvLSTM gate pointwise and its backward                      (reference convLSTM.py:76-83)
//   diagonal-Gaussian split prior: log-prob / eps / sample / bwd  (reference flowUtils.py:176-209, :274-275, :307-334)
//   checker squeeze / un-squeeze                                  (reference flowUtils.py:114-122, :137-145)
//   bilinear align_corners up-sampling and its adjoint            (reference misc.py:34-35)
//   BatchNorm batch moments and backward                          (reference denseBlock.py:49)
//   the growth-1 dense layers of the coupling network (C_out = 1) (reference denseBlock.py:135-138)
//   masked gradient scatter (ReLU masks + channel-concat adjoint)
// Every tensor argument is (pointer, pixel stride in floats, channel offset) so channel slices of
// wider NHWC buffers are addressed in place.
#include "tmg_common.h"
#include <stdlib.h>

#define LN5 1.6094379124341003f
#define LOG2PI 1.8378770664093453f

static inline int grid_for(size_t n, int cap = 4096) {
    size_t b = (n + 255) / 256;
    return (int)(b < (size_t)cap ? (b ? b : 1) : cap);
}

// ---------------------------------------------------------------------------------------------
// affine coupling
// ---------------------------------------------------------------------------------------------
// hh: [npix][C] interleaved (shift_j, r_j).  x2 -> y2 over C/2 channels.  logdet[b] += sum 2*softsign(r).
__global__ __launch_bounds__(256) void affine_apply_kernel(const float* __restrict__ hh, int hs, int ho, const float* x2, int xs,
                                                           int xo, float* y2, int ys, int yo, float* __restrict__ rsave,
                                                           float* __restrict__ logdet, int pix_per_img, int Ch, int reverse, int vec,
                                                           const float* x1, int x1s, float* y1, int y1s) {
    // x1 / y1 (optional): the pass-through half of the coupling, copied here instead of in a launch of its own
    __shared__ float red[4];
    const int b = blockIdx.y;
    const size_t base = (size_t)b * pix_per_img;
    const size_t total = (size_t)pix_per_img * Ch;
    float ld = 0.f;
    if (vec) {
        // 4 channel pairs per thread: two float4 of hh, one float4 of x2 / y2 / r
        const int c4n = Ch >> 2;
        const size_t tot4 = (size_t)pix_per_img * c4n;
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < tot4; i += (size_t)gridDim.x * 256) {
            const size_t pix = base + i / c4n;
            const int j = (int)(i % c4n) * 4;
            const float4 ha = *reinterpret_cast<const float4*>(hh + pix * hs + ho + 2 * j);
            const float4 hb = *reinterpret_cast<const float4*>(hh + pix * hs + ho + 2 * j + 4);
            const float4 xv = *reinterpret_cast<const float4*>(x2 + pix * xs + xo + j);
            const float sh[4] = {ha.x, ha.z, hb.x, hb.z}, rr[4] = {ha.y, ha.w, hb.y, hb.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
            float oo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sg = 2.f * rr[e] / (1.f + fabsf(rr[e]));
                oo[e] = reverse ? xx[e] * expf(-sg) - sh[e] : (xx[e] + sh[e]) * expf(sg);
                ld += sg;
            }
            *reinterpret_cast<float4*>(y2 + pix * ys + yo + j) = make_float4(oo[0], oo[1], oo[2], oo[3]);
            if (rsave) *reinterpret_cast<float4*>(rsave + pix * Ch + j) = make_float4(rr[0], rr[1], rr[2], rr[3]);
            if (y1) *reinterpret_cast<float4*>(y1 + pix * y1s + j) = *reinterpret_cast<const float4*>(x1 + pix * x1s + j);
        }
    } else
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t pix = base + i / Ch;
        const int j = i % Ch;
        const float2 h2 = *reinterpret_cast<const float2*>(hh + pix * hs + ho + 2 * j);
        const float r = h2.y;
        const float sg = 2.f * r / (1.f + fabsf(r));
        const float xv = x2[pix * xs + xo + j];
        float out;
        if (reverse) out = xv * expf(-sg) - h2.x;
        else out = (xv + h2.x) * expf(sg);
        y2[pix * ys + yo + j] = out;
        if (rsave) rsave[pix * Ch + j] = r;
        if (y1) y1[pix * y1s + j] = x1[pix * x1s + j];
        ld += sg;
    }
    const float tot = block_sum_256(ld, red);
    if (threadIdx.x == 0) atomicAdd(logdet + b, tot);
}

// gout: grad w.r.t. op output half; yref: forward -> op OUTPUT y2, reverse -> op INPUT y2.
// gin: grad w.r.t. op input half; dhh: [npix][C] interleaved (da, dr).  g: per-sample grad on logdet.
__global__ void affine_bwd_kernel(const float* gout, int gs, int go, const float* __restrict__ yref, int rs_, int ro,
                                  const float* __restrict__ rsave, const float* __restrict__ g, float* gin, int is, int io,
                                  float* __restrict__ dhh, int ds, int dof, int pix_per_img, int Ch, size_t npix, int reverse,
                                  const float* __restrict__ kappa) {
    const float hsc = out_scale_of(kappa);   // kappa given: dhh is written pre-multiplied by exp(clamp(kappa))
    const size_t total = npix * Ch;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / Ch;
        const int j = i % Ch;
        const int b = (int)(pix / pix_per_img);
        const float r = rsave[pix * Ch + j];
        const float den = 1.f + fabsf(r);
        const float sg = 2.f * r / den;
        const float go_ = gout[pix * gs + go + j];
        const float yv = yref[pix * rs_ + ro + j];
        const float gb = g ? g[b] : 0.f;
        float gi, da, dsg;
        if (reverse) {
            const float inv = expf(-sg);
            gi = go_ * inv;
            da = -go_;
            dsg = -2.f * go_ * (yv * inv) + 2.f * gb;
        } else {
            const float sc = expf(sg);
            gi = go_ * sc;
            da = gi;
            dsg = 2.f * go_ * yv + 2.f * gb;
        }
        gin[pix * is + io + j] = gi;
        *reinterpret_cast<float2*>(dhh + pix * ds + dof + 2 * j) = make_float2(da * hsc, hsc * dsg / (den * den));
    }
}

// ---------------------------------------------------------------------------------------------
// ConvLSTM pointwise.  gates: [npix][4R] pre-activation in the order i,f,o,g; left as they are (kept for
// backward, which evaluates the same activation functions on them again: the activated gates are never
// written - 4R floats per pixel less HBM traffic in the forward pass, bit-identical values in backward).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void lstm_pointwise_fwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev, int cps, int cpo,
                                          float* __restrict__ c_next, float* __restrict__ h_next, int R, size_t npix) {
    const size_t total = npix * R;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / R;
        const int j = i % R;
        const float* gp = gates + pix * 4 * R + j;
        const float gi = sigmoidf_(gp[0]), gf = sigmoidf_(gp[R]), go = sigmoidf_(gp[2 * R]), gg = tanhf(gp[3 * R]);
        const float cp = c_prev ? c_prev[pix * cps + cpo + j] : 0.f;
        const float cn = gf * cp + gi * gg;
        c_next[pix * R + j] = cn;
        h_next[pix * R + j] = go * tanhf(cn);
    }
}

// The same, four consecutive hidden channels per thread (float4 loads / stores, 32-bit index arithmetic): R and the c_prev stride /
// offset multiples of 4, 16-byte aligned tensors, fewer than 2^31 quads (the launcher checks).  Element for element the arithmetic
// of the scalar kernel.  Rq = R / 4.
#define TMG_F4(V) {(V).x, (V).y, (V).z, (V).w}
struct LstmB1 { float r0, r1, r2, r3, dp; };
// one hidden channel of lstm_pointwise_bwd_kernel: pre-activation gate gradients (i, f, o, g) and the gradient of the previous cell state
__device__ __forceinline__ LstmB1 lstm_bwd1(float ai, float af, float ao, float ag, float cp, float cnx, float dhv, float dci) {
    const float gi = sigmoidf_(ai), gf = sigmoidf_(af), go = sigmoidf_(ao), gg = tanhf(ag);
    const float tc = tanhf(cnx);
    const float dc = dci + dhv * go * (1.f - tc * tc);
    LstmB1 o;
    o.r0 = dc * gg * gi * (1.f - gi);
    o.r1 = dc * cp * gf * (1.f - gf);
    o.r2 = dhv * tc * go * (1.f - go);
    o.r3 = dc * gi * (1.f - gg * gg);
    o.dp = dc * gf;
    return o;
}
__global__ __launch_bounds__(256) void lstm_pointwise_fwd4_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev, int cps, int cpo,
                                           float* __restrict__ c_next, float* __restrict__ h_next, int Rq, unsigned npix) {
    const unsigned total = npix * (unsigned)Rq;
    const int R = 4 * Rq;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned pix = i / (unsigned)Rq;
        const int j = 4 * (int)(i - pix * (unsigned)Rq);
        const float* gp = gates + (size_t)pix * 4 * R + j;
        const float4 a4 = *reinterpret_cast<const float4*>(gp), f4 = *reinterpret_cast<const float4*>(gp + R);
        const float4 o4 = *reinterpret_cast<const float4*>(gp + 2 * R), g4 = *reinterpret_cast<const float4*>(gp + 3 * R);
        const float4 c4 = *reinterpret_cast<const float4*>(c_prev ? c_prev + (size_t)pix * cps + cpo + j : tmg_zero_page);   // (address select: a struct ternary goes through the stack)
        const float ai[4] = TMG_F4(a4), af[4] = TMG_F4(f4), ao[4] = TMG_F4(o4), ag[4] = TMG_F4(g4), cp[4] = TMG_F4(c4);
        float cn[4], hn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = sigmoidf_(ai[e]), gf = sigmoidf_(af[e]), go = sigmoidf_(ao[e]), gg = tanhf(ag[e]);
            cn[e] = gf * cp[e] + gi * gg;
            hn[e] = go * tanhf(cn[e]);
        }
        *reinterpret_cast<float4*>(c_next + (size_t)pix * R + j) = make_float4(cn[0], cn[1], cn[2], cn[3]);
        *reinterpret_cast<float4*>(h_next + (size_t)pix * R + j) = make_float4(hn[0], hn[1], hn[2], hn[3]);
    }
}

__global__ __launch_bounds__(256) void lstm_pointwise_bwd4_kernel(float* __restrict__ acts, const float* __restrict__ c_prev, int cps, int cpo,
                                           const float* __restrict__ c_next, const float* __restrict__ dh,
                                           const float* __restrict__ dc_in, float* __restrict__ dc_prev, int Rq, unsigned npix) {
    const unsigned total = npix * (unsigned)Rq;
    const int R = 4 * Rq;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned pix = i / (unsigned)Rq;
        const int j = 4 * (int)(i - pix * (unsigned)Rq);
        float* gp = acts + (size_t)pix * 4 * R + j;
        const size_t pj = (size_t)pix * R + j;
        const float4 a4 = *reinterpret_cast<const float4*>(gp), f4 = *reinterpret_cast<const float4*>(gp + R);
        const float4 o4 = *reinterpret_cast<const float4*>(gp + 2 * R), g4 = *reinterpret_cast<const float4*>(gp + 3 * R);
        const float4 c4 = *reinterpret_cast<const float4*>(c_prev ? c_prev + (size_t)pix * cps + cpo + j : tmg_zero_page);
        const float4 n4 = *reinterpret_cast<const float4*>(c_next + pj);
        const float4 h4 = *reinterpret_cast<const float4*>(dh ? dh + pj : tmg_zero_page);
        const float4 d4 = *reinterpret_cast<const float4*>(dc_in ? dc_in + pj : tmg_zero_page);
        const LstmB1 ex = lstm_bwd1(a4.x, f4.x, o4.x, g4.x, c4.x, n4.x, h4.x, d4.x), ey = lstm_bwd1(a4.y, f4.y, o4.y, g4.y, c4.y, n4.y, h4.y, d4.y);
        const LstmB1 ez = lstm_bwd1(a4.z, f4.z, o4.z, g4.z, c4.z, n4.z, h4.z, d4.z), ew = lstm_bwd1(a4.w, f4.w, o4.w, g4.w, c4.w, n4.w, h4.w, d4.w);
        *reinterpret_cast<float4*>(gp) = make_float4(ex.r0, ey.r0, ez.r0, ew.r0);
        *reinterpret_cast<float4*>(gp + R) = make_float4(ex.r1, ey.r1, ez.r1, ew.r1);
        *reinterpret_cast<float4*>(gp + 2 * R) = make_float4(ex.r2, ey.r2, ez.r2, ew.r2);
        *reinterpret_cast<float4*>(gp + 3 * R) = make_float4(ex.r3, ey.r3, ez.r3, ew.r3);
        if (dc_prev) *reinterpret_cast<float4*>(dc_prev + pj) = make_float4(ex.dp, ey.dp, ez.dp, ew.dp);
    }
}
#undef TMG_F4

// acts: the pre-activation gates of the forward pass, overwritten in place by the pre-activation gradients.
__global__ void lstm_pointwise_bwd_kernel(float* __restrict__ acts, const float* __restrict__ c_prev, int cps, int cpo,
                                          const float* __restrict__ c_next, const float* __restrict__ dh,
                                          const float* __restrict__ dc_in, float* __restrict__ dc_prev, int R, size_t npix) {
    const size_t total = npix * R;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / R;
        const int j = i % R;
        float* gp = acts + pix * 4 * R + j;
        const float gi = sigmoidf_(gp[0]), gf = sigmoidf_(gp[R]), go = sigmoidf_(gp[2 * R]), gg = tanhf(gp[3 * R]);
        const float cp = c_prev ? c_prev[pix * cps + cpo + j] : 0.f;
        const float tc = tanhf(c_next[pix * R + j]);
        const float dhv = dh ? dh[pix * R + j] : 0.f;
        const float dc = (dc_in ? dc_in[pix * R + j] : 0.f) + dhv * go * (1.f - tc * tc);
        gp[0] = dc * gg * gi * (1.f - gi);
        gp[R] = dc * cp * gf * (1.f - gf);
        gp[2 * R] = dhv * tc * go * (1.f - go);
        gp[3 * R] = dc * gi * (1.f - gg * gg);
        if (dc_prev) dc_prev[pix * R + j] = dc * gf;     // null: the previous cell state carries no gradient (nothing to write)
    }
}

// ---------------------------------------------------------------------------------------------
// Diagonal Gaussian.  hz: [npix][2*Ch] = (mean | log-std) halves.  mean clipped to [mlo,mhi] when
// clip_mean (split prior: hardtanh(-2, ln5) hits both halves), log-std clipped to [slo, shi].
// mode 0 (x->z): given z2: logp[b] += sum -0.5(ln2pi + 2 lsd + (z2-mean)^2 e^{-2 lsd}); eps out optional.
// mode 1 (z->x): given eps: z2 = mean + e^{lsd} eps (written), logp[b] += sum -0.5(ln2pi + 2 lsd + eps^2).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gauss_fwd_kernel(const float* __restrict__ hz, int hs, int ho, const float* zin, int zs,
                                                        int zo, float* zout, int os, int oo, float* __restrict__ logp,
                                                        int pix_per_img, int Ch, int mode, int clip_mean, float mlo, float mhi,
                                                        float slo, float shi) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const size_t base = (size_t)b * pix_per_img;
    const unsigned total = (unsigned)pix_per_img * (unsigned)Ch;   // per image: < 2^31 (checked by the launcher); 32-bit div / mod
    float lp = 0.f;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned pl = i / (unsigned)Ch;
        const size_t pix = base + pl;
        const int j = (int)(i - pl * (unsigned)Ch);
        float mean = hz[pix * hs + ho + j];
        float lsd = hz[pix * hs + ho + Ch + j];
        if (clip_mean) mean = fminf(fmaxf(mean, mlo), mhi);
        lsd = fminf(fmaxf(lsd, slo), shi);
        const float v = zin[pix * zs + zo + j];
        if (mode == 0) {
            const float e = (v - mean) * expf(-lsd);
            lp += -0.5f * (LOG2PI + 2.f * lsd + e * e);
            if (zout) zout[pix * os + oo + j] = e;
        } else {
            lp += -0.5f * (LOG2PI + 2.f * lsd + v * v);
            zout[pix * os + oo + j] = mean + expf(lsd) * v;
        }
    }
    const float tot = block_sum_256(lp, red);
    if (threadIdx.x == 0) atomicAdd(logp + b, tot);
}

// mode 0: inputs z2 (zin), g[b]; outputs dz2 (dzout, optional) and dhz (grad w.r.t. raw hz, clip masks applied).
// mode 1: inputs eps (zin), dz2 (dzin: grad w.r.t. sampled z2), g[b]; outputs dhz.
__global__ void gauss_bwd_kernel(const float* __restrict__ hz, int hs, int ho, const float* zin, int zs, int zo,
                                 const float* dzin, int dis, int dio, const float* __restrict__ g, float* dzout, int dos,
                                 int doo, float* __restrict__ dhz, int ds, int dof, int pix_per_img, int Ch, size_t npix,
                                 int mode, int clip_mean, float mlo, float mhi, float slo, float shi) {
    const size_t total = npix * Ch;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / Ch;
        const int j = i % Ch;
        const int b = (int)(pix / pix_per_img);
        const float mraw = hz[pix * hs + ho + j], sraw = hz[pix * hs + ho + Ch + j];
        float mean = mraw, lsd = fminf(fmaxf(sraw, slo), shi);
        bool mpass = true;
        if (clip_mean) {
            mean = fminf(fmaxf(mraw, mlo), mhi);
            mpass = (mraw > mlo) && (mraw < mhi);
        }
        const bool spass = (sraw > slo) && (sraw < shi);
        const float gb = g ? g[b] : 0.f;
        const float v = zin[pix * zs + zo + j];
        float dmean, dlsd;
        if (mode == 0) {
            const float il = expf(-lsd);
            const float e = (v - mean) * il;
            dmean = gb * e * il;
            dlsd = gb * (e * e - 1.f);
            float dz = -gb * e * il;
            if (dzin) dz += dzin[pix * dis + dio + j];
            if (dzout) dzout[pix * dos + doo + j] = dz;
        } else {
            const float dz = dzin ? dzin[pix * dis + dio + j] : 0.f;
            dmean = dz;
            dlsd = dz * expf(lsd) * v - gb;
        }
        dhz[pix * ds + dof + j] = mpass ? dmean : 0.f;
        dhz[pix * ds + dof + Ch + j] = spass ? dlsd : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// checker squeeze: big [B,2h,2w,C] <-> small [B,h,w,4C]; block k takes (row,col) offset (0,0),(1,0),(1,1),(0,1)
// ---------------------------------------------------------------------------------------------
__global__ void checker_kernel(const float* __restrict__ src, int ss, int so, float* __restrict__ dst, int ds, int dof, int B,
                               int h, int w, int C, int to_small) {
    const size_t total = (size_t)B * h * w * 4 * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % (4 * C);
        size_t r = i / (4 * C);
        const int x = r % w;
        r /= w;
        const int y = r % h;
        const int b = r / h;
        const int k = c4 / C, c = c4 - k * C;
        const int ry = (k == 1 || k == 2) ? 1 : 0, rx = (k >= 2) ? 1 : 0;
        const size_t bigpix = ((size_t)b * 2 * h + 2 * y + ry) * (2 * w) + 2 * x + rx;
        const size_t smallpix = ((size_t)b * h + y) * w + x;
        if (to_small) dst[smallpix * ds + dof + c4] = src[bigpix * ss + so + c];
        else dst[bigpix * ds + dof + c] = src[smallpix * ss + so + c4];
    }
}

// The same move one channel quad (16 bytes) per thread and iteration, 32-bit index arithmetic: C, strides and offsets multiples of 4,
// 16-byte aligned tensors, fewer than 2^31 quads (the launcher checks).  Cq = C / 4.
__global__ __launch_bounds__(256) void checker4_kernel(const float* __restrict__ src, int ss, int so, float* __restrict__ dst, int ds, int dof, int B,
                                int h, int w, int Cq, int to_small) {
    const unsigned total = (unsigned)B * (unsigned)h * (unsigned)w * 4u * (unsigned)Cq;
    const unsigned q4 = 4u * (unsigned)Cq;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned kq = i % q4;
        unsigned r = i / q4;
        const unsigned x = r % (unsigned)w;
        r /= (unsigned)w;
        const unsigned y = r % (unsigned)h;
        const unsigned b = r / (unsigned)h;
        const unsigned k = kq / (unsigned)Cq, c = 4u * (kq - k * (unsigned)Cq);
        const unsigned ry = (k == 1 || k == 2) ? 1u : 0u, rx = (k >= 2) ? 1u : 0u;
        const size_t bigpix = ((size_t)b * 2 * h + 2 * y + ry) * (2 * w) + 2 * x + rx;
        const size_t smallpix = ((size_t)b * h + y) * w + x;
        if (to_small) *reinterpret_cast<float4*>(dst + smallpix * ds + dof + 4u * kq) = *reinterpret_cast<const float4*>(src + bigpix * ss + so + c);
        else *reinterpret_cast<float4*>(dst + bigpix * ds + dof + c) = *reinterpret_cast<const float4*>(src + smallpix * ss + so + 4u * kq);
    }
}

// ---------------------------------------------------------------------------------------------
// zero-padded channel halves: compact [npix][2 ch] <-> padded [npix][2 (ch + pad)] = [x1 | 0.. | x2 | 0..]
// (3-channel fields: ch = 6 on the first level; every kernel of the fast path wants float4-addressable halves)
// to_padded: every element of the padded tensor is written (the padding channels with zeros): no fill launch
// ---------------------------------------------------------------------------------------------
__global__ void pad_halves_kernel(const float* __restrict__ src, int ss, float* __restrict__ dst, int ds, size_t npix, int ch, int pad,
                                  int to_padded) {
    const int chp = ch + pad;
    const int cw = to_padded ? 2 * chp : 2 * ch;        // channels written per pixel
    const size_t total = npix * (size_t)cw;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cw);
        const size_t pix = i / cw;
        if (to_padded) {
            const int half = c >= chp, k = c - half * chp;
            dst[pix * ds + c] = k < ch ? src[pix * ss + half * ch + k] : 0.f;
        } else {
            const int half = c >= ch, k = c - half * ch;
            dst[pix * ds + c] = src[pix * ss + half * chp + k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bilinear up-sampling, align_corners = True
// ---------------------------------------------------------------------------------------------
__global__ void upsample_fwd_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int hi, int wi, int ho, int wo,
                                    int C) {
    const float ry = ho > 1 ? (float)(hi - 1) / (float)(ho - 1) : 0.f;
    const float rx = wo > 1 ? (float)(wi - 1) / (float)(wo - 1) : 0.f;
    const size_t total = (size_t)B * ho * wo * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % C;
        size_t r = i / C;
        const int ox = r % wo;
        r /= wo;
        const int oy = r % ho;
        const int b = r / ho;
        const float sy = ry * oy, sx = rx * ox;
        int y0 = (int)sy, x0 = (int)sx;
        y0 = min(y0, hi - 1);
        x0 = min(x0, wi - 1);
        const int y1 = min(y0 + 1, hi - 1), x1 = min(x0 + 1, wi - 1);
        const float fy = sy - y0, fx = sx - x0;
        const float* sb = src + (size_t)b * hi * wi * C + c;
        const float v00 = sb[((size_t)y0 * wi + x0) * C], v01 = sb[((size_t)y0 * wi + x1) * C];
        const float v10 = sb[((size_t)y1 * wi + x0) * C], v11 = sb[((size_t)y1 * wi + x1) * C];
        dst[i] = (1.f - fy) * ((1.f - fx) * v00 + fx * v01) + fy * ((1.f - fx) * v10 + fx * v11);
    }
}

// adjoint as a gather: each low-res element sums the hat-function weights of the outputs that touch it
__global__ void upsample_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int B, int hi, int wi, int ho, int wo,
                                    int C) {
    const float ry = ho > 1 ? (float)(hi - 1) / (float)(ho - 1) : 0.f;
    const float rx = wo > 1 ? (float)(wi - 1) / (float)(wo - 1) : 0.f;
    const size_t total = (size_t)B * hi * wi * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % C;
        size_t r = i / C;
        const int ix = r % wi;
        r /= wi;
        const int iy = r % hi;
        const int b = r / hi;
        // outputs oy with source coordinate in (iy-1, iy+1)
        int oy_lo = 0, oy_hi = ho - 1, ox_lo = 0, ox_hi = wo - 1;
        if (ry > 0.f) {
            oy_lo = max(0, (int)floorf((iy - 1) / ry) - 1);
            oy_hi = min(ho - 1, (int)ceilf((iy + 1) / ry) + 1);
        }
        if (rx > 0.f) {
            ox_lo = max(0, (int)floorf((ix - 1) / rx) - 1);
            ox_hi = min(wo - 1, (int)ceilf((ix + 1) / rx) + 1);
        }
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const float sy = ry * oy;
            int y0 = min((int)sy, hi - 1);
            const int y1 = min(y0 + 1, hi - 1);
            const float fy = sy - y0;
            float wy = 0.f;
            if (y0 == iy) wy += 1.f - fy;
            if (y1 == iy) wy += fy;
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const float sx = rx * ox;
                int x0 = min((int)sx, wi - 1);
                const int x1 = min(x0 + 1, wi - 1);
                const float fx = sx - x0;
                float wx = 0.f;
                if (x0 == ix) wx += 1.f - fx;
                if (x1 == ix) wx += fx;
                if (wx == 0.f) continue;
                acc += wy * wx * dout[(((size_t)b * ho + oy) * wo + ox) * C + c];
            }
        }
        din[i] = acc;
    }
}

// The same two maps for C % 4 == 0 (every conditioning map of the path: 32 channels): one float4 of channels per thread, 32-bit index
// arithmetic (the scalar kernels above spend most of their time in 64-bit div / mod per element: 122 us for the 134 MB first-level map,
// a quarter of the achievable bandwidth), and - backward - the 1-D hat weights of the <= 6 candidate rows / columns evaluated once
// per thread instead of inside the 2-D loop.
__global__ __launch_bounds__(256) void upsample_fwd4_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int hi, int wi,
                                                            int ho, int wo, int C4) {
    const float ry = ho > 1 ? (float)(hi - 1) / (float)(ho - 1) : 0.f;
    const float rx = wo > 1 ? (float)(wi - 1) / (float)(wo - 1) : 0.f;
    const unsigned total = (unsigned)B * ho * wo * C4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned c4 = i % (unsigned)C4;
        unsigned r = i / (unsigned)C4;
        const int ox = r % (unsigned)wo;
        r /= (unsigned)wo;
        const int oy = r % (unsigned)ho;
        const int b = r / (unsigned)ho;
        const float sy = ry * oy, sx = rx * ox;
        const int y0 = min((int)sy, hi - 1), x0 = min((int)sx, wi - 1);
        const int y1 = min(y0 + 1, hi - 1), x1 = min(x0 + 1, wi - 1);
        const float fy = sy - y0, fx = sx - x0;
        const float4* sb = reinterpret_cast<const float4*>(src) + (size_t)b * hi * wi * C4 + c4;
        const float4 v00 = sb[(size_t)(y0 * wi + x0) * C4], v01 = sb[(size_t)(y0 * wi + x1) * C4];
        const float4 v10 = sb[(size_t)(y1 * wi + x0) * C4], v11 = sb[(size_t)(y1 * wi + x1) * C4];
        // (same expression, evaluated per component in the same order as the scalar kernel: bit-identical results)
        float4 o;
        o.x = (1.f - fy) * ((1.f - fx) * v00.x + fx * v01.x) + fy * ((1.f - fx) * v10.x + fx * v11.x);
        o.y = (1.f - fy) * ((1.f - fx) * v00.y + fx * v01.y) + fy * ((1.f - fx) * v10.y + fx * v11.y);
        o.z = (1.f - fy) * ((1.f - fx) * v00.z + fx * v01.z) + fy * ((1.f - fx) * v10.z + fx * v11.z);
        o.w = (1.f - fy) * ((1.f - fx) * v00.w + fx * v01.w) + fy * ((1.f - fx) * v10.w + fx * v11.w);
        reinterpret_cast<float4*>(dst)[i] = o;
    }
}

// hat weight of output coordinate o (source coordinate r * o) on input index idx, as the forward kernel rounds it
__device__ __forceinline__ float upsample_hat(float r, int o, int idx, int n_in) {
    const float sv = r * o;
    const int i0 = min((int)sv, n_in - 1), i1 = min(i0 + 1, n_in - 1);
    const float f = sv - i0;
    return (i0 == idx ? 1.f - f : 0.f) + (i1 == idx ? f : 0.f);
}

__global__ __launch_bounds__(256) void upsample_bwd4_kernel(const float* __restrict__ dout, float* __restrict__ din, int B, int hi, int wi,
                                                            int ho, int wo, int C4) {
    constexpr int NW = 8;   // candidate outputs per axis: (2 / r) + 3 <= 8 for up-scaling factors >= 0.4 (the launcher checks)
    const float ry = ho > 1 ? (float)(hi - 1) / (float)(ho - 1) : 0.f;
    const float rx = wo > 1 ? (float)(wi - 1) / (float)(wo - 1) : 0.f;
    const unsigned total = (unsigned)B * hi * wi * C4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned c4 = i % (unsigned)C4;
        unsigned r = i / (unsigned)C4;
        const int ix = r % (unsigned)wi;
        r /= (unsigned)wi;
        const int iy = r % (unsigned)hi;
        const int b = r / (unsigned)hi;
        int oy_lo = 0, ox_lo = 0;
        if (ry > 0.f) oy_lo = max(0, (int)floorf((iy - 1) / ry) - 1);
        if (rx > 0.f) ox_lo = max(0, (int)floorf((ix - 1) / rx) - 1);
        float wy[NW], wx[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            wy[k] = oy_lo + k < ho ? upsample_hat(ry, oy_lo + k, iy, hi) : 0.f;
            wx[k] = ox_lo + k < wo ? upsample_hat(rx, ox_lo + k, ix, wi) : 0.f;
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* db = reinterpret_cast<const float4*>(dout) + (size_t)b * ho * wo * C4 + c4;
#pragma unroll
        for (int ky = 0; ky < NW; ++ky) {
            if (wy[ky] == 0.f) continue;
#pragma unroll
            for (int kx = 0; kx < NW; ++kx) {
                if (wx[kx] == 0.f) continue;
                const float4 v = db[(size_t)((oy_lo + ky) * wo + ox_lo + kx) * C4];
                const float wgt = wy[ky] * wx[kx];
                acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
            }
        }
        reinterpret_cast<float4*>(din)[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// per-channel reductions over pixels (BatchNorm moments and backward sums)
//   mode 0: s0 += sum (x - shift_c),          s1 += sum (x - shift_c)^2
//   mode 1: du = g * [x*a_c + b_c > 0];  xhat = (x - mean_c) * rstd_c;  s0 += sum du, s1 += sum du * xhat
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float* __restrict__ x, int xs, int xo, const float* __restrict__ g,
                                                          int gs, int go, const float* __restrict__ v0, const float* __restrict__ v1,
                                                          const float* __restrict__ v2, const float* __restrict__ v3,
                                                          float* __restrict__ s0, float* __restrict__ s1, size_t npix, int C,
                                                          int mode, float v0_scale) {
    // mode 0: s0 += sum (x - v0*v0_scale), s1 += sum (x - v0*v0_scale)^2   (v0 null: plain sums; v0 = channel sums and
    //         v0_scale = 1/n: centred second pass without a separate mean kernel)
    // mode 1: BatchNorm+ReLU backward sums (see bn_bwd_apply_kernel)
    __shared__ float l0[256], l1[256];
    const int tid = threadIdx.x;
    const int lanes = 256 / C > 0 ? 256 / C : 1;  // pixel lanes per block (C <= 256)
    const int c = tid % C, pl = tid / C;
    float a0 = 0.f, a1 = 0.f;
    if (pl < lanes) {
        const float off = (mode == 0 && v0) ? v0[c] * v0_scale : 0.f;
        const size_t step = (size_t)gridDim.x * lanes;
        // four pixels per iteration: the loads are independent, a one-pixel loop is a chain of exposed latencies
        for (size_t pix = blockIdx.x * (size_t)lanes + pl; pix < npix; pix += 4 * step) {
            float xv[4], gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t pp = pix + u * step < npix ? pix + u * step : pix;
                xv[u] = x[pp * xs + xo + c];
                gv[u] = (mode != 0) ? g[pp * gs + go + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (pix + u * step < npix) {
                    if (mode == 0) {
                        const float d = xv[u] - off;
                        a0 += d;
                        a1 += d * d;
                    } else {
                        const float uu = xv[u] * v0[c] + v1[c];
                        const float du = uu > 0.f ? gv[u] : 0.f;
                        a0 += du;
                        a1 += du * (xv[u] - v2[c]) * v3[c];
                    }
                }
            }
        }
    }
    l0[tid] = a0;
    l1[tid] = a1;
    __syncthreads();
    if (tid < C) {
        float t0 = 0.f, t1 = 0.f;
        for (int k = 0; k < lanes; ++k) {
            t0 += l0[k * C + tid];
            t1 += l1[k * C + tid];
        }
        atomicAdd(s0 + tid, t0);
        atomicAdd(s1 + tid, t1);
    }
}

// BatchNorm batch moments in ONE pass over the activation: per-channel sum and sum of squares accumulated in fp64 (per thread, per
// block, and in the double atomics), so that var = E[x^2] - E[x]^2 carries no fp32 cancellation error; the centred two-pass form
// read every encoder activation twice (0.68 ms per step in 48 launches).  acc: double [2][C], zeroed by the caller.
__global__ __launch_bounds__(256) void chan_moments_kernel(const float* __restrict__ x, int xs, int xo, double* __restrict__ acc, size_t npix,
                                                           int C) {
    __shared__ double l0[256], l1[256];
    const int tid = threadIdx.x;
    const int lanes = 256 / C > 0 ? 256 / C : 1;
    const int c = tid % C, pl = tid / C;
    double a0 = 0.0, a1 = 0.0;
    if (pl < lanes) {
        const size_t step = (size_t)gridDim.x * lanes;
        for (size_t pix = blockIdx.x * (size_t)lanes + pl; pix < npix; pix += 4 * step) {
            float xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t pp = pix + u * step < npix ? pix + u * step : pix;
                xv[u] = x[pp * xs + xo + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (pix + u * step < npix) {
                    const double d = (double)xv[u];
                    a0 += d;
                    a1 = fma(d, d, a1);
                }
            }
        }
    }
    l0[tid] = a0;
    l1[tid] = a1;
    __syncthreads();
    if (tid < C) {
        double t0 = 0.0, t1 = 0.0;
        for (int k = 0; k < lanes; ++k) {
            t0 += l0[k * C + tid];
            t1 += l1[k * C + tid];
        }
        atomicAdd(acc + tid, t0);
        atomicAdd(acc + C + tid, t1);
    }
}

// bn_finalize_kernel on the fp64 moments of chan_moments_kernel: mean = S1 / n, var = S2 / n - mean^2 (biased), evaluated in fp64
__global__ void bn_finalize64_kernel(const double* __restrict__ acc, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ out, int C, double n, float eps,
                                     float momentum, long long* __restrict__ nbt) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (nbt && c == 0) *nbt += 1;      // nn.BatchNorm2d.num_batches_tracked (a launch of its own otherwise, 16 per step)
    const double m = acc[c] / n;
    double v = acc[C + c] / n - m * m;
    if (v < 0.0) v = 0.0;
    const float mean = (float)m, var = (float)v;
    const float rstd = rsqrtf(var + eps);
    const float a = gamma[c] * rstd;
    out[c] = mean;
    out[C + c] = var;
    out[2 * C + c] = rstd;
    out[3 * C + c] = a;
    out[4 * C + c] = beta[c] - mean * a;
    if (rmean) {
        const float nf = (float)n;
        rmean[c] = rmean[c] * (1.f - momentum) + momentum * mean;
        rvar[c] = rvar[c] * (1.f - momentum) + momentum * var * (nf / fmaxf(nf - 1.f, 1.f));
    }
}

// Batch statistics -> everything the BatchNorm(+ReLU) fold needs, in one launch of C threads:
//   mean = sum/n, var = centred_sq/n (biased), rstd, a = gamma*rstd, bsh = beta - mean*a  -> out[0..4][C]
//   running_mean / running_var (optional) updated in place with `momentum`, the variance unbiased (n/(n-1)) as nn.BatchNorm2d.
__global__ void bn_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ csq, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                   float* __restrict__ out, int C, float n, float eps, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = sum[c] / n, var = csq[c] / n;
    const float rstd = rsqrtf(var + eps);
    const float a = gamma[c] * rstd;
    out[c] = mean;
    out[C + c] = var;
    out[2 * C + c] = rstd;
    out[3 * C + c] = a;
    out[4 * C + c] = beta[c] - mean * a;
    if (rmean) {
        rmean[c] = rmean[c] * (1.f - momentum) + momentum * mean;
        rvar[c] = rvar[c] * (1.f - momentum) + momentum * var * (n / fmaxf(n - 1.f, 1.f));
    }
}

// BatchNorm(+ReLU) input gradient:  u = x*a_c + b_c,  du = g*[u>0],
//   dx = gamma_c * rstd_c * (du - m0_c - xhat * m1_c)     (m0 = mean du, m1 = mean du*xhat)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, int xs, int xo, const float* __restrict__ g, int gs, int go,
                                    const float* __restrict__ a, const float* __restrict__ bsh, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ m0,
                                    const float* __restrict__ m1, float* __restrict__ dx, int ds, int dof, size_t npix, int C,
                                    int accumulate, float m_scale) {
    const size_t total = npix * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / C;
        const int c = i % C;
        const float xv = x[pix * xs + xo + c];
        const float u = xv * a[c] + bsh[c];
        const float du = u > 0.f ? g[pix * gs + go + c] : 0.f;
        const float xh = (xv - mean[c]) * rstd[c];
        const float v = gamma[c] * rstd[c] * (du - m0[c] * m_scale - xh * m1[c] * m_scale);
        float* d = dx + pix * ds + dof + c;
        *d = accumulate ? (*d + v) : v;
    }
}

// ---------------------------------------------------------------------------------------------
// masked gradient scatter: dst[c] (+)= src[c] * [ref[c] > 0] (+ add[c])   over n channels
// ---------------------------------------------------------------------------------------------
__global__ void masked_add_kernel(const float* __restrict__ src, int ss, int so, const float* __restrict__ ref, int rs_, int ro,
                                  const float* __restrict__ add, int as, int ao, float* dst, int ds, int dof, size_t npix, int n,
                                  int accumulate) {
    const size_t total = npix * n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / n;
        const int c = i % n;
        float v = src ? src[pix * ss + so + c] : 0.f;
        if (ref && !(ref[pix * rs_ + ro + c] > 0.f)) v = 0.f;
        if (add) v += add[pix * as + ao + c];
        float* d = dst + pix * ds + dof + c;
        *d = accumulate ? (*d + v) : v;
    }
}

// The same one channel quad per thread and iteration, 32-bit index arithmetic (n, strides and offsets multiples of 4, 16-byte aligned
// tensors, fewer than 2^31 quads: checked by the launcher).  nq = n / 4.
__global__ __launch_bounds__(256) void masked_add4_kernel(const float* __restrict__ src, int ss, int so, const float* __restrict__ ref, int rs_, int ro,
                                   const float* __restrict__ add, int as, int ao, float* dst, int ds, int dof, unsigned npix, int nq,
                                   int accumulate) {
    const unsigned total = npix * (unsigned)nq;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned pix = i / (unsigned)nq;
        const unsigned c = 4u * (i - pix * (unsigned)nq);
        float4 v = src ? *reinterpret_cast<const float4*>(src + (size_t)pix * ss + so + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (ref) {
            const float4 r = *reinterpret_cast<const float4*>(ref + (size_t)pix * rs_ + ro + c);
            if (!(r.x > 0.f)) v.x = 0.f;
            if (!(r.y > 0.f)) v.y = 0.f;
            if (!(r.z > 0.f)) v.z = 0.f;
            if (!(r.w > 0.f)) v.w = 0.f;
        }
        if (add) {
            const float4 a = *reinterpret_cast<const float4*>(add + (size_t)pix * as + ao + c);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        float4* d = reinterpret_cast<float4*>(dst + (size_t)pix * ds + dof + c);
        if (accumulate) {
            const float4 o = *d;
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *d = v;
    }
}

// ---------------------------------------------------------------------------------------------
// growth-1 dense layer (C_out = 1): out[p] = sum_{tap,ci} w[ci][tap] * relu(in(p+tap))[ci], zero pad.
// Vector-ALU kernel: one pixel per thread over a 256-pixel tile, input patch through LDS in chunks.
// ---------------------------------------------------------------------------------------------
struct C1P {
    TmgSeg in[TMG_MAX_IN_SEG];
    int nseg, vec4;
    int B, Hin, Win;
    int Cin;
    const float* in_scale;
    const float* in_shift;
    int relu_in, pad_rep;
    const float* w;  // [rows][9]; input channel c < w_rows reads row c (+ w_gap if c >= w_split); others have zero weight
    int w_rows, w_split, w_gap;
    const float* add; int add_stride, add_off;  // optional per-pixel value added to the result (null: none)
    int fill4;  // write (value, 0, 0, 0) as one float4: initialises the 4-channel growth buffer D without a separate fill
    float* out; int out_stride, out_off;
    int TW_log2, tiles_x, tiles_y, KCH;
};

__global__ __launch_bounds__(256) void c1_fwd_kernel(C1P p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int TWl = p.TW_log2, TW = 1 << TWl, TH = 256 >> TWl;
    const int PW = TW + 2, PH = TH + 2;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int row = tid >> TWl, col = tid & (TW - 1);
    const int Cpad = (p.Cin + 3) & ~3;
    float acc = 0.f;
    for (int c0 = 0; c0 < Cpad; c0 += p.KCH) {
        const int kch = min(p.KCH, Cpad - c0);
        const int CS = kch + 4;
        float* lw = lds + PH * PW * CS;  // [9][kch]
        __syncthreads();
        stage_patch_bf(p, lds, b, oy0 - 1, ox0 - 1, PH, PW, c0, kch, CS);
        for (int i = tid; i < 9 * kch; i += 256) {
            const int tap = i / kch, c = i - tap * kch;
            const int cs = c0 + c;
            lw[i] = (cs < p.w_rows) ? p.w[(size_t)(cs + (cs < p.w_split ? 0 : p.w_gap)) * 9 + tap] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int tyy = tap / 3, txx = tap - tyy * 3;
            const float* pp = lds + ((row + tyy) * PW + col + txx) * CS;
            const float* wp = lw + tap * kch;
            for (int c = 0; c < kch; c += 4) {
                const float4 a = *reinterpret_cast<const float4*>(pp + c);
                const float4 w4 = *reinterpret_cast<const float4*>(wp + c);
                acc += a.x * w4.x + a.y * w4.y + a.z * w4.z + a.w * w4.w;
            }
        }
    }
    const int oy = oy0 + row, ox = ox0 + col;
    if (oy < p.Hin && ox < p.Win) {
        const size_t opx = ((size_t)b * p.Hin + oy) * p.Win + ox;
        if (p.add) acc += p.add[opx * p.add_stride + p.add_off];
        float* o = p.out + opx * p.out_stride + p.out_off;
        if (p.fill4) *reinterpret_cast<float4*>(o) = make_float4(acc, 0.f, 0.f, 0.f);
        else *o = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Both growth-1 layers of a coupling network in one launch (denseBlock.py:135-152 with growth 1, two layers):
//   d1 = conv3x3_zero(relu(t0); W1) + add1,   d2 = conv3x3_zero(relu(cat(t0, d1)); W2) + add2,   D = (d1, d2, 0, 0)
// The input patch is staged once with halo 2; d1 is evaluated on the tile plus a one-pixel ring (two pixels per thread,
// zero outside the image: the second conv zero-pads) into LDS, then d2 on the tile.  Halves the launches and the reads of
// t0 against two c1_fwd_kernel launches.
// ---------------------------------------------------------------------------------------------
struct C1X2P {
    TmgSeg in[TMG_MAX_IN_SEG];
    int nseg, vec4;
    int B, Hin, Win;
    int Cin;
    const float* in_scale;
    const float* in_shift;
    int relu_in, pad_rep;
    const float* w1;  // [rows][9]: input channel c < w_rows reads row c (+ w_gap if c >= w_split)
    const float* w2;  // same mapping for the t0 channels; the d1 channel reads row w2_d1
    int w_rows, w_split, w_gap, w2_d1;
    const float* add1; int a1_stride, a1_off;
    const float* add2; int a2_stride, a2_off;
    float* out; int out_stride;  // float4 (d1, d2, 0, 0) per pixel
    int TW_log2, tiles_x, tiles_y, KCH;
    int xmap;                    // XCD-aware tile order (tmg_common.h)
};

// CG: threads per pixel.  The kernel is a chain of LDS-read latencies with one pixel per thread, so a 256-pixel tile per block leaves the
// small levels on a fraction of the chip (64 tiles at 16 x 16 x 64 samples) with one wave per SIMD and nothing to hide the chain behind;
// with CG = 2 / 4 adjacent lanes splitting a pixel's channel quads, a tile is 128 / 64 pixels: 2 - 4 x the blocks and a 2 - 4 x shorter
// chain per thread, the partial sums meeting in two xor-shuffles.
template <int CG>
__global__ __launch_bounds__(256) void c1x2_fwd_kernel(C1X2P p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NPX = 256 / CG;
    const int tid = threadIdx.x;
    const int pid = tid / CG, cg = tid % CG;   // the CG lanes of a pixel are adjacent lanes of one wave
    const bool lead = cg == 0;
    int t = tmg_xcd_block((int)blockIdx.x, (int)gridDim.x, p.xmap);
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int TWl = p.TW_log2, TW = 1 << TWl, TH = NPX >> TWl;
    const int PW = TW + 4, PH = TH + 4;   // staged patch (halo 2)
    const int RW = TW + 2, RH = TH + 2;   // d1 region (halo 1)
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int row = pid >> TWl, col = pid & (TW - 1);
    const int Cpad = (p.Cin + 3) & ~3;
    // this pixel's d1-region pixels: [0] its OWN pixel (region coordinates (row + 1, col + 1): the conv-1 and conv-2 sums then read the
    // same patch values - the kernel is bound by LDS reads, a third of them were this duplicate), [1] one pixel of the one-pixel ring
    // around the tile for the first 2 (RW + RH) - 4 pixels' threads (top row, bottom row, left column, right column)
    int ry[2], rx[2];
    bool rok[2];
    ry[0] = row + 1; rx[0] = col + 1; rok[0] = true;
    const int nring = 2 * RW + 2 * (RH - 2);   // <= NPX (the launcher's tile shapes)
    {
        rok[1] = pid < nring;
        int i = rok[1] ? pid : 0;
        if (i < RW) { ry[1] = 0; rx[1] = i; }
        else if (i < 2 * RW) { ry[1] = RH - 1; rx[1] = i - RW; }
        else if (i < 2 * RW + RH - 2) { ry[1] = 1 + i - 2 * RW; rx[1] = 0; }
        else { ry[1] = 1 + i - (2 * RW + RH - 2); rx[1] = RW - 1; }
    }
    const bool ring_wave = (tid & ~63) / CG < nring;   // wave-uniform: does any lane of this wave own a ring pixel
    // all global operands of the epilogue up front (latency hides under the staging + compute)
    const int oy = oy0 + row, ox = ox0 + col;
    const bool own = oy < p.Hin && ox < p.Win;
    const size_t opx = ((size_t)b * p.Hin + min(oy, p.Hin - 1)) * p.Win + min(ox, p.Win - 1);
    const float a2v = *((p.add2 && own && lead) ? p.add2 + opx * p.a2_stride + p.a2_off : tmg_zero_page);
    float a1v[2];
    bool rin[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int y = oy0 - 1 + ry[u], x = ox0 - 1 + rx[u];
        rin[u] = rok[u] && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
        const size_t px = ((size_t)b * p.Hin + min(max(y, 0), p.Hin - 1)) * p.Win + min(max(x, 0), p.Win - 1);
        a1v[u] = *((p.add1 && rin[u] && lead) ? p.add1 + px * p.a1_stride + p.a1_off : tmg_zero_page);
    }
    float acc1[2] = {0.f, 0.f}, acc2 = 0.f;
    for (int c0 = 0; c0 < Cpad; c0 += p.KCH) {
        const int kch = min(p.KCH, Cpad - c0);
        const int CS = kch + 4;
        float* lw1 = lds + PH * PW * CS;  // [9][kch]
        float* lw2 = lw1 + 9 * p.KCH;     // [9][kch]
        __syncthreads();
        stage_patch_bf(p, lds, b, oy0 - 2, ox0 - 2, PH, PW, c0, kch, CS);
        for (int i = tid; i < 9 * kch; i += 256) {
            const int tap = i / kch, c = i - tap * kch;
            const int cs = c0 + c;
            const bool ok = cs < p.w_rows;
            const size_t r = (size_t)(cs + (cs < p.w_split ? 0 : p.w_gap)) * 9 + tap;
            lw1[i] = ok ? p.w1[r] : 0.f;
            lw2[i] = ok ? p.w2[r] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int tyy = tap / 3, txx = tap - tyy * 3;
            const float* w1p = lw1 + tap * kch;
            const float* w2p = lw2 + tap * kch;
            const float* p0 = lds + ((ry[0] + tyy) * PW + rx[0] + txx) * CS;          // own pixel + tap (region pixel + tap; patch = region - 1)
            const float* p1 = lds + ((ry[1] + tyy) * PW + rx[1] + txx) * CS;          // ring pixel + tap
            for (int c = 4 * cg; c < kch; c += 4 * CG) {
                const float4 wa = *reinterpret_cast<const float4*>(w1p + c);
                const float4 wb = *reinterpret_cast<const float4*>(w2p + c);
                const float4 x0 = *reinterpret_cast<const float4*>(p0 + c);
                acc1[0] += x0.x * wa.x + x0.y * wa.y + x0.z * wa.z + x0.w * wa.w;
                acc2 += x0.x * wb.x + x0.y * wb.y + x0.z * wb.z + x0.w * wb.w;
                if (ring_wave) {
                    const float4 x1 = *reinterpret_cast<const float4*>(p1 + c);
                    acc1[1] += x1.x * wa.x + x1.y * wa.y + x1.z * wa.z + x1.w * wa.w;
                }
            }
        }
    }
    // the CG partial sums of a pixel meet in its lanes (every lane ends up with the total)
#pragma unroll
    for (int o = 1; o < CG; o <<= 1) {
        acc1[0] += __shfl_xor(acc1[0], o);
        acc1[1] += __shfl_xor(acc1[1], o);
        acc2 += __shfl_xor(acc2, o);
    }
    // d1 on the ring-extended region: raw value at this thread's own pixel is needed for the output, relu(d1) for conv 2
    __syncthreads();
    float* ld1 = lds;  // [RH*RW] relu(d1), zero outside the image
    if (lead) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (rok[u]) ld1[ry[u] * RW + rx[u]] = rin[u] ? fmaxf(acc1[u] + a1v[u], 0.f) : 0.f;
    }
    const float d1own = acc1[0] + a1v[0];   // raw d1 of the own pixel
    __syncthreads();
    if (!lead) return;
    float w2d[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) w2d[tap] = p.w2[(size_t)p.w2_d1 * 9 + tap];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int tyy = tap / 3, txx = tap - tyy * 3;
        acc2 += w2d[tap] * ld1[(row + tyy) * RW + col + txx];   // region index of own pixel + tap - 1 = (row + tyy, col + txx)
    }
    if (own) *reinterpret_cast<float4*>(p.out + opx * p.out_stride) = make_float4(d1own, acc2 + a2v, 0.f, 0.f);
}

// Backward of the C_out = 1 layer.  ddm(p) = dd(p) * [dref(p) > 0]  (dref null -> no mask)
//   G[p][ci]  += sum_tap w[ci][tap] * ddm(p - tap + 1)          (raw gradient w.r.t. the ReLU'd input)
//   dW[ci][tap] += sum_p relu(in(p + tap - 1))[ci] * ddm(p)
struct C1BP {
    TmgSeg in[TMG_MAX_IN_SEG];
    int nseg, vec4;
    int B, Hin, Win;
    int Cin;
    const float* in_scale;
    const float* in_shift;
    int relu_in, pad_rep;
    const float* w;   // [Cin][9]
    float* dW;        // [Cin][9] atomically accumulated
    const float* dd; int dd_stride, dd_off;
    const float* dref; int dref_stride, dref_off;
    TmgOSeg g[TMG_MAX_OUT_SEG];  // gradient destinations for input channels (accumulated, +=)
    int TW_log2, tiles_x, tiles_y, ntiles, KCH;
};

__global__ __launch_bounds__(256) void c1_bwd_kernel(C1BP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int TWl = p.TW_log2, TW = 1 << TWl, TH = 256 >> TWl;
    const int PW = TW + 2, PH = TH + 2;
    const int row = tid >> TWl, col = tid & (TW - 1);
    const int Cpad = (p.Cin + 3) & ~3;
    const int nchunks = (Cpad + p.KCH - 1) / p.KCH;
    float* ldd = lds;                       // [PH*PW] masked dd with halo 1 (zero outside the image)
    float* lw = ldd + ((PH * PW + 3) & ~3); // [9][KCH]
    float* lin = lw + 9 * p.KCH;            // [PH*PW][CS]
    // weight-gradient accumulators: this thread owns outputs o = tid + 256*k of the chunk's kch*9 products
    float wacc[2][8];  // [k][chunk]  (kch*9 <= 512 -> KCH <= 56 ; nchunks <= 8)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int c = 0; c < 8; ++c) wacc[k][c] = 0.f;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const int b = t / p.tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        __syncthreads();
        for (int i = tid; i < PH * PW; i += 256) {
            const int py = i / PW, px = i - py * PW;
            const int y = oy0 - 1 + py, x = ox0 - 1 + px;
            float v = 0.f;
            if (y >= 0 && y < p.Hin && x >= 0 && x < p.Win) {
                const size_t pix = ((size_t)b * p.Hin + y) * p.Win + x;
                v = p.dd[pix * p.dd_stride + p.dd_off];
                if (p.dref && !(p.dref[pix * p.dref_stride + p.dref_off] > 0.f)) v = 0.f;
            }
            ldd[i] = v;
        }
        __syncthreads();
        float dnb[9];  // ddm(p - tap + 1): tap (ky,kx) reads patch (row + 2 - ky, col + 2 - kx)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            dnb[tap] = ldd[(row + 2 - ky) * PW + col + 2 - kx];
        }
        const int oy = oy0 + row, ox = ox0 + col;
        const bool inside = oy < p.Hin && ox < p.Win;
        const size_t opix = ((size_t)b * p.Hin + oy) * p.Win + ox;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            if (ch >= nchunks) break;
            const int c0 = ch * p.KCH;
            const int kch = min(p.KCH, Cpad - c0);
            const int CS = kch + 4;
            __syncthreads();
            stage_patch_bf(p, lin, b, oy0 - 1, ox0 - 1, PH, PW, c0, kch, CS);
            for (int i = tid; i < 9 * kch; i += 256) {
                const int tap = i / kch, c = i - tap * kch;
                lw[i] = (c0 + c < p.Cin) ? p.w[(size_t)(c0 + c) * 9 + tap] : 0.f;
            }
            __syncthreads();
            // (a) input gradient for this pixel, channels of the chunk
            if (inside) {
                for (int c = 0; c < kch; ++c) {
                    const int ci = c0 + c;
                    if (ci >= p.Cin) break;
                    float v = 0.f;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) v += lw[tap * kch + c] * dnb[tap];
                    int nl = ci;
                    TMG_PICK_OSEG(p.g, nl, gptr, gstride, goff)
                    float* d = gptr + opix * gstride + goff + nl;
                    *d += v;
                }
            }
            // (b) weight gradient: outputs (c, tap) of the chunk spread over the threads
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int o = tid + 256 * k;
                if (o < kch * 9) {
                    const int tap = o / kch, c = o - tap * kch;
                    const int ky = tap / 3, kx = tap - ky * 3;
                    float sacc = 0.f;
                    for (int r = 0; r < TH; ++r) {
                        const float* ip = lin + ((r + ky) * PW + kx) * CS + c;
                        const float* dp = ldd + (r + 1) * PW + 1;
                        for (int cc = 0; cc < TW; ++cc) sacc += ip[cc * CS] * dp[cc];
                    }
                    wacc[k][ch] += sacc;
                }
            }
        }
    }
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        if (ch >= nchunks) break;
        const int c0 = ch * p.KCH;
        const int kch = min(p.KCH, Cpad - c0);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int o = tid + 256 * k;
            if (o < kch * 9) {
                const int tap = o / kch, c = o - tap * kch;
                if (c0 + c < p.Cin) atomicAdd(p.dW + (size_t)(c0 + c) * 9 + tap, wacc[k][ch]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused backward of the two growth-1 dense layers (one pass over the coupling-network input):
//   dd2m = GD[1] * [d2 > 0]
//   dd1m = [d1 > 0] * (GD[0] + sum_tap W2[cin_nn][tap] * dd2m(p - tap + 1))
//   grad(t0)[c] = [t0[c] > 0] * (G0[c] + sum_tap W2[c][tap] dd2m(p-tap+1) + W1[c][tap] dd1m(p-tap+1)) (+ add0)
//   dW1[c][tap] += sum_p relu(t0)(p+tap-1)[c] dd1m(p);   dW2[c][tap] += sum_p relu(t1)(p+tap-1)[c] dd2m(p)
// G0 = raw zero-conv input gradient (MFMA kernel output), GD its 4-channel part for D = (d1, d2, 0, 0).
// Input segments = nn inputs followed by D (so Cin counts D's 4 channels; W1/W2/dW1/dW2 are [Cin][9], zero-padded).
// ---------------------------------------------------------------------------------------------
struct D2BP {
    TmgSeg in[TMG_MAX_IN_SEG];
    int nseg, vec4;
    int B, Hin, Win;
    int Cin;
    const float* in_scale;
    const float* in_shift;
    int relu_in, pad_rep;
    const float* w1;
    const float* w2;
    float* dW1;
    float* dW2;
    const float* GD; int gd_stride;
    const float* Dp; int d_stride;
    int cin_nn, rows1, rows2;  // input channel c uses w1 row c if c < rows1 and w2 row c (+ gap2 if c >= split2) if c < rows2
    int split2, gap2;
    TmgSeg g0[2];
    TmgOSeg out[2];
    const float* add0; int add0_stride;
    float* dd1_out; float* dd2_out; int dd_stride;  // optional: masked gradients w.r.t. d1 / d2 per pixel (null: not written)
    int dd_quad;                                    // the two are channels 0, 1 of a float4 slot whose channels 2, 3 are to be zero: one 16-byte store
    int TW_log2, tiles_x, tiles_y, ntiles, KCH;
    int xmap;                                       // XCD-aware tile order (tmg_common.h)
};

// WG: with the weight gradients of both layers (false: the caller computes them with the MFMA weight-gradient kernel)
template <bool WG>
__global__ __launch_bounds__(256) void dense2_bwd_kernel(D2BP p) {
    // grid = (pixel-tile shares, channel chunks): a block owns ONE chunk of <= KCH input channels (KCH*9 <= 256, one
    // weight-gradient output per thread) and walks its share of the 256-pixel tiles.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int TWl = p.TW_log2, TW = 1 << TWl, TH = 256 >> TWl;
    const int PW = TW + 2, PH = TH + 2, QW = TW + 4, QH = TH + 4;
    const int row = tid >> TWl, col = tid & (TW - 1);
    const int Cpad = (p.Cin + 3) & ~3;
    const int c0 = blockIdx.y * p.KCH;
    const int kch = min(p.KCH, Cpad - c0);
    const int CS = kch + 8;
    float* A2 = lds;                              // [QH*QW] dd2m, halo 2
    float* A1 = A2 + ((QH * QW + 3) & ~3);        // [PH*PW] dd1m, halo 1
    float* lw1 = A1 + ((PH * PW + 3) & ~3);       // [9][KCH]
    float* lw2 = lw1 + 9 * p.KCH;                 // [9][KCH]
    float* lin = lw2 + 9 * p.KCH;                 // [PH*PW][CS]
    float wa1 = 0.f, wa2 = 0.f;
    float w2d[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w2d[t] = p.w2[(size_t)(p.cin_nn + (p.cin_nn < p.split2 ? 0 : p.gap2)) * 9 + t];
    for (int i = tid; i < 9 * kch; i += 256) {
        const int tap = i / kch, c = i - tap * kch;
        lw1[i] = (c0 + c < p.rows1) ? p.w1[(size_t)(c0 + c) * 9 + tap] : 0.f;
        lw2[i] = (c0 + c < p.rows2) ? p.w2[(size_t)(c0 + c + (c0 + c < p.split2 ? 0 : p.gap2)) * 9 + tap] : 0.f;
    }
    const int wo_tap = tid / kch, wo_c = tid - wo_tap * kch;   // this thread's weight-gradient output
    const bool wo_ok = WG && tid < kch * 9;
    const int wo_ky = wo_tap / 3, wo_kx = wo_tap - wo_ky * 3;

    const TmgTileRange tr_ = tmg_xcd_tiles(p.ntiles, (int)blockIdx.x, (int)gridDim.x, p.xmap);
    for (int tile = tr_.first; tile < tr_.end; tile += tr_.step) {
        int t = tile;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const int b = t / p.tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        __syncthreads();
        // All global loads of the tile are issued up front and without control flow (out-of-image lanes read the zero
        // page): the kernel is latency-bound (a handful of blocks per CU, ~6 dependent phases per tile before), so
        // one exposed memory latency per tile instead of five is the whole game.
        const int oy = oy0 + row, ox = ox0 + col;
        const bool own = oy < p.Hin && ox < p.Win;
        const size_t opix = ((size_t)b * p.Hin + min(oy, p.Hin - 1)) * p.Win + min(ox, p.Win - 1);
        float a2g[2], a2d[2], a1g[2], a1d[2];
        bool a2in[2], a1in[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 256;
            {
                const int py = i / QW, px = i - py * QW;
                const int y = oy0 - 2 + py, x = ox0 - 2 + px;
                a2in[u] = i < QH * QW && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
                const size_t pix = ((size_t)b * p.Hin + min(max(y, 0), p.Hin - 1)) * p.Win + min(max(x, 0), p.Win - 1);
                a2g[u] = *(a2in[u] ? p.GD + pix * p.gd_stride + 1 : tmg_zero_page);
                a2d[u] = *(a2in[u] ? p.Dp + pix * p.d_stride + 1 : tmg_zero_page);
            }
            {
                const int py = i / PW, px = i - py * PW;
                const int y = oy0 - 1 + py, x = ox0 - 1 + px;
                a1in[u] = i < PH * PW && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
                const size_t pix = ((size_t)b * p.Hin + min(max(y, 0), p.Hin - 1)) * p.Win + min(max(x, 0), p.Win - 1);
                a1g[u] = *(a1in[u] ? p.GD + pix * p.gd_stride : tmg_zero_page);
                a1d[u] = *(a1in[u] ? p.Dp + pix * p.d_stride : tmg_zero_page);
            }
        }
        // gradient / add operands of this thread's own pixel for the first PQ channel quads of the chunk
        constexpr int PQ = 4;
        float4 gq[PQ], aq[PQ];
        const bool pre_ok = p.vec4 && own && c0 < p.cin_nn;
#pragma unroll
        for (int j = 0; j < PQ; ++j) {
            const int ci = c0 + 4 * j;
            int nl = ci;
            const int sgi = (nl >= p.g0[0].n) ? 1 : 0;
            if (sgi) nl -= p.g0[0].n;
            const bool ok = pre_ok && 4 * j < kch && ci < p.cin_nn;
            const float* gp = (sgi ? p.g0[1].p : p.g0[0].p) + opix * (sgi ? p.g0[1].stride : p.g0[0].stride) + (sgi ? p.g0[1].off : p.g0[0].off) + nl;
            gq[j] = *reinterpret_cast<const float4*>(ok ? gp : tmg_zero_page);
            const bool aok = ok && p.add0 && !sgi;
            aq[j] = *reinterpret_cast<const float4*>(aok ? p.add0 + opix * p.add0_stride + nl : tmg_zero_page);
        }
        stage_patch_bf(p, lin, b, oy0 - 1, ox0 - 1, PH, PW, c0, kch, CS);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 256;
            if (i < QH * QW) A2[i] = (a2in[u] && a2d[u] > 0.f) ? a2g[u] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 256;
            if (i < PH * PW) {
                const int py = i / PW, px = i - py * PW;
                float v = 0.f;
                if (a1in[u] && a1d[u] > 0.f) {
                    v = a1g[u];
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int ky = tap / 3, kx = tap - ky * 3;
                        v += w2d[tap] * A2[(py + 2 - ky) * QW + px + 2 - kx];
                    }
                }
                A1[i] = v;
            }
        }
        __syncthreads();
        if (p.dd1_out && blockIdx.y == 0 && own) {
            const size_t px_ = ((size_t)b * p.Hin + oy) * p.Win + ox;
            const float d1v = A1[(row + 1) * PW + col + 1], d2v = A2[(row + 2) * QW + col + 2];
            if (p.dd_quad) {
                *reinterpret_cast<float4*>(p.dd1_out + px_ * p.dd_stride) = make_float4(d1v, d2v, 0.f, 0.f);   // no zero fill of the stash needed
            } else if (p.dd2_out == p.dd1_out + 1 && !(p.dd_stride & 1) && !(((uintptr_t)p.dd1_out) & 7)) {
                *reinterpret_cast<float2*>(p.dd1_out + px_ * p.dd_stride) = make_float2(d1v, d2v);  // adjacent channels: one store
            } else {
                p.dd1_out[px_ * p.dd_stride] = d1v;
                p.dd2_out[px_ * p.dd_stride] = d2v;
            }
        }
        if (own && c0 < p.cin_nn) {
            float n1[9], n2[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                n1[tap] = A1[(row + 2 - ky) * PW + col + 2 - kx];
                n2[tap] = A2[(row + 3 - ky) * QW + col + 3 - kx];
            }
            const float* cen = lin + ((row + 1) * PW + col + 1) * CS;
            // one channel quad: input gradient of both layers, ReLU mask, upstream gradient and optional add operand
            // (G_/AD_: prefetched registers when PRE_, else loaded here)
#define TMG_D2_QUAD(C_, G_, AD_, PRE_)                                                                                    \
            {                                                                                                             \
                const int c = (C_), ci = c0 + c;                                                                          \
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                               \
                _Pragma("unroll") for (int tap = 0; tap < 9; ++tap) {                                                     \
                    const float4 a = *reinterpret_cast<const float4*>(lw1 + tap * kch + c);                               \
                    const float4 bq = *reinterpret_cast<const float4*>(lw2 + tap * kch + c);                              \
                    v.x += a.x * n1[tap] + bq.x * n2[tap];                                                                \
                    v.y += a.y * n1[tap] + bq.y * n2[tap];                                                                \
                    v.z += a.z * n1[tap] + bq.z * n2[tap];                                                                \
                    v.w += a.w * n1[tap] + bq.w * n2[tap];                                                                \
                }                                                                                                         \
                const float4 m = *reinterpret_cast<const float4*>(cen + c);                                               \
                if (p.vec4) {                                                                                             \
                    int nl = ci;                                                                                          \
                    const int sgi = (nl >= p.g0[0].n) ? 1 : 0;                                                            \
                    if (sgi) nl -= p.g0[0].n;                                                                             \
                    float* op = (sgi ? p.out[1].p : p.out[0].p) + opix * (sgi ? p.out[1].stride : p.out[0].stride) + (sgi ? p.out[1].off : p.out[0].off) + nl; \
                    float4 g = G_, ad = AD_;                                                                              \
                    if (!(PRE_)) {                                                                                        \
                        g = *reinterpret_cast<const float4*>((sgi ? p.g0[1].p : p.g0[0].p) + opix * (sgi ? p.g0[1].stride : p.g0[0].stride) + (sgi ? p.g0[1].off : p.g0[0].off) + nl); \
                        ad = (p.add0 && !sgi) ? *reinterpret_cast<const float4*>(p.add0 + opix * p.add0_stride + nl) : make_float4(0.f, 0.f, 0.f, 0.f); \
                    }                                                                                                     \
                    float4 o;                                                                                             \
                    o.x = (m.x > 0.f ? g.x + v.x : 0.f) + ad.x;                                                           \
                    o.y = (m.y > 0.f ? g.y + v.y : 0.f) + ad.y;                                                           \
                    o.z = (m.z > 0.f ? g.z + v.z : 0.f) + ad.z;                                                           \
                    o.w = (m.w > 0.f ? g.w + v.w : 0.f) + ad.w;                                                           \
                    *reinterpret_cast<float4*>(op) = o;                                                                   \
                } else {                                                                                                  \
                    const float vv[4] = {v.x, v.y, v.z, v.w};                                                             \
                    const float mm[4] = {m.x, m.y, m.z, m.w};                                                             \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                       \
                        int nl = ci + e;                                                                                  \
                        if (nl < p.cin_nn) {                                                                              \
                            const int sgi = (nl >= p.g0[0].n) ? 1 : 0;                                                    \
                            if (sgi) nl -= p.g0[0].n;                                                                     \
                            const float g = (sgi ? p.g0[1].p : p.g0[0].p)[opix * (sgi ? p.g0[1].stride : p.g0[0].stride) + (sgi ? p.g0[1].off : p.g0[0].off) + nl]; \
                            float o = mm[e] > 0.f ? g + vv[e] : 0.f;                                                      \
                            if (p.add0 && !sgi) o += p.add0[opix * p.add0_stride + nl];                                   \
                            (sgi ? p.out[1].p : p.out[0].p)[opix * (sgi ? p.out[1].stride : p.out[0].stride) + (sgi ? p.out[1].off : p.out[0].off) + nl] = o; \
                        }                                                                                                 \
                    }                                                                                                     \
                }                                                                                                         \
            }
#pragma unroll
            for (int j = 0; j < PQ; ++j)
                if (4 * j < kch && c0 + 4 * j < p.cin_nn) TMG_D2_QUAD(4 * j, gq[j], aq[j], true)
            for (int cq = 4 * PQ; cq < kch && c0 + cq < p.cin_nn; cq += 4) TMG_D2_QUAD(cq, make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), false)
#undef TMG_D2_QUAD
        }
        // weight gradients of both layers from the same staged activations: one (channel, tap) output per thread
        if (WG && wo_ok) {
            float s1 = 0.f, s2 = 0.f;
            for (int r = 0; r < TH; ++r) {
                const float* ip = lin + ((r + wo_ky) * PW + wo_kx) * CS + wo_c;
                const float* d1p = A1 + (r + 1) * PW + 1;
                const float* d2p = A2 + (r + 2) * QW + 2;
#pragma unroll 4
                for (int cc = 0; cc < TW; ++cc) {
                    const float a = ip[cc * CS];
                    s1 += a * d1p[cc];
                    s2 += a * d2p[cc];
                }
            }
            wa1 += s1;
            wa2 += s2;
        }
    }
    if (WG && wo_ok) {
        if (c0 + wo_c < p.rows1) atomicAdd(p.dW1 + (size_t)(c0 + wo_c) * 9 + wo_tap, wa1);
        if (c0 + wo_c < p.rows2) atomicAdd(p.dW2 + (size_t)(c0 + wo_c + (c0 + wo_c < p.split2 ? 0 : p.gap2)) * 9 + wo_tap, wa2);
    }
}

// The same backward WITHOUT the weight gradients (the level node computes those with the grouped 4x4x1-MFMA kernel) for the shape the
// level node calls it with: inputs = (x1 [cin_nn channels], D), one G0 / output segment, everything float4-addressable.  No patch is
// staged: the only use of the inputs here is the ReLU mask of the thread's OWN pixel, read straight from x1; a block owns NQ channel
// quads (blockIdx.y) and prefetches mask, upstream gradient and add operand of its pixel for all of them before the two barriers.
// The general kernel above needs 193 VGPRs (two blocks per CU: SQ counters show its waves 52 % of their life in s_waitcnt, 30 %
// issuing); this one is built for four waves per SIMD.
// (TWL = log2 of the tile width as a template parameter: with the patch widths known at compile time every tap of the two stencils is
// an immediate offset from one LDS address per thread; with a run-time width the compiler hoists ~50 tap addresses out of the tile
// loop and spills.)
template <int NQ, int TWL>
__global__ __launch_bounds__(256, NQ <= 2 ? 4 : 3) void dense2_bwd_lean_kernel(D2BP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    constexpr int TWl = TWL, TW = 1 << TWl, TH = 256 >> TWl;
    constexpr int PW = TW + 2, PH = TH + 2, QW = TW + 4, QH = TH + 4;
    const int row = tid >> TWl, col = tid & (TW - 1);
    constexpr int KC = 4 * NQ;
    const int c0 = blockIdx.y * KC;
    float* A2 = lds;                              // [QH*QW] dd2m, halo 2
    float* A1 = A2 + ((QH * QW + 3) & ~3);        // [PH*PW] dd1m, halo 1
    float* lw1 = A1 + ((PH * PW + 3) & ~3);       // [9][KC]
    float* lw2 = lw1 + 9 * KC;                    // [9][KC]
    float w2d[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w2d[t] = p.w2[(size_t)(p.cin_nn + (p.cin_nn < p.split2 ? 0 : p.gap2)) * 9 + t];
    for (int i = tid; i < 9 * KC; i += 256) {
        const int tap = i / KC, c = c0 + (i - tap * KC);
        lw1[i] = (c < p.cin_nn && c < p.rows1) ? p.w1[(size_t)c * 9 + tap] : 0.f;
        lw2[i] = (c < p.cin_nn && c < p.rows2) ? p.w2[(size_t)(c + (c < p.split2 ? 0 : p.gap2)) * 9 + tap] : 0.f;
    }
    // Addresses are a block-uniform base (scalar registers) + a 32-bit BYTE offset per lane (the launcher checks that every tensor
    // stays below 4 GB): 64-bit per-lane pointers for the ~20 loads issued up front were what filled the register file.  Lanes with
    // nothing to read load a valid clamped address and drop the value.
    const char* xb = reinterpret_cast<const char*>(p.in[0].p + p.in[0].off + c0);
    const char* gb = reinterpret_cast<const char*>(p.g0[0].p + p.g0[0].off + c0);
    char* ob = reinterpret_cast<char*>(p.out[0].p + p.out[0].off + c0);
    const char* ab = reinterpret_cast<const char*>(p.add0 ? p.add0 + c0 : tmg_zero_page);
    const unsigned xs4 = 4u * (unsigned)p.in[0].stride, gs4 = 4u * (unsigned)p.g0[0].stride, os4 = 4u * (unsigned)p.out[0].stride;
    const unsigned as4 = p.add0 ? 4u * (unsigned)p.add0_stride : 0u, gds4 = 4u * (unsigned)p.gd_stride, ds4 = 4u * (unsigned)p.d_stride;
    const char* gdb = reinterpret_cast<const char*>(p.GD);
    const char* dpb = reinterpret_cast<const char*>(p.Dp);
    const int nq_here = min(NQ, (p.cin_nn - c0) >> 2);     // live channel quads of this block (block-uniform)
    const TmgTileRange tr_ = tmg_xcd_tiles(p.ntiles, (int)blockIdx.x, (int)gridDim.x, p.xmap);
    for (int tile = tr_.first; tile < tr_.end; tile += tr_.step) {
        int t = tile;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const int b = t / p.tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        // (an opaque copy of the thread index per tile: the staging items' patch coordinates i / QW, i / PW are tile-invariant, hoisted
        // out of the loop and - with four blocks per CU, 128 registers - one of them spilled: 8 bytes of scratch per lane)
        int tidl = tid;
        asm volatile("" : "+v"(tidl));
        __syncthreads();
        const int oy = oy0 + row, ox = ox0 + col;
        const bool own = oy < p.Hin && ox < p.Win;
        const unsigned opix = ((unsigned)b * (unsigned)p.Hin + (unsigned)min(oy, p.Hin - 1)) * (unsigned)p.Win + (unsigned)min(ox, p.Win - 1);
        // all global loads of the tile up front, branch-free (lanes with nothing to read load the zero page)
        float a2g[2], a2d[2], a1g[2], a1d[2];
        bool a2in[2], a1in[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tidl + u * 256;
            {
                const int py = i / QW, px = i - py * QW;
                const int y = oy0 - 2 + py, x = ox0 - 2 + px;
                a2in[u] = i < QH * QW && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
                const unsigned pix = ((unsigned)b * (unsigned)p.Hin + (unsigned)min(max(y, 0), p.Hin - 1)) * (unsigned)p.Win + (unsigned)min(max(x, 0), p.Win - 1);
                a2g[u] = *reinterpret_cast<const float*>(gdb + (pix * gds4 + 4u));
                a2d[u] = *reinterpret_cast<const float*>(dpb + (pix * ds4 + 4u));
            }
            {
                const int py = i / PW, px = i - py * PW;
                const int y = oy0 - 1 + py, x = ox0 - 1 + px;
                a1in[u] = i < PH * PW && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
                const unsigned pix = ((unsigned)b * (unsigned)p.Hin + (unsigned)min(max(y, 0), p.Hin - 1)) * (unsigned)p.Win + (unsigned)min(max(x, 0), p.Win - 1);
                a1g[u] = *reinterpret_cast<const float*>(gdb + pix * gds4);
                a1d[u] = *reinterpret_cast<const float*>(dpb + pix * ds4);
            }
        }
        float4 mq[NQ], gq[NQ], aq[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const unsigned cj = 16u * (unsigned)min(j, nq_here - 1);     // a quad past the end re-reads the last live one (never stored)
            mq[j] = *reinterpret_cast<const float4*>(xb + (opix * xs4 + cj));
            gq[j] = *reinterpret_cast<const float4*>(gb + (opix * gs4 + cj));
            aq[j] = *reinterpret_cast<const float4*>(ab + (opix * as4 + (p.add0 ? cj : 0u)));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tidl + u * 256;
            if (i < QH * QW) A2[i] = (a2in[u] && a2d[u] > 0.f) ? a2g[u] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tidl + u * 256;
            if (i < PH * PW) {
                const int py = i / PW, px = i - py * PW;
                float v = 0.f;
                if (a1in[u] && a1d[u] > 0.f) {
                    v = a1g[u];
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int ky = tap / 3, kx = tap - ky * 3;
                        v += w2d[tap] * A2[(py + 2 - ky) * QW + px + 2 - kx];
                    }
                }
                A1[i] = v;
            }
        }
        __syncthreads();
        if (p.dd1_out && blockIdx.y == 0 && own) {
            char* dq_ = reinterpret_cast<char*>(p.dd1_out) + opix * (4u * (unsigned)p.dd_stride);
            const float d1v_ = A1[(row + 1) * PW + col + 1], d2v_ = A2[(row + 2) * QW + col + 2];
            if (p.dd_quad) *reinterpret_cast<float4*>(dq_) = make_float4(d1v_, d2v_, 0.f, 0.f);   // no zero fill of the stash needed
            else *reinterpret_cast<float2*>(dq_) = make_float2(d1v_, d2v_);                         // compact stash: two channels per layer
        }
        if (own) {
            float n1[9], n2[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                n1[tap] = A1[(row + 2 - ky) * PW + col + 2 - kx];
                n2[tap] = A2[(row + 3 - ky) * QW + col + 3 - kx];
            }
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                if (j < nq_here) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const float4 a = *reinterpret_cast<const float4*>(lw1 + tap * KC + 4 * j);
                        const float4 bq = *reinterpret_cast<const float4*>(lw2 + tap * KC + 4 * j);
                        v.x += a.x * n1[tap] + bq.x * n2[tap];
                        v.y += a.y * n1[tap] + bq.y * n2[tap];
                        v.z += a.z * n1[tap] + bq.z * n2[tap];
                        v.w += a.w * n1[tap] + bq.w * n2[tap];
                    }
                    float4 o;
                    o.x = (mq[j].x > 0.f ? gq[j].x + v.x : 0.f) + aq[j].x;
                    o.y = (mq[j].y > 0.f ? gq[j].y + v.y : 0.f) + aq[j].y;
                    o.z = (mq[j].z > 0.f ? gq[j].z + v.z : 0.f) + aq[j].z;
                    o.w = (mq[j].w > 0.f ? gq[j].w + v.w : 0.f) + aq[j].w;
                    *reinterpret_cast<float4*>(ob + (opix * os4 + 16u * j)) = o;
                }
            }
        }
    }
}

// d(kappa) of a Conv2dZeros from its parameter gradients: h = e^k (W*x + b) is degree-1 homogeneous in (W, b), so
// sum dh*h = <W, dW> + <b, db>; zero when kappa sits outside the clamp range (flowUtils.py:247).
__global__ __launch_bounds__(256) void dkappa_kernel(const float* __restrict__ w, const float* __restrict__ dw, int nw,
                                                     const float* __restrict__ b, const float* __restrict__ db, int nb,
                                                     const float* __restrict__ kappa, float* __restrict__ dk) {
    __shared__ float red[4];
    float a = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nw; i += gridDim.x * 256) a += w[i] * dw[i];
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < nb; i += 256) a += b[i] * db[i];
    const float tot = block_sum_256(a, red);
    if (threadIdx.x == 0) {
        const float k = *kappa;
        if (k >= -4.0f && k <= 1.3862943611198906f) atomicAdd(dk, tot);   // dk is zero-filled by the caller
    }
}

extern "C" int tmg_affine_bwd_scaled(const void* gout, const int64_t* go_d, const void* yref, const int64_t* yr_d, const void* rsave,
                                     const void* g, void* gin, const int64_t* gi_d, void* dhh, const int64_t* dh_d,
                                     const void* kappa, const int64_t* dims, hipStream_t st);

extern "C" int tmg_c1_fwd_add(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* w, const void* add,
                              const int64_t* add_d, void* out, const int64_t* out_d, const int64_t* dims, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
// dims: [B, pix_per_img, Ch, reverse]; each *_d = [stride, off]
// As tmg_affine_apply; additionally copies the Ch pass-through channels x1 -> y1 (x1_d / y1_d = [pixel stride, 0]) in the
// same pass (the reference's cat(x1, x2') at flowAffine.py:83,109).  x1 == null: no copy.
extern "C" int tmg_affine_apply_pass(const void* hh, const int64_t* hh_d, const void* x2, const int64_t* x_d, void* y2,
                                     const int64_t* y_d, void* rsave, void* logdet, const void* x1, const int64_t* x1_d, void* y1,
                                     const int64_t* y1_d, const int64_t* dims, hipStream_t st) {
    const int B = (int)dims[0], ppi = (int)dims[1], Ch = (int)dims[2];
    int vec = ((Ch & 3) == 0) && (((hh_d[0] | hh_d[1] | x_d[0] | x_d[1] | y_d[0] | y_d[1]) & 3) == 0) &&
              (((((uintptr_t)hh) | ((uintptr_t)x2) | ((uintptr_t)y2) | ((uintptr_t)rsave)) & 15) == 0);
    if (x1 && (((x1_d[0] | y1_d[0]) & 3) || ((((uintptr_t)x1) | ((uintptr_t)y1)) & 15))) vec = 0;
    const size_t per = (size_t)ppi * (vec ? Ch / 4 : Ch);
    int gx = (int)((per + 1023) / 1024);  // >= 4 items per thread
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    TmgProf prof(TMG_PROF_AFF, 4.0 * B * (double)ppi * Ch * (x1 ? 7 : 5), st);
    hipLaunchKernelGGL(affine_apply_kernel, dim3(gx, B), dim3(256), 0, st, (const float*)hh, (int)hh_d[0], (int)hh_d[1],
                       (const float*)x2, (int)x_d[0], (int)x_d[1], (float*)y2, (int)y_d[0], (int)y_d[1], (float*)rsave, (float*)logdet,
                       ppi, Ch, (int)dims[3], vec, (const float*)x1, x1 ? (int)x1_d[0] : 0, x1 ? (float*)y1 : nullptr, x1 ? (int)y1_d[0] : 0);
    TMG_CHECK_LAUNCH();
    return 0;
}

extern "C" int tmg_affine_apply(const void* hh, const int64_t* hh_d, const void* x2, const int64_t* x_d, void* y2,
                                const int64_t* y_d, void* rsave, void* logdet, const int64_t* dims, hipStream_t st) {
    return tmg_affine_apply_pass(hh, hh_d, x2, x_d, y2, y_d, rsave, logdet, nullptr, nullptr, nullptr, nullptr, dims, st);
}

extern "C" int tmg_affine_bwd(const void* gout, const int64_t* go_d, const void* yref, const int64_t* yr_d, const void* rsave,
                              const void* g, void* gin, const int64_t* gi_d, void* dhh, const int64_t* dh_d, const int64_t* dims,
                              hipStream_t st) {
    return tmg_affine_bwd_scaled(gout, go_d, yref, yr_d, rsave, g, gin, gi_d, dhh, dh_d, nullptr, dims, st);
}

// As tmg_affine_bwd; dhh is multiplied by exp(clamp(*kappa)) (the Conv2dZeros output scale) when kappa is given.
extern "C" int tmg_affine_bwd_scaled(const void* gout, const int64_t* go_d, const void* yref, const int64_t* yr_d, const void* rsave,
                                     const void* g, void* gin, const int64_t* gi_d, void* dhh, const int64_t* dh_d,
                                     const void* kappa, const int64_t* dims, hipStream_t st) {
    const int B = (int)dims[0], ppi = (int)dims[1], Ch = (int)dims[2];
    const size_t npix = (size_t)B * ppi;
    TmgProf prof(TMG_PROF_AFFB, 4.0 * (double)npix * Ch * 6, st);   // reads gout, yref, r; writes gin, dhh (2 Ch)
    hipLaunchKernelGGL(affine_bwd_kernel, dim3(grid_for(npix * Ch)), dim3(256), 0, st, (const float*)gout, (int)go_d[0], (int)go_d[1],
                       (const float*)yref, (int)yr_d[0], (int)yr_d[1], (const float*)rsave, (const float*)g, (float*)gin, (int)gi_d[0],
                       (int)gi_d[1], (float*)dhh, (int)dh_d[0], (int)dh_d[1], ppi, Ch, npix, (int)dims[3], (const float*)kappa);
    TMG_CHECK_LAUNCH();
    return 0;
}

// dims: [npix, R]; cprev_d = [stride, off]
extern "C" int tmg_lstm_pointwise_fwd(void* gates, const void* c_prev, const int64_t* cprev_d, void* c_next, void* h_next,
                                      const int64_t* dims, hipStream_t st) {
    const size_t npix = (size_t)dims[0];
    const int R = (int)dims[1];
    TmgProf prof(TMG_PROF_LSTMF, 4.0 * (double)npix * R * (c_prev ? 7 : 6), st);   // gates 4R read, c_prev, c_next, h_next
    if (npix == 0 || R == 0) return 0;
    if ((R & 3) == 0 && (!c_prev || ((cprev_d[0] | cprev_d[1]) & 3) == 0) && npix * (size_t)(R / 4) < (1ull << 31) &&
        ((((uintptr_t)gates) | ((uintptr_t)c_prev) | ((uintptr_t)c_next) | ((uintptr_t)h_next)) & 15) == 0) {
        hipLaunchKernelGGL(lstm_pointwise_fwd4_kernel, dim3(grid_for(npix * (R / 4))), dim3(256), 0, st, (const float*)gates, (const float*)c_prev,
                           (int)cprev_d[0], (int)cprev_d[1], (float*)c_next, (float*)h_next, R / 4, (unsigned)npix);
        TMG_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(lstm_pointwise_fwd_kernel, dim3(grid_for(npix * R)), dim3(256), 0, st, (const float*)gates, (const float*)c_prev,
                       (int)cprev_d[0], (int)cprev_d[1], (float*)c_next, (float*)h_next, R, npix);
    TMG_CHECK_LAUNCH();
    return 0;
}

extern "C" int tmg_lstm_pointwise_bwd(void* acts, const void* c_prev, const int64_t* cprev_d, const void* c_next, const void* dh,
                                      const void* dc_in, void* dc_prev, const int64_t* dims, hipStream_t st) {
    const size_t npix = (size_t)dims[0];
    const int R = (int)dims[1];
    TmgProf prof(TMG_PROF_LSTMB, 4.0 * (double)npix * R * (8 + (c_prev ? 1 : 0) + 1 + (dh ? 1 : 0) + (dc_in ? 1 : 0) + (dc_prev ? 1 : 0)), st);
    if (npix == 0 || R == 0) return 0;
    if ((R & 3) == 0 && (!c_prev || ((cprev_d[0] | cprev_d[1]) & 3) == 0) && npix * (size_t)(R / 4) < (1ull << 31) &&
        ((((uintptr_t)acts) | ((uintptr_t)c_prev) | ((uintptr_t)c_next) | ((uintptr_t)dh) | ((uintptr_t)dc_in) | ((uintptr_t)dc_prev)) & 15) == 0) {
        hipLaunchKernelGGL(lstm_pointwise_bwd4_kernel, dim3(grid_for(npix * (R / 4))), dim3(256), 0, st, (float*)acts, (const float*)c_prev,
                           (int)cprev_d[0], (int)cprev_d[1], (const float*)c_next, (const float*)dh, (const float*)dc_in, (float*)dc_prev, R / 4,
                           (unsigned)npix);
        TMG_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(lstm_pointwise_bwd_kernel, dim3(grid_for(npix * R)), dim3(256), 0, st, (float*)acts, (const float*)c_prev,
                       (int)cprev_d[0], (int)cprev_d[1], (const float*)c_next, (const float*)dh, (const float*)dc_in, (float*)dc_prev, R,
                       npix);
    TMG_CHECK_LAUNCH();
    return 0;
}

// dims: [B, pix_per_img, Ch, mode, clip_mean]; fl: [mlo, mhi, slo, shi]
extern "C" int tmg_gauss_fwd(const void* hz, const int64_t* hz_d, const void* zin, const int64_t* zi_d, void* zout,
                             const int64_t* zo_d, void* logp, const int64_t* dims, const float* fl, hipStream_t st) {
    const int B = (int)dims[0], ppi = (int)dims[1], Ch = (int)dims[2];
    const size_t per = (size_t)ppi * Ch;
    if (per >= (1ull << 31)) return -2;
    // blocks per image: every block ends with ONE atomic on its image's log-prob, so few blocks per image (<= 32 per address) as long
    // as the grid still fills the chip (256 blocks per image at 64 images were 16 384 atomics on 64 addresses: 135 us for a 30-us map)
    int gx = (int)((per + 255) / 256);
    int cap = (2048 + B - 1) / B;
    if (cap < 4) cap = 4;
    if (cap > 256) cap = 256;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(gauss_fwd_kernel, dim3(gx, B), dim3(256), 0, st, (const float*)hz, (int)hz_d[0], (int)hz_d[1], (const float*)zin,
                       (int)zi_d[0], (int)zi_d[1], (float*)zout, (int)zo_d[0], (int)zo_d[1], (float*)logp, ppi, Ch, (int)dims[3],
                       (int)dims[4], fl[0], fl[1], fl[2], fl[3]);
    TMG_CHECK_LAUNCH();
    return 0;
}

extern "C" int tmg_gauss_bwd(const void* hz, const int64_t* hz_d, const void* zin, const int64_t* zi_d, const void* dzin,
                             const int64_t* dzi_d, const void* g, void* dzout, const int64_t* dzo_d, void* dhz, const int64_t* dh_d,
                             const int64_t* dims, const float* fl, hipStream_t st) {
    const int B = (int)dims[0], ppi = (int)dims[1], Ch = (int)dims[2];
    const size_t npix = (size_t)B * ppi;
    hipLaunchKernelGGL(gauss_bwd_kernel, dim3(grid_for(npix * Ch)), dim3(256), 0, st, (const float*)hz, (int)hz_d[0], (int)hz_d[1],
                       (const float*)zin, (int)zi_d[0], (int)zi_d[1], (const float*)dzin, (int)dzi_d[0], (int)dzi_d[1], (const float*)g,
                       (float*)dzout, (int)dzo_d[0], (int)dzo_d[1], (float*)dhz, (int)dh_d[0], (int)dh_d[1], ppi, Ch, npix, (int)dims[3],
                       (int)dims[4], fl[0], fl[1], fl[2], fl[3]);
    TMG_CHECK_LAUNCH();
    return 0;
}

// dims: [npix, ch, pad, to_padded]; s_d / d_d: {pixel stride, channel offset} of the source / destination
extern "C" int tmg_pad_halves(const void* src, const int64_t* s_d, void* dst, const int64_t* d_d, const int64_t* dims, hipStream_t st) {
    const size_t npix = (size_t)dims[0];
    const int ch = (int)dims[1], pad = (int)dims[2], to_padded = (int)dims[3];
    const size_t total = npix * (size_t)(to_padded ? 2 * (ch + pad) : 2 * ch);
    hipLaunchKernelGGL(pad_halves_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const float*)src + s_d[1], (int)s_d[0],
                       (float*)dst + d_d[1], (int)d_d[0], npix, ch, pad, to_padded);
    TMG_CHECK_LAUNCH();
    return 0;
}

// Host values -> device memory as KERNEL ARGUMENTS (by-value struct, 2 KB): the segment tables of the grouped launches hold raw
// activation pointers; a host-to-device copy cannot be recorded by a hipGraph capture (pageable source), a kernel node carries
// its arguments with it.
struct TmgI64x256 { long long v[256]; };
__global__ void fill_i64_kernel(long long* __restrict__ dst, TmgI64x256 t, int n) {
    const int i = threadIdx.x;
    if (i < n) dst[i] = t.v[i];
}

extern "C" int tmg_fill_i64(void* dst, const int64_t* vals, int64_t n, hipStream_t st) {
    if (n < 0 || (n && (!dst || !vals))) return -3;
    for (int64_t o = 0; o < n; o += 256) {
        TmgI64x256 t;
        const int m = (int)(n - o < 256 ? n - o : 256);
        for (int i = 0; i < 256; ++i) t.v[i] = i < m ? (long long)vals[o + i] : 0;
        hipLaunchKernelGGL(fill_i64_kernel, dim3(1), dim3(256), 0, st, (long long*)dst + o, t, m);
        TMG_CHECK_LAUNCH();
    }
    return 0;
}

// dims: [B, h, w, C, to_small]  (h, w: the SMALL spatial size; C: channels of the big tensor)
extern "C" int tmg_checker(const void* src, const int64_t* s_d, void* dst, const int64_t* d_d, const int64_t* dims, hipStream_t st) {
    const size_t total = (size_t)dims[0] * dims[1] * dims[2] * 4 * dims[3];
    if (total == 0) return 0;
    if ((dims[3] & 3) == 0 && ((s_d[0] | s_d[1] | d_d[0] | d_d[1]) & 3) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0 &&
        total / 4 < (1ull << 31)) {
        hipLaunchKernelGGL(checker4_kernel, dim3(grid_for(total / 4)), dim3(256), 0, st, (const float*)src, (int)s_d[0], (int)s_d[1], (float*)dst,
                           (int)d_d[0], (int)d_d[1], (int)dims[0], (int)dims[1], (int)dims[2], (int)(dims[3] / 4), (int)dims[4]);
        TMG_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(checker_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const float*)src, (int)s_d[0], (int)s_d[1], (float*)dst,
                       (int)d_d[0], (int)d_d[1], (int)dims[0], (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4]);
    TMG_CHECK_LAUNCH();
    return 0;
}

// dims: [B, hi, wi, ho, wo, C]; dense NHWC tensors
extern "C" int tmg_upsample_fwd(const void* src, void* dst, const int64_t* dims, hipStream_t st) {
    const size_t total = (size_t)dims[0] * dims[3] * dims[4] * dims[5];
    if (dims[5] % 4 == 0 && total / 4 < (1ull << 31) && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
        hipLaunchKernelGGL(upsample_fwd4_kernel, dim3(grid_for(total / 4)), dim3(256), 0, st, (const float*)src, (float*)dst, (int)dims[0],
                           (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4], (int)(dims[5] / 4));
        TMG_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const float*)src, (float*)dst, (int)dims[0],
                       (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4], (int)dims[5]);
    TMG_CHECK_LAUNCH();
    return 0;
}

extern "C" int tmg_upsample_bwd(const void* dout, void* din, const int64_t* dims, hipStream_t st) {
    const size_t total = (size_t)dims[0] * dims[1] * dims[2] * dims[5];
    // window of 8 candidate outputs per axis: enough while an input step spans <= 2.5 output steps (integer up-scaling by 2: 2.02)
    const bool win_ok = (dims[3] - 1) * 2 <= (dims[1] - 1) * 5 + 2 && (dims[4] - 1) * 2 <= (dims[2] - 1) * 5 + 2 && dims[1] > 1 && dims[2] > 1;
    if (dims[5] % 4 == 0 && win_ok && (size_t)dims[0] * dims[3] * dims[4] * dims[5] / 4 < (1ull << 31) && ((((uintptr_t)dout) | ((uintptr_t)din)) & 15) == 0) {
        hipLaunchKernelGGL(upsample_bwd4_kernel, dim3(grid_for(total / 4)), dim3(256), 0, st, (const float*)dout, (float*)din, (int)dims[0],
                           (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4], (int)(dims[5] / 4));
        TMG_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const float*)dout, (float*)din, (int)dims[0],
                       (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4], (int)dims[5]);
    TMG_CHECK_LAUNCH();
    return 0;
}

// dims: [npix, C, mode, divisor]; s0/s1 accumulate (caller zeroes); C <= 256
extern "C" int tmg_chan_reduce(const void* x, const int64_t* x_d, const void* g, const int64_t* g_d, const void* v0, const void* v1,
                               const void* v2, const void* v3, void* s0, void* s1, const int64_t* dims, hipStream_t st) {
    const size_t npix = (size_t)dims[0];
    const int C = (int)dims[1];
    if (C > 256 || C < 1) return -2;
    const int lanes = 256 / C;
    size_t blocks = (npix + lanes - 1) / lanes;
    blocks = (blocks + 31) / 32;  // >= 32 pixels per lane
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(chan_reduce_kernel, dim3((int)blocks), dim3(256), 0, st, (const float*)x, (int)x_d[0], (int)x_d[1],
                       (const float*)g, g ? (int)g_d[0] : 0, g ? (int)g_d[1] : 0, (const float*)v0, (const float*)v1, (const float*)v2,
                       (const float*)v3, (float*)s0, (float*)s1, npix, C, (int)dims[2], dims[3] > 0 ? 1.0f / (float)dims[3] : 1.0f);
    TMG_CHECK_LAUNCH();
    return 0;
}

// dims: [npix, C, accumulate]
extern "C" int tmg_bn_bwd_apply(const void* x, const int64_t* x_d, const void* g, const int64_t* g_d, const void* a, const void* bsh,
                                const void* mean, const void* rstd, const void* gamma, const void* m0, const void* m1, void* dx,
                                const int64_t* dx_d, const int64_t* dims, hipStream_t st) {
    const size_t npix = (size_t)dims[0];
    const int C = (int)dims[1];
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(npix * C)), dim3(256), 0, st, (const float*)x, (int)x_d[0], (int)x_d[1],
                       (const float*)g, (int)g_d[0], (int)g_d[1], (const float*)a, (const float*)bsh, (const float*)mean,
                       (const float*)rstd, (const float*)gamma, (const float*)m0, (const float*)m1, (float*)dx, (int)dx_d[0],
                       (int)dx_d[1], npix, C, (int)dims[2], dims[3] > 0 ? 1.0f / (float)dims[3] : 1.0f);
    TMG_CHECK_LAUNCH();
    return 0;
}

// One-pass BatchNorm moments (see chan_moments_kernel).  dims = {npix, C}; acc: double [2][C], zeroed by the caller; C <= 256.
extern "C" int tmg_chan_moments(const void* x, const int64_t* x_d, void* acc, const int64_t* dims, hipStream_t st) {
    const size_t npix = (size_t)dims[0];
    const int C = (int)dims[1];
    if (C > 256 || C < 1 || (((uintptr_t)acc) & 7)) return -2;
    const int lanes = 256 / C;
    size_t blocks = (npix + lanes - 1) / lanes;
    blocks = (blocks + 31) / 32;  // >= 32 pixels per lane
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(chan_moments_kernel, dim3((int)blocks), dim3(256), 0, st, (const float*)x, (int)x_d[0], (int)x_d[1], (double*)acc, npix, C);
    TMG_CHECK_LAUNCH();
    return 0;
}

// dims: [C, n, address of the module's int64 num_batches_tracked counter (0: none; incremented by one)]; fl: {eps, momentum};
// acc = the fp64 moments of tmg_chan_moments; otherwise as tmg_bn_finalize
extern "C" int tmg_bn_finalize64(const void* acc, const void* gamma, const void* beta, void* rmean, void* rvar, void* out, const int64_t* dims,
                                 const float* fl, hipStream_t st) {
    const int C = (int)dims[0];
    hipLaunchKernelGGL(bn_finalize64_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const double*)acc, (const float*)gamma, (const float*)beta,
                       (float*)rmean, (float*)rvar, (float*)out, C, (double)dims[1], fl[0], fl[1], reinterpret_cast<long long*>(dims[2]));
    TMG_CHECK_LAUNCH();
    return 0;
}

// dims: [C, n]; fl: {eps, momentum}; rmean / rvar may be null; out: [5][C] = mean, var, rstd, a, bsh (see bn_finalize_kernel)
extern "C" int tmg_bn_finalize(const void* sum, const void* csq, const void* gamma, const void* beta, void* rmean, void* rvar, void* out,
                               const int64_t* dims, const float* fl, hipStream_t st) {
    const int C = (int)dims[0];
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const float*)sum, (const float*)csq, (const float*)gamma,
                       (const float*)beta, (float*)rmean, (float*)rvar, (float*)out, C, (float)dims[1], fl[0], fl[1]);
    TMG_CHECK_LAUNCH();
    return 0;
}

// dims: [npix, n, accumulate]; src/ref/add may be null
extern "C" int tmg_masked_add(const void* src, const int64_t* s_d, const void* ref, const int64_t* r_d, const void* add,
                              const int64_t* a_d, void* dst, const int64_t* d_d, const int64_t* dims, hipStream_t st) {
    const size_t npix = (size_t)dims[0];
    const int n = (int)dims[1];
    if (npix == 0 || n == 0) return 0;
    {
        long long m = n | d_d[0] | d_d[1];
        uintptr_t al = (uintptr_t)dst;
        if (src) { m |= s_d[0] | s_d[1]; al |= (uintptr_t)src; }
        if (ref) { m |= r_d[0] | r_d[1]; al |= (uintptr_t)ref; }
        if (add) { m |= a_d[0] | a_d[1]; al |= (uintptr_t)add; }
        if ((m & 3) == 0 && (al & 15) == 0 && npix * (size_t)(n / 4) < (1ull << 31)) {
            hipLaunchKernelGGL(masked_add4_kernel, dim3(grid_for(npix * (n / 4))), dim3(256), 0, st, (const float*)src, src ? (int)s_d[0] : 0,
                               src ? (int)s_d[1] : 0, (const float*)ref, ref ? (int)r_d[0] : 0, ref ? (int)r_d[1] : 0, (const float*)add,
                               add ? (int)a_d[0] : 0, add ? (int)a_d[1] : 0, (float*)dst, (int)d_d[0], (int)d_d[1], (unsigned)npix, n / 4,
                               (int)dims[2]);
            TMG_CHECK_LAUNCH();
            return 0;
        }
    }
    hipLaunchKernelGGL(masked_add_kernel, dim3(grid_for(npix * n)), dim3(256), 0, st, (const float*)src, src ? (int)s_d[0] : 0,
                       src ? (int)s_d[1] : 0, (const float*)ref, ref ? (int)r_d[0] : 0, ref ? (int)r_d[1] : 0, (const float*)add,
                       add ? (int)a_d[0] : 0, add ? (int)a_d[1] : 0, (float*)dst, (int)d_d[0], (int)d_d[1], npix, n, (int)dims[2]);
    TMG_CHECK_LAUNCH();
    return 0;
}

static void fill_segs_pw(TmgSeg* dst, const void* const* ptrs, const int64_t* desc, int n, int* vec4) {
    for (int i = 0; i < TMG_MAX_IN_SEG; ++i) dst[i] = TmgSeg{nullptr, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        dst[i] = TmgSeg{(const float*)ptrs[i], (int)desc[3 * i], (int)desc[3 * i + 1], (int)desc[3 * i + 2]};
        if ((dst[i].stride | dst[i].off | dst[i].n) & 3) *vec4 = 0;
        if (((uintptr_t)ptrs[i]) & 15) *vec4 = 0;
    }
}

static int c1_tile(int W, int H, int* twl) {
    int l = 0;
    while ((1 << l) < W) ++l;
    if (l > 5) l = 5;
    if (l < 3) l = 3;   // >= 8 wide keeps the halo-2 frame of a 256-pixel tile under 512 positions
    *twl = l;
    (void)H;
    return 0;
}

// dims: [B,H,W,Cin,relu_in,w_rows,fill4,w_split,w_gap]; out_d = [stride, off]; w is [w_rows][9] (w_rows = 0 -> Cin);
// fill4 = 1: out points at channel 0 of a 16-byte aligned 4-channel pixel and (value,0,0,0) is stored
// Both growth-1 layers in one launch (see c1x2_fwd_kernel).  dims: [B,H,W,Cin,relu_in,w_rows,w_split,w_gap,w2_d1_row];
// add1_d / add2_d / out_d = [stride, off]; out receives (d1, d2, 0, 0) per pixel as one float4 (16-byte aligned, off 0).
extern "C" int tmg_c1x2_fwd(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* w1, const void* w2,
                            const void* add1, const int64_t* add1_d, const void* add2, const int64_t* add2_d, void* out,
                            const int64_t* out_d, const int64_t* dims, hipStream_t st) {
    C1X2P p;
    p.xmap = tmg_xcd_map_on();
    p.nseg = (int)nseg;
    p.vec4 = 1;
    fill_segs_pw(p.in, in_ptrs, in_desc, (int)nseg, &p.vec4);
    p.B = (int)dims[0]; p.Hin = (int)dims[1]; p.Win = (int)dims[2]; p.Cin = (int)dims[3]; p.relu_in = (int)dims[4];
    p.w_rows = (dims[5] > 0 && dims[5] < p.Cin) ? (int)dims[5] : p.Cin;
    p.w_split = dims[6] > 0 ? (int)dims[6] : 0x7fffffff; p.w_gap = (int)dims[7]; p.w2_d1 = (int)dims[8];
    if (p.Cin & 3) p.vec4 = 0;
    p.pad_rep = 0; p.in_scale = nullptr; p.in_shift = nullptr;
    p.w1 = (const float*)w1; p.w2 = (const float*)w2;
    p.add1 = (const float*)add1; p.a1_stride = add1 ? (int)add1_d[0] : 0; p.a1_off = add1 ? (int)add1_d[1] : 0;
    p.add2 = (const float*)add2; p.a2_stride = add2 ? (int)add2_d[0] : 0; p.a2_off = add2 ? (int)add2_d[1] : 0;
    if ((out_d[0] & 3) || out_d[1] != 0 || (((uintptr_t)out) & 15)) return -2;
    p.out = (float*)out; p.out_stride = (int)out_d[0];
    c1_tile(p.Win, p.Hin, &p.TW_log2);
    // threads per pixel (see the kernel): by the number of 256-pixel tiles the image offers
    int CG = 1;
    {
        const int tw = 1 << p.TW_log2, th = 256 >> p.TW_log2;
        const long t256 = (long)p.B * ((p.Win + tw - 1) / tw) * ((p.Hin + th - 1) / th);
        const int Cq = (p.Cin + 3) / 4;
        // (measured at config M: 4 threads per pixel take the 64- and 128-channel levels from 19 / 32 us to 15; 2 per pixel on the
        // 1 024 tiles of the 32-channel level lose 5 us against one)
        if (t256 < 512 && Cq >= 4) CG = 4;
        else if (t256 < 1024 && Cq >= 2) CG = 2;
        static const int force = getenv("TMG_C1X2_CG") ? atoi(getenv("TMG_C1X2_CG")) : 0;
        if (force == 1 || force == 2 || force == 4) CG = force;
    }
    const int NPX = 256 / CG;
    while ((NPX >> p.TW_log2) < 4 && p.TW_log2 > 3) --p.TW_log2;   // tiles at least 4 rows high: the ring fits the tile's pixels
    const int TW = 1 << p.TW_log2, TH = NPX >> p.TW_log2;
    if (2 * (TW + 2) + 2 * TH > NPX) return -2;
    p.tiles_x = (p.Win + TW - 1) / TW;
    p.tiles_y = (p.Hin + TH - 1) / TH;
    const int Cpad = (p.Cin + 3) & ~3;
    p.KCH = Cpad < 32 ? Cpad : 32;
    size_t lds_floats = (size_t)(TH + 4) * (TW + 4) * (p.KCH + 4) + 18 * p.KCH;
    const size_t d1_floats = 2 * (size_t)(((TH + 2) * (TW + 2) + 3) & ~3);
    if (lds_floats < d1_floats) lds_floats = d1_floats;
    TmgProf prof(TMG_PROF_C1X2, 4.0 * p.B * (double)p.Hin * p.Win * (p.Cin + 4 + (p.add1 ? 1 : 0) + (p.add2 ? 1 : 0)), st);   // input once, D written
    const dim3 grid(p.B * p.tiles_x * p.tiles_y);
    if (CG == 1) {
        TMG_LDS_OPTIN((&c1x2_fwd_kernel<1>));
        hipLaunchKernelGGL(c1x2_fwd_kernel<1>, grid, dim3(256), lds_floats * 4, st, p);
    } else if (CG == 2) {
        TMG_LDS_OPTIN((&c1x2_fwd_kernel<2>));
        hipLaunchKernelGGL(c1x2_fwd_kernel<2>, grid, dim3(256), lds_floats * 4, st, p);
    } else {
        TMG_LDS_OPTIN((&c1x2_fwd_kernel<4>));
        hipLaunchKernelGGL(c1x2_fwd_kernel<4>, grid, dim3(256), lds_floats * 4, st, p);
    }
    TMG_CHECK_LAUNCH();
    return 0;
}

extern "C" int tmg_c1_fwd(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* w, void* out,
                          const int64_t* out_d, const int64_t* dims, hipStream_t st) {
    return tmg_c1_fwd_add(in_ptrs, in_desc, nseg, w, nullptr, nullptr, out, out_d, dims, st);
}

// As tmg_c1_fwd plus a per-pixel scalar `add` ({stride, off}) summed onto the result.
extern "C" int tmg_c1_fwd_add(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* w, const void* add,
                              const int64_t* add_d, void* out, const int64_t* out_d, const int64_t* dims, hipStream_t st) {
    C1P p;
    p.nseg = (int)nseg;
    p.vec4 = 1;
    fill_segs_pw(p.in, in_ptrs, in_desc, (int)nseg, &p.vec4);
    p.B = (int)dims[0]; p.Hin = (int)dims[1]; p.Win = (int)dims[2]; p.Cin = (int)dims[3]; p.relu_in = (int)dims[4];
    p.w_rows = (dims[5] > 0 && dims[5] < p.Cin) ? (int)dims[5] : p.Cin;
    p.fill4 = (int)dims[6];
    p.w_split = dims[7] > 0 ? (int)dims[7] : 0x7fffffff; p.w_gap = (int)dims[8];
    p.add = (const float*)add; p.add_stride = add ? (int)add_d[0] : 0; p.add_off = add ? (int)add_d[1] : 0;
    if (p.Cin & 3) p.vec4 = 0;
    p.pad_rep = 0; p.in_scale = nullptr; p.in_shift = nullptr;
    p.w = (const float*)w; p.out = (float*)out; p.out_stride = (int)out_d[0]; p.out_off = (int)out_d[1];
    c1_tile(p.Win, p.Hin, &p.TW_log2);
    const int TW = 1 << p.TW_log2, TH = 256 >> p.TW_log2;
    p.tiles_x = (p.Win + TW - 1) / TW;
    p.tiles_y = (p.Hin + TH - 1) / TH;
    const int Cpad = (p.Cin + 3) & ~3;
    { const char* e = getenv("TMG_C1_KCH"); const int k = e ? atoi(e) : 32; p.KCH = Cpad < k ? Cpad : k; }
    const size_t lds_bytes = ((size_t)(TH + 2) * (TW + 2) * (p.KCH + 4) + 9 * p.KCH) * 4;
    TMG_LDS_OPTIN((&c1_fwd_kernel));
    hipLaunchKernelGGL(c1_fwd_kernel, dim3(p.B * p.tiles_x * p.tiles_y), dim3(256), lds_bytes, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

// dims: [B,H,W,Cin,relu_in]; dd_d/dref_d = [stride, off]; g segments accumulate (+=); dW accumulates atomically
extern "C" int tmg_c1_bwd(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* w, void* dW, const void* dd,
                          const int64_t* dd_d, const void* dref, const int64_t* dref_d, void* const* g_ptrs, const int64_t* g_desc,
                          int64_t ng, const int64_t* dims, hipStream_t st) {
    C1BP p;
    p.nseg = (int)nseg;
    p.vec4 = 1;
    fill_segs_pw(p.in, in_ptrs, in_desc, (int)nseg, &p.vec4);
    p.B = (int)dims[0]; p.Hin = (int)dims[1]; p.Win = (int)dims[2]; p.Cin = (int)dims[3]; p.relu_in = (int)dims[4];
    if (p.Cin & 3) p.vec4 = 0;
    p.pad_rep = 0; p.in_scale = nullptr; p.in_shift = nullptr;
    p.w = (const float*)w; p.dW = (float*)dW;
    p.dd = (const float*)dd; p.dd_stride = (int)dd_d[0]; p.dd_off = (int)dd_d[1];
    p.dref = (const float*)dref; p.dref_stride = dref ? (int)dref_d[0] : 0; p.dref_off = dref ? (int)dref_d[1] : 0;
    for (int i = 0; i < TMG_MAX_OUT_SEG; ++i) p.g[i] = TmgOSeg{nullptr, 0, 0, 0};
    for (int i = 0; i < (int)ng; ++i)
        p.g[i] = TmgOSeg{(float*)g_ptrs[i], (int)g_desc[3 * i], (int)g_desc[3 * i + 1], (int)g_desc[3 * i + 2]};
    c1_tile(p.Win, p.Hin, &p.TW_log2);
    const int TW = 1 << p.TW_log2, TH = 256 >> p.TW_log2;
    p.tiles_x = (p.Win + TW - 1) / TW;
    p.tiles_y = (p.Hin + TH - 1) / TH;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    const int Cpad = (p.Cin + 3) & ~3;
    p.KCH = Cpad < 32 ? Cpad : 32;
    if ((Cpad + p.KCH - 1) / p.KCH > 8) return -2;  // Cin <= 256
    const int PP = (TH + 2) * (TW + 2);
    const size_t lds_bytes = ((size_t)((PP + 3) & ~3) + 9 * p.KCH + (size_t)PP * (p.KCH + 4)) * 4;
    TMG_LDS_OPTIN((&c1_bwd_kernel));
    int gx = p.ntiles < 1024 ? p.ntiles : 1024;
    hipLaunchKernelGGL(c1_bwd_kernel, dim3(gx), dim3(256), lds_bytes, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

// Fused backward of both growth-1 layers (see dense2_bwd_kernel).
// in segments: nn inputs followed by D (4 channels); dims = {B,H,W,Cin_total (incl. D's 4),cin_nn,rows1,rows2,dd1_out ptr,dd2_out ptr,dd stride,split2,gap2,dd_quad}
// (dd_quad: dd1 / dd2 are channels 0, 1 of a 16-byte aligned float4 slot per pixel; the kernel writes (dd1, dd2, 0, 0) in one store)
// g0/out: up to two segments each (same channel split as the nn inputs); add0 optional (null) added to out segment 0.
extern "C" int tmg_dense2_bwd(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* w1, const void* w2,
                              void* dW1, void* dW2, const void* GD, int64_t gd_stride, const void* Dp, int64_t d_stride,
                              const void* const* g0_ptrs, const int64_t* g0_desc, void* const* out_ptrs, const int64_t* out_desc,
                              int64_t ng, const void* add0, int64_t add0_stride, const int64_t* dims, hipStream_t st) {
    D2BP p;
    p.xmap = tmg_xcd_map_on();
    p.nseg = (int)nseg;
    p.vec4 = 1;
    fill_segs_pw(p.in, in_ptrs, in_desc, (int)nseg, &p.vec4);
    p.B = (int)dims[0]; p.Hin = (int)dims[1]; p.Win = (int)dims[2]; p.Cin = (int)dims[3]; p.cin_nn = (int)dims[4];
    p.rows1 = (int)dims[5]; p.rows2 = (int)dims[6];
    p.relu_in = 1; p.pad_rep = 0; p.in_scale = nullptr; p.in_shift = nullptr;
    if (p.Cin & 3) p.vec4 = 0;
    p.w1 = (const float*)w1; p.w2 = (const float*)w2; p.dW1 = (float*)dW1; p.dW2 = (float*)dW2;
    p.GD = (const float*)GD; p.gd_stride = (int)gd_stride; p.Dp = (const float*)Dp; p.d_stride = (int)d_stride;
    for (int i = 0; i < 2; ++i) { p.g0[i] = TmgSeg{nullptr, 0, 0, 0}; p.out[i] = TmgOSeg{nullptr, 0, 0, 0}; }
    for (int i = 0; i < (int)ng; ++i) {
        p.g0[i] = TmgSeg{(const float*)g0_ptrs[i], (int)g0_desc[3 * i], (int)g0_desc[3 * i + 1], (int)g0_desc[3 * i + 2]};
        p.out[i] = TmgOSeg{(float*)out_ptrs[i], (int)out_desc[3 * i], (int)out_desc[3 * i + 1], (int)out_desc[3 * i + 2]};
        if ((p.g0[i].stride | p.g0[i].off | p.g0[i].n | p.out[i].stride | p.out[i].off) & 3) p.vec4 = 0;
        if ((((uintptr_t)g0_ptrs[i]) | ((uintptr_t)out_ptrs[i])) & 15) p.vec4 = 0;
    }
    p.add0 = (const float*)add0; p.add0_stride = (int)add0_stride;
    p.dd1_out = (float*)dims[7]; p.dd2_out = (float*)dims[8]; p.dd_stride = (int)dims[9];
    p.dd_quad = dims[12] && p.dd1_out && p.dd2_out == p.dd1_out + 1 && !(p.dd_stride & 3) && !(((uintptr_t)p.dd1_out) & 15);
    if (dims[12] && !p.dd_quad) return -4;   // the caller relies on the zero channels
    p.split2 = dims[10] > 0 ? (int)dims[10] : 0x7fffffff; p.gap2 = (int)dims[11];
    if (add0 && ((add0_stride & 3) || (((uintptr_t)add0) & 15))) p.vec4 = 0;
    c1_tile(p.Win, p.Hin, &p.TW_log2);
    const int TW = 1 << p.TW_log2, TH = 256 >> p.TW_log2;
    p.tiles_x = (p.Win + TW - 1) / TW;
    p.tiles_y = (p.Hin + TH - 1) / TH;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    const int Cpad = (p.Cin + 3) & ~3;
    p.KCH = Cpad < 28 ? Cpad : 28;  // KCH*9 <= 256: one weight-gradient output per thread
    const int nchunks = (Cpad + p.KCH - 1) / p.KCH;
    const int PP = (TH + 2) * (TW + 2), QQ = (TH + 4) * (TW + 4);
    const size_t lds_bytes = ((size_t)((QQ + 3) & ~3) + ((PP + 3) & ~3) + 18 * p.KCH + (size_t)PP * (p.KCH + 8)) * 4;
    TMG_LDS_OPTIN((&dense2_bwd_kernel<true>));
    TMG_LDS_OPTIN((&dense2_bwd_kernel<false>));
    // blocks per channel chunk: an even number of tiles per block (an uneven split costs ~10 %: 768 blocks for 4 096 tiles 2.79 ms per
    // step, 512 or 1 024 blocks 2.50); with the weight gradients inside the kernel fewer blocks win (one atomic per block and weight:
    // 512 blocks 137 us, 1 024 blocks 180 us at 64 x 128 x 128)
    static const int d2_env = getenv("TMG_D2_BLOCKS") ? atoi(getenv("TMG_D2_BLOCKS")) : 0;
    const int d2_blocks = d2_env > 0 ? d2_env : (p.dW1 ? 512 : 1024);
    int gx = d2_blocks / nchunks;
    if (gx < 1) gx = 1;
    {
        const int per_blk = (p.ntiles + gx - 1) / gx;
        gx = (p.ntiles + per_blk - 1) / per_blk;
    }
    TmgProf prof(TMG_PROF_D2B, 4.0 * p.B * (double)p.Hin * p.Win * (3.0 * p.cin_nn + 4 + 4 + 4 + 2), st);   // x, G0 read, dx written; D, GD; add0 ~ included in 3 cin
    if (!p.dW1 && !p.dW2 && p.vec4 && ng == 1 && p.nseg == 2 && p.in[0].n == p.cin_nn && p.in[1].n == 4 && (p.cin_nn & 3) == 0 &&
        p.g0[0].n == p.cin_nn && p.dd1_out &&
        (p.dd_quad || (p.dd2_out == p.dd1_out + 1 && !(p.dd_stride & 1) && !(((uintptr_t)p.dd1_out) & 7))) &&
        (double)p.B * p.Hin * p.Win * 4.0 * (double)std::max(std::max(std::max(p.in[0].stride, p.g0[0].stride), std::max(p.out[0].stride, p.add0_stride)),
                                                              std::max(std::max(p.gd_stride, p.d_stride), p.dd_stride)) < 4.0e9) {
        // the level node's call: no staged patch, <= 128 registers (see dense2_bwd_lean_kernel)
        const int nq = p.cin_nn / 4, NQ = nq <= 2 ? 2 : 4, nch = (nq + NQ - 1) / NQ;
        int gl = d2_blocks / nch;
        if (gl < 1) gl = 1;
        const int per_blk = (p.ntiles + gl - 1) / gl;
        gl = (p.ntiles + per_blk - 1) / per_blk;
        const size_t lb = ((size_t)((QQ + 3) & ~3) + ((PP + 3) & ~3) + 18 * 4 * NQ) * 4;
#define TMG_D2L(NQ_, TWL_) hipLaunchKernelGGL((dense2_bwd_lean_kernel<NQ_, TWL_>), dim3(gl, nch), dim3(256), lb, st, p)
        if (NQ == 2) { if (p.TW_log2 == 5) TMG_D2L(2, 5); else if (p.TW_log2 == 4) TMG_D2L(2, 4); else TMG_D2L(2, 3); }
        else { if (p.TW_log2 == 5) TMG_D2L(4, 5); else if (p.TW_log2 == 4) TMG_D2L(4, 4); else TMG_D2L(4, 3); }
#undef TMG_D2L
        TMG_CHECK_LAUNCH();
        return 0;
    }
    if (p.dW1) hipLaunchKernelGGL(dense2_bwd_kernel<true>, dim3(gx, nchunks), dim3(256), lds_bytes, st, p);
    else hipLaunchKernelGGL(dense2_bwd_kernel<false>, dim3(gx, nchunks), dim3(256), lds_bytes, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

extern "C" int tmg_dkappa(const void* w, const void* dw, int64_t nw, const void* b, const void* db, int64_t nb, const void* kappa,
                          void* dk, hipStream_t st) {
    int blocks = (int)((nw + 2047) / 2048);
    if (blocks > 64) blocks = 64;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(dkappa_kernel, dim3(blocks), dim3(256), 0, st, (const float*)w, (const float*)dw, (int)nw, (const float*)b,
                       (const float*)db, (int)nb, (const float*)kappa, (float*)dk);
    TMG_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================================================
// Parameter-side folding of a flow level: ActNorm + PLU-parameterised invertible 1x1 conv of ALL K layers -> the [K,C,C] mix matrices
// and [K,C] biases the coupling kernels consume, and the backward of that map - two launches per level instead of the ~70 tiny
// torch launches (stack / mask / diag_embed / bmm / scale and their autograd) the same arithmetic cost before.
//   lower = l (.) strictly-lower mask + I,  upper = u (.) strictly-upper mask + diag(exp(log_s) sign_s) + 0.01 I      (glowConv.py:151-160)
//   W = P lower upper                                                                                           (glowConv.py:161)
//   reverse: Wm = diag(1/a) W, bm = -b / a          forward: Wm = W diag(a), bm = W b                           (actNorm.py:66-83 folded)
//   ld = hw (sgn sum log_s + sum log|a|)
// Parameters stay the module's own per-layer tensors: the kernels read them through a device pointer table [K][5] (l, u, log_s, a, b;
// a / b null: no ActNorm); P enters as the row permutation perm[k][i] (row i of W = row perm of lower upper).
// =================================================================================================================================
struct LuFoldP {
    const long long* tab;     // [K][5] pointers
    const float* sign_s;      // [K][C]
    const int* perm;          // [K][C]: P[i][perm[i]] = 1
    const int* iperm;         // [K][C]: inverse
    int K, C, reverse;
    float sgn, hw;
};

// Arithmetic: fp64 throughout, rounded once to fp32 on store.  The mixes are applied to every pixel of every sample, so a rounding
// error in W is COHERENT over the whole field and - unlike per-pixel rounding noise - does not average out in the weight-gradient
// sums: with fp32 dot products and a rounded reciprocal 1 / a shared by a whole row (and inconsistent with the exactly divided bias),
// the generative direction's gradients sat at 4.1e-4 global rel-L2 from the fp64 oracle on config M against 3.7e-5 with torch's fp32
// fold (profiles/r3_parity_report_*.json); in fp64 the fold is exact to the last fp32 bit and costs nothing (K C^3 flops).
__device__ __forceinline__ double lu_lower(const float* l, int C, int r, int k) { return k < r ? (double)l[r * C + k] : (k == r ? 1.0 : 0.0); }
// diag: exp(log_s) sign_s + 0.01 per channel, staged in LDS once per block (an exp per inner-loop term made the kernels 3x slower)
__device__ __forceinline__ double lu_upper(const float* u, const double* diag, int C, int k, int j) {
    return k < j ? (double)u[k * C + j] : (k == j ? diag[k] : 0.0);
}

__device__ __forceinline__ double block_sum_256_f64(double v, double* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void lu_fold_fwd_kernel(LuFoldP p, double* __restrict__ W, float* __restrict__ Wm, float* __restrict__ bm,
                                                          float* __restrict__ ld) {
    const int C = p.C, k = blockIdx.y;
    const long long* t = p.tab + (size_t)k * 5;
    const float* l = reinterpret_cast<const float*>(t[0]);
    const float* u = reinterpret_cast<const float*>(t[1]);
    const float* ls = reinterpret_cast<const float*>(t[2]);
    const float* a = reinterpret_cast<const float*>(t[3]);
    const float* b = reinterpret_cast<const float*>(t[4]);
    const float* sg = p.sign_s + (size_t)k * C;
    const int* perm = p.perm + (size_t)k * C;
    __shared__ double diag[256], sa[256];
    for (int i = threadIdx.x; i < C; i += 256) {
        diag[i] = exp((double)ls[i]) * (double)sg[i] + 0.01;
        sa[i] = a ? (p.reverse ? 1.0 / (double)a[i] : (double)a[i]) : 1.0;
    }
    __syncthreads();
    for (int e = blockIdx.x * 256 + threadIdx.x; e < C * C; e += gridDim.x * 256) {
        const int i = e / C, j = e - i * C, r = perm[i];
        // row r of lower times column j of upper: terms q < min(r, j) are plain products, the last one involves a unit / diagonal entry
        const int kmin = min(r, j);
        double acc = r < j ? (double)u[r * C + j] : (r == j ? diag[j] : (double)l[r * C + j] * diag[j]);
        // four partial sums: one fma chain of up to C terms with a global load per term is a chain of exposed latencies
        double a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int q = 0;
        for (; q + 4 <= kmin; q += 4) {
            acc = fma((double)l[r * C + q], (double)u[q * C + j], acc);
            a1 = fma((double)l[r * C + q + 1], (double)u[(q + 1) * C + j], a1);
            a2 = fma((double)l[r * C + q + 2], (double)u[(q + 2) * C + j], a2);
            a3 = fma((double)l[r * C + q + 3], (double)u[(q + 3) * C + j], a3);
        }
        for (; q < kmin; ++q) acc = fma((double)l[r * C + q], (double)u[q * C + j], acc);
        acc = (acc + a1) + (a2 + a3);
        W[((size_t)k * C + i) * C + j] = acc;       // fp64 copy of W = P L U for the backward launch
        Wm[((size_t)k * C + i) * C + j] = (float)(acc * (p.reverse ? sa[i] : sa[j]));
    }
    if (blockIdx.x == 0) {
        // biases need whole rows of W: recomputed here from the factors (C <= 256: a few thousand flops per thread)
        for (int i = threadIdx.x; i < C; i += 256) {
            double v;
            if (p.reverse) v = b ? -(double)b[i] / (a ? (double)a[i] : 1.0) : 0.0;
            else {
                v = 0.0;
                if (b) {
                    const int r = perm[i];
                    for (int j = 0; j < C; ++j) {
                        double acc = 0.0;
                        const int kmax = min(r, j);
                        for (int q = 0; q <= kmax; ++q) acc = fma(lu_lower(l, C, r, q), lu_upper(u, diag, C, q, j), acc);
                        v = fma(acc, (double)b[j], v);
                    }
                }
            }
            bm[(size_t)k * C + i] = (float)v;
        }
        if (k == 0) {
            // scalar log-det of all K mixes (one block: K C <= 4096 terms)
            __shared__ double red[4];
            double s = 0.0;
            for (int e = threadIdx.x; e < p.K * C; e += 256) {
                const int kk = e / C, i = e - kk * C;
                const long long* tt = p.tab + (size_t)kk * 5;
                const float* aa = reinterpret_cast<const float*>(tt[3]);
                s += (double)p.sgn * (double)reinterpret_cast<const float*>(tt[2])[i] + (aa ? log(fabs((double)aa[i])) : 0.0);
            }
            const double tot = block_sum_256_f64(s, red);
            if (threadIdx.x == 0) ld[0] = (float)(tot * (double)p.hw);
        }
    }
}

// Backward: dWm [K,C,C], dbm [K,C], dld (device scalar) -> dl, du [K,C,C] (zero outside the masks), dlog_s, da, db [K,C].
// dWt / dbt (optional): the last layer's upstream gradients live in tensors of their own ([C,C] / [C]); dWm / dbm then hold the
// first K - 1 layers (the level-fused node consumes layers 0..K-2 as one slice, the ConvLSTM layer takes the last one).
__global__ __launch_bounds__(256) void lu_fold_bwd_kernel(LuFoldP p, const double* __restrict__ W, const float* __restrict__ dWm,
                                                          const float* __restrict__ dbm, const float* __restrict__ dWt,
                                                          const float* __restrict__ dbt, const float* __restrict__ dld, float* __restrict__ dl,
                                                          float* __restrict__ du, float* __restrict__ dlogs, float* __restrict__ da,
                                                          float* __restrict__ db) {
    const int C = p.C, k = blockIdx.y;
    const long long* t = p.tab + (size_t)k * 5;
    const float* l = reinterpret_cast<const float*>(t[0]);
    const float* u = reinterpret_cast<const float*>(t[1]);
    const float* ls = reinterpret_cast<const float*>(t[2]);
    const float* a = reinterpret_cast<const float*>(t[3]);
    const float* b = reinterpret_cast<const float*>(t[4]);
    const float* sg = p.sign_s + (size_t)k * C;
    const int* iperm = p.iperm + (size_t)k * C;
    const bool tail = dWt != nullptr && k == p.K - 1;
    const float* dWk = tail ? dWt : dWm + (size_t)k * C * C;
    const float* dbk = tail ? dbt : (dbm ? dbm + (size_t)k * C : nullptr);
    const double g = dld ? (double)dld[0] : 0.0;
    __shared__ double diag[256], sa[256], sb[256], sdb[256];
    __shared__ int sip[256];      // the inverse permutation: read per term of the upper-factor sums (a dependent global load otherwise)
    for (int i = threadIdx.x; i < C; i += 256) {
        sip[i] = iperm[i];
        diag[i] = exp((double)ls[i]) * (double)sg[i] + 0.01;
        sa[i] = a ? (p.reverse ? 1.0 / (double)a[i] : (double)a[i]) : 1.0;
        sb[i] = (!p.reverse && b && dbk) ? (double)b[i] : 0.0;
        sdb[i] = (!p.reverse && b && dbk) ? (double)dbk[i] : 0.0;
    }
    __syncthreads();
    // gradient w.r.t. W (before the ActNorm fold) at row i, column j
#define TMG_LU_DW(I, J) (p.reverse ? (double)dWk[(I) * C + (J)] * sa[I] : fma((double)dWk[(I) * C + (J)], sa[J], sdb[I] * sb[J]))
    for (int e = blockIdx.x * 256 + threadIdx.x; e < C * C; e += gridDim.x * 256) {
        const int x = e / C, y = e - x * C;
        double vl = 0.0, vu = 0.0;
        if (x > y) {
            // dlower[x][y] = sum_j M[x][j] upper[y][j], M = P^T dW: M[x][j] = dW[sip[x]][j]; upper[y][j] = 0 for j < y
            const int i = sip[x];
            vl = TMG_LU_DW(i, y) * diag[y];
            double v1 = 0.0, v2 = 0.0, v3 = 0.0;   // four partial sums (see the forward kernel)
            int j = y + 1;
            for (; j + 4 <= C; j += 4) {
                vl = fma(TMG_LU_DW(i, j), (double)u[y * C + j], vl);
                v1 = fma(TMG_LU_DW(i, j + 1), (double)u[y * C + j + 1], v1);
                v2 = fma(TMG_LU_DW(i, j + 2), (double)u[y * C + j + 2], v2);
                v3 = fma(TMG_LU_DW(i, j + 3), (double)u[y * C + j + 3], v3);
            }
            for (; j < C; ++j) vl = fma(TMG_LU_DW(i, j), (double)u[y * C + j], vl);
            vl = (vl + v1) + (v2 + v3);
        } else {
            // dupper[x][y] = sum_r lower[r][x] M[r][y]; lower[r][x] = 0 for r < x
            vu = TMG_LU_DW(sip[x], y);
            double v1 = 0.0, v2 = 0.0, v3 = 0.0;
            int r = x + 1;
            for (; r + 4 <= C; r += 4) {
                vu = fma((double)l[r * C + x], TMG_LU_DW(sip[r], y), vu);
                v1 = fma((double)l[(r + 1) * C + x], TMG_LU_DW(sip[r + 1], y), v1);
                v2 = fma((double)l[(r + 2) * C + x], TMG_LU_DW(sip[r + 2], y), v2);
                v3 = fma((double)l[(r + 3) * C + x], TMG_LU_DW(sip[r + 3], y), v3);
            }
            for (; r < C; ++r) vu = fma((double)l[r * C + x], TMG_LU_DW(sip[r], y), vu);
            vu = (vu + v1) + (v2 + v3);
        }
        dl[((size_t)k * C + x) * C + y] = (float)vl;
        du[((size_t)k * C + x) * C + y] = x < y ? (float)vu : 0.f;
        if (x == y) dlogs[(size_t)k * C + x] = (float)(vu * (diag[x] - 0.01) + (double)p.sgn * (double)p.hw * g);
    }
    if (blockIdx.x == 0) {
        const double* Wk = W + (size_t)k * C * C;
        for (int i = threadIdx.x; i < C; i += 256) {
            double va = 0.0, vb = 0.0;
            const double ai = a ? (double)a[i] : 1.0;
            if (a) {
                if (p.reverse) {
                    // Wm = W / a_i, bm = -b_i / a_i
                    double s = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                    int j = 0;
                    for (; j + 4 <= C; j += 4) {
                        s = fma((double)dWk[i * C + j], Wk[i * C + j], s);
                        s1 = fma((double)dWk[i * C + j + 1], Wk[i * C + j + 1], s1);
                        s2 = fma((double)dWk[i * C + j + 2], Wk[i * C + j + 2], s2);
                        s3 = fma((double)dWk[i * C + j + 3], Wk[i * C + j + 3], s3);
                    }
                    for (; j < C; ++j) s = fma((double)dWk[i * C + j], Wk[i * C + j], s);
                    s = (s + s1) + (s2 + s3);
                    va = -s / (ai * ai) + ((dbk && b) ? (double)dbk[i] * (double)b[i] / (ai * ai) : 0.0) + g * (double)p.hw / ai;
                } else {
                    // Wm = W a_j: da_j = sum_i dWm[i][j] W[i][j]   (index i of this thread plays the role of j)
                    double s = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                    int r = 0;
                    for (; r + 4 <= C; r += 4) {
                        s = fma((double)dWk[r * C + i], Wk[r * C + i], s);
                        s1 = fma((double)dWk[(r + 1) * C + i], Wk[(r + 1) * C + i], s1);
                        s2 = fma((double)dWk[(r + 2) * C + i], Wk[(r + 2) * C + i], s2);
                        s3 = fma((double)dWk[(r + 3) * C + i], Wk[(r + 3) * C + i], s3);
                    }
                    for (; r < C; ++r) s = fma((double)dWk[r * C + i], Wk[r * C + i], s);
                    s = (s + s1) + (s2 + s3);
                    va = s + g * (double)p.hw / ai;
                }
            }
            if (b && dbk) {
                if (p.reverse) vb = -(double)dbk[i] / ai;
                else {
                    for (int r = 0; r < C; ++r) vb = fma(Wk[r * C + i], (double)dbk[r], vb);   // db = W^T dbm
                }
            }
            da[(size_t)k * C + i] = (float)va;
            db[(size_t)k * C + i] = (float)vb;
        }
    }
#undef TMG_LU_DW
}

// dims = {K, C, reverse}; fl = {sgn (+1 / -1: sign of the log_s term of the log-det), hw}.  tab: device int64 [K][5]; sign_s [K][C];
// perm / iperm: device int32 [K][C].  Outputs W [K,C,C] FP64 (scratch for the backward launch), Wm [K,C,C], bm [K,C], ld [1] fp32.
extern "C" int tmg_lu_fold_fwd(const void* tab, const void* sign_s, const void* perm, const void* iperm, void* W, void* Wm, void* bm, void* ld,
                               const int64_t* dims, const float* fl, hipStream_t st) {
    LuFoldP p;
    p.tab = (const long long*)tab; p.sign_s = (const float*)sign_s; p.perm = (const int*)perm; p.iperm = (const int*)iperm;
    p.K = (int)dims[0]; p.C = (int)dims[1]; p.reverse = (int)dims[2]; p.sgn = fl[0]; p.hw = fl[1];
    if (p.K < 1 || p.C < 1 || p.C > 256) return -1;     // per-channel vectors live in 256-entry LDS arrays
    const int gx = (p.C * p.C + 255) / 256 < 64 ? (p.C * p.C + 255) / 256 : 64;
    hipLaunchKernelGGL(lu_fold_fwd_kernel, dim3(gx, p.K), dim3(256), 0, st, p, (double*)W, (float*)Wm, (float*)bm, (float*)ld);
    TMG_CHECK_LAUNCH();
    return 0;
}

// Inverse of the K folded channel mixes of a level, for the recompute-from-output backward (tmg_ops.LevelCouplingFn, round 5): a
// coupling layer of the generative direction ends with out = Wm [x1; y2] + bm (glowConv.py:207-222 + actNorm.py:71-85 folded), so its
// input is rebuilt from its output with Winv = Wm^-1, binv = -Wm^-1 bm.  (The reference's own forward-direction matrix is NOT that
// inverse: glowConv.py:164-174 adds 0.01 to U's diagonal.)  One block per matrix, Gauss-Jordan with partial pivoting in FP64 on the
// augmented [C][2C] array in LDS, rounded once on store: an error in a mix matrix is coherent over every pixel it is applied to.
__global__ __launch_bounds__(256) void mat_inverse_kernel(const float* __restrict__ W, const float* __restrict__ b, float* __restrict__ Winv,
                                                          float* __restrict__ binv, int C) {
    extern __shared__ __attribute__((aligned(16))) double aug[];      // [C][2C], then C doubles of pivot-column factors, then 2 + C ints
    double* fac = aug + 2 * C * C;
    int* piv = reinterpret_cast<int*>(fac + C);
    int* dead = piv + 2;              // [C]: channel with an all-zero row and column (zero-padded halves of 3-channel fields, section 3 of
                                      // DESIGN.md): the mix is block diagonal with a zero block there - inverted on the live block, zero elsewhere
    const int k = blockIdx.x, tid = threadIdx.x, C2 = 2 * C;
    const float* Wk = W + (size_t)k * C * C;
    for (int i = tid; i < C * C2; i += 256) {
        const int r = i / C2, c = i - r * C2;
        aug[i] = c < C ? (double)Wk[r * C + c] : (c - C == r ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int col = 0; col < C; ++col) {
        if (tid == 0) {
            int best = col;
            double bv = fabs(aug[col * C2 + col]);
            for (int r = col + 1; r < C; ++r) {
                const double v = fabs(aug[r * C2 + col]);
                if (v > bv) { bv = v; best = r; }
            }
            piv[0] = best;
            dead[col] = bv == 0.0 ? 1 : 0;
        }
        __syncthreads();
        if (dead[col]) continue;      // (block-uniform; rows / columns of a padding channel hold zeros and take no part in the elimination)
        const int pr = piv[0];
        if (pr != col)
            for (int c = tid; c < C2; c += 256) { const double t = aug[col * C2 + c]; aug[col * C2 + c] = aug[pr * C2 + c]; aug[pr * C2 + c] = t; }
        __syncthreads();
        const double inv = 1.0 / aug[col * C2 + col];
        for (int r = tid; r < C; r += 256) fac[r] = aug[r * C2 + col];
        __syncthreads();
        for (int c = tid; c < C2; c += 256) aug[col * C2 + c] *= inv;
        __syncthreads();
        for (int i = tid; i < C * C2; i += 256) {
            const int r = i / C2, c = i - r * C2;
            if (r != col) aug[i] -= fac[r] * aug[col * C2 + c];
        }
        __syncthreads();
    }
    float* Wo = Winv + (size_t)k * C * C;
    for (int i = tid; i < C * C; i += 256) {
        const int r = i / C, c = i - r * C;
        Wo[i] = dead[r] ? 0.f : (float)aug[r * C2 + C + c];
    }
    if (b && binv)
        for (int r = tid; r < C; r += 256) {
            double acc = 0.0;
            for (int c = 0; c < C; ++c) acc += aug[r * C2 + C + c] * (double)b[(size_t)k * C + c];
            binv[(size_t)k * C + r] = dead[r] ? 0.f : (float)(-acc);
        }
}

// Winv[k] = W[k]^-1 (fp64 arithmetic), binv[k] = -Winv[k] b[k] (b / binv may be null).  dims = {K, C}; C <= 64.
extern "C" int tmg_mat_inverse(const void* W, const void* b, void* Winv, void* binv, const int64_t* dims, hipStream_t st) {
    const int K = (int)dims[0], C = (int)dims[1];
    if (K < 1 || C < 1 || C > 64) return -1;
    const size_t lds = ((size_t)2 * C * C + C) * sizeof(double) + 16 + (size_t)C * sizeof(int);
    if (lds > 64 * 1024) TMG_LDS_OPTIN((&mat_inverse_kernel));
    hipLaunchKernelGGL(mat_inverse_kernel, dim3(K), dim3(256), lds, st, (const float*)W, (const float*)b, (float*)Winv, (float*)binv, C);
    TMG_CHECK_LAUNCH();
    return 0;
}

extern "C" int tmg_lu_fold_bwd_split(const void* tab, const void* sign_s, const void* perm, const void* iperm, const void* W,
                                     const void* dWm, const void* dbm, const void* dWm_tail, const void* dbm_tail, const void* dld, void* dl,
                                     void* du, void* dlogs, void* da, void* db, const int64_t* dims, const float* fl, hipStream_t st) {
    LuFoldP p;
    p.tab = (const long long*)tab; p.sign_s = (const float*)sign_s; p.perm = (const int*)perm; p.iperm = (const int*)iperm;
    p.K = (int)dims[0]; p.C = (int)dims[1]; p.reverse = (int)dims[2]; p.sgn = fl[0]; p.hw = fl[1];
    if (p.K < 1 || p.C < 1 || p.C > 256) return -1;     // per-channel vectors live in 256-entry LDS arrays
    const int gx = (p.C * p.C + 255) / 256 < 64 ? (p.C * p.C + 255) / 256 : 64;
    if (!dWm && !(dWm_tail && p.K == 1)) return -1;
    hipLaunchKernelGGL(lu_fold_bwd_kernel, dim3(gx, p.K), dim3(256), 0, st, p, (const double*)W, (const float*)dWm, (const float*)dbm,
                       (const float*)dWm_tail, (const float*)dbm_tail, (const float*)dld, (float*)dl, (float*)du, (float*)dlogs, (float*)da,
                       (float*)db);
    TMG_CHECK_LAUNCH();
    return 0;
}

extern "C" int tmg_lu_fold_bwd(const void* tab, const void* sign_s, const void* perm, const void* iperm, const void* W, const void* dWm,
                               const void* dbm, const void* dld, void* dl, void* du, void* dlogs, void* da, void* db, const int64_t* dims,
                               const float* fl, hipStream_t st) {
    return tmg_lu_fold_bwd_split(tab, sign_s, perm, iperm, W, dWm, dbm, nullptr, nullptr, dld, dl, du, dlogs, da, db, dims, fl, st);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Parameter-gradient epilogue of a level's fused coupling node (LevelCouplingFn.backward), one launch for all NL layers:
//   dK_k   = (<Wz_k, dWz_k> + <bz_k, dBz_k>) [-4 <= kappa_k <= ln 4]      (homogeneity of the zero conv in (W, b); flowUtils.py:104-106;
//            the two inner products nearly cancel for small gradients, hence fp64 accumulation)
//   dW1_k[0, j]   += tmpX_k[0, j] (j < ch),  tmpC_k[0, j - ch] (ch <= j < cin)
//   dW2_k[0, j]   += tmpX_k[1, j] (j < ch),  tmpC_k[1, j - ch] (ch <= j < cin),  tmpX_k[1, ch] (j = cin: the d1 input row)
// tmpX [NL,4,ch+4,3,3] / tmpC [NL,4,Cc,3,3]: the grouped weight-gradient launches' 4-row results (rows 0 / 1 = growth layers 1 / 2).
// acc / cnt: zero-initialised fp64 sums and arrival counters per layer (the last of the S blocks of a layer writes dK).
struct LevelFinP {
    const float *Wz, *dWz, *Bz, *dBz, *Kp, *tmpX, *tmpC;
    float *dW1, *dW2, *dK;
    double* acc;
    unsigned* cnt;
    int NL, C, ch, Cc, S;
    int rowsC;      // rows per layer in tmpC: 4 (the grouped launches' quad rows) or 2 (compact: the level-wide conditioning launch)
};

__global__ __launch_bounds__(256) void level_finish_kernel(LevelFinP p) {
    const int k = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int cin = p.ch + p.Cc;
    const size_t nz = (size_t)p.C * (cin + 2) * 9;
    const float* w = p.Wz + (size_t)k * nz;
    const float* g = p.dWz + (size_t)k * nz;
    double a = 0.0;
    if ((nz & 3) == 0 && ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(g)) & 15) == 0) {
        const float4* w4 = reinterpret_cast<const float4*>(w);
        const float4* g4 = reinterpret_cast<const float4*>(g);
        for (size_t i = (size_t)s * 256 + tid; i < nz / 4; i += (size_t)p.S * 256) {
            const float4 x = w4[i], y = g4[i];
            a += (double)x.x * (double)y.x + (double)x.y * (double)y.y + (double)x.z * (double)y.z + (double)x.w * (double)y.w;
        }
    } else {
        for (size_t i = (size_t)s * 256 + tid; i < nz; i += (size_t)p.S * 256) a += (double)w[i] * (double)g[i];
    }
    if (s == 0) {
        for (int i = tid; i < p.C; i += 256) a += (double)p.Bz[(size_t)k * p.C + i] * (double)p.dBz[(size_t)k * p.C + i];
    }
    if (s == p.S - 1) {
        // scatter of the grouped launches' rows into the native weight-gradient layout
        const float* tx = p.tmpX ? p.tmpX + (size_t)k * 4 * (p.ch + 4) * 9 : nullptr;
        const float* tc = p.tmpC ? p.tmpC + (size_t)k * p.rowsC * p.Cc * 9 : nullptr;
        float* d1 = p.dW1 + (size_t)k * cin * 9;
        float* d2 = p.dW2 + (size_t)k * (cin + 1) * 9;
        for (int e = tid; e < (cin + 1) * 9; e += 256) {
            const int j = e / 9, t = e - j * 9;
            float v1 = 0.f, v2 = 0.f;
            if (j < p.ch) {
                if (tx) { v1 = tx[j * 9 + t]; v2 = tx[((p.ch + 4) + j) * 9 + t]; }
            } else if (j < cin) {
                if (tc) { v1 = tc[(j - p.ch) * 9 + t]; v2 = tc[(p.Cc + (j - p.ch)) * 9 + t]; }
            } else if (tx) {
                v2 = tx[((p.ch + 4) + p.ch) * 9 + t];
            }
            if (j < cin) d1[e] += v1;
            d2[e] += v2;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    __shared__ double part[4];
    if ((tid & 63) == 0) part[tid >> 6] = a;
    __syncthreads();
    if (tid == 0) {
        const double tot = part[0] + part[1] + part[2] + part[3];
        unsafeAtomicAdd(p.acc + k, tot);
        __threadfence();
        if (atomicAdd(p.cnt + k, 1u) == (unsigned)p.S - 1) {
            __threadfence();
            const double sum = unsafeAtomicAdd(p.acc + k, 0.0);
            const float kp = p.Kp[k];
            p.dK[k] = (kp >= -4.0f && kp <= 1.3862943611198906f) ? (float)sum : 0.f;
        }
    }
}

// dims = {NL, C, ch, Cc, rows per layer in tmpC (0 -> 4)}.  tmpX / tmpC may be null (nothing to scatter from that source).
// ws: >= 4 * NL zero-initialised floats, 8-byte aligned.
extern "C" int tmg_level_finish(const void* Wz, const void* dWz, const void* Bz, const void* dBz, const void* Kp, const void* tmpX,
                                const void* tmpC, void* dW1, void* dW2, void* dK, void* ws, const int64_t* dims, hipStream_t st) {
    LevelFinP p;
    p.Wz = (const float*)Wz; p.dWz = (const float*)dWz; p.Bz = (const float*)Bz; p.dBz = (const float*)dBz; p.Kp = (const float*)Kp;
    p.tmpX = (const float*)tmpX; p.tmpC = (const float*)tmpC; p.dW1 = (float*)dW1; p.dW2 = (float*)dW2; p.dK = (float*)dK;
    p.NL = (int)dims[0]; p.C = (int)dims[1]; p.ch = (int)dims[2]; p.Cc = (int)dims[3];
    p.rowsC = dims[4] == 2 ? 2 : 4;
    if (p.NL < 1 || p.C < 1 || !ws || ((uintptr_t)ws & 7)) return -1;
    p.acc = (double*)ws;
    p.cnt = (unsigned*)((float*)ws + 2 * (size_t)p.NL);
    const size_t nz = (size_t)p.C * (p.ch + p.Cc + 2) * 9;
    int S = (int)((nz / 4 + 256 * 8 - 1) / (256 * 8));
    p.S = S < 1 ? 1 : (S > 32 ? 32 : S);
    hipLaunchKernelGGL(level_finish_kernel, dim3(p.S, p.NL), dim3(256), 0, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}
