// Physics-constrained reverse-KL loss of the TM-Glow trainer as fused gfx950 kernels.
// Replaces TMGLowLoss.forward and the Sobel-type stencil residuals it calls
//   reference tmglow/nn/trainFlowParallel.py:121-177, tmglow/pc/physicsConstrained.py:42-94,
//   tmglow/pc/grad1Filter.py:37-88, tmglow/pc/grad2Filter.py:28-101
// (7 F.conv2d launches + ~30 element-wise launches per call there) by one forward pass, one per-pixel time-statistics
// pass and one backward pass.  Purely HBM-bound: every prediction byte is read once per pass (plus a 1-2 pixel halo).
//
// Layout: predictions / targets are [N = B*T, 3, H, W] planar fp32 (channels u_x, u_y, p), contiguous.
#include "tmg_common.h"

struct PhysP {
    const float* y;       // [N,3,H,W]
    const float* target;  // [N,3,H,W] or null
    int N, H, W;
    float sd[3], mu[3];   // un-normalisation: field = sd*y + mu
    float dx, dy, rho;
    float* sums;          // [3]: sum pstar^2 (interior), sum ustar^2 (interior rows), sum (y-target)^2   (atomics)
    float* pstar_out;     // optional [N,1,H,W]
    float* ustar_out;     // optional [N,1,H,W+2]
};

__device__ __forceinline__ float g1w(int a, int b) {  // Grad1 x-kernel  [[-1,0,1],[-2,0,2],[-1,0,1]] / 8
    const float row = (a == 1) ? 2.f : 1.f;
    return (b == 0 ? -row : (b == 2 ? row : 0.f)) * 0.125f;
}
__device__ __forceinline__ float g2w(int a, int b) {  // Grad2 x-kernel  [[1,-2,1],[2,-4,2],[1,-2,1]] / 4
    const float row = (a == 1) ? 2.f : 1.f;
    return (b == 1 ? -2.f * row : row) * 0.25f;
}

#define PT 16  // tile edge

// ---- forward: residual sums (and optional residual fields) -------------------------------------------------------
__global__ __launch_bounds__(256) void phys_fwd_kernel(PhysP p) {
    __shared__ float f[3][PT + 2][PT + 2];
    __shared__ float red[4];
    const int n = blockIdx.z;
    const int i0 = blockIdx.y * PT, j0 = blockIdx.x * PT;
    const int tid = threadIdx.x;
    const size_t plane = (size_t)p.H * p.W;
    for (int k = tid; k < 3 * (PT + 2) * (PT + 2); k += 256) {
        const int c = k / ((PT + 2) * (PT + 2));
        const int r = k - c * (PT + 2) * (PT + 2);
        const int li = r / (PT + 2), lj = r - li * (PT + 2);
        const int i = i0 - 1 + li, j = j0 - 1 + lj;
        float v = 0.f;   // zero padding applies to the un-normalised field (F.pad of yPredHat)
        if (i >= 0 && i < p.H && j >= 0 && j < p.W) v = p.sd[c] * p.y[((size_t)n * 3 + c) * plane + (size_t)i * p.W + j] + p.mu[c];
        f[c][li][lj] = v;
    }
    __syncthreads();
    const int ti = tid / PT, tj = tid % PT;
    const int i = i0 + ti, j = j0 + tj;
    float sp = 0.f, su = 0.f, sl = 0.f;
    if (i < p.H && j < p.W) {
        float ux_x = 0.f, ux_y = 0.f, uy_x = 0.f, uy_y = 0.f, pxx = 0.f, pyy = 0.f, dux = 0.f, dvy = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float U = f[0][ti + a][tj + b], V = f[1][ti + a][tj + b], P = f[2][ti + a][tj + b];
                ux_x += g1w(a, b) * U; ux_y += g1w(b, a) * U;
                uy_x += g1w(a, b) * V; uy_y += g1w(b, a) * V;
                pxx += g2w(a, b) * P;  pyy += g2w(b, a) * P;
                // divergence: first / last column replicated (physicsConstrained.py:54), rows see the zero padding
                const int jc = min(max(j + b - 1, 0), p.W - 1) - j0 + 1;
                dux += g1w(a, b) * f[0][ti + a][jc];
                dvy += g1w(b, a) * f[1][ti + a][jc];
            }
        ux_x /= p.dx; uy_x /= p.dx; ux_y /= p.dy; uy_y /= p.dy;
        const float raw_p = p.dx * p.dy * ((pxx / (p.dx * p.dx) + pyy / (p.dy * p.dy)) / p.rho + ux_x * ux_x + 2.f * ux_y * uy_x + uy_y * uy_y);
        const float raw_d = p.dx * (dvy / p.dy + dux / p.dx);
        const float ps = fminf(fmaxf(raw_p, -1.f), 1.f), us = fminf(fmaxf(raw_d, -1.f), 1.f);
        if (p.pstar_out) p.pstar_out[(size_t)n * plane + (size_t)i * p.W + j] = ps;
        if (p.ustar_out) p.ustar_out[((size_t)n * p.H + i) * (p.W + 2) + j + 1] = us;
        const bool rin = (i >= 1 && i <= p.H - 2);
        if (rin && j >= 1 && j <= p.W - 2) sp = ps * ps;
        if (rin) su = us * us;
        if (p.target) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t o = ((size_t)n * 3 + c) * plane + (size_t)i * p.W + j;
                const float d = p.y[o] - p.target[o];
                sl += d * d;
            }
        }
    }
    const float a0 = block_sum_256(sp, red);
    const float a1 = block_sum_256(su, red);
    const float a2 = block_sum_256(sl, red);
    if (tid == 0 && p.sums) {
        atomicAdd(p.sums + 0, a0);
        atomicAdd(p.sums + 1, a1);
        atomicAdd(p.sums + 2, a2);
    }
}

// the two outer columns of the width-(W+2) divergence field (only needed by the stand-alone calcDivergence API)
__global__ void phys_div_edge_kernel(PhysP p) {
    const size_t total = (size_t)p.N * p.H * 2;
    const size_t plane = (size_t)p.H * p.W;
    for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < total; k += (size_t)gridDim.x * blockDim.x) {
        const int side = k & 1;
        const int i = (k >> 1) % p.H;
        const int n = (k >> 1) / p.H;
        const int c = side ? p.W + 1 : 0;   // column of the padded array
        float dux = 0.f, dvy = 0.f;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                const int ii = i + a - 1, cc = c + b - 1;
                if (ii < 0 || ii >= p.H || cc < 0 || cc > p.W + 1) continue;
                const int jj = min(max(cc - 1, 0), p.W - 1);
                const float U = p.sd[0] * p.y[((size_t)n * 3 + 0) * plane + (size_t)ii * p.W + jj] + p.mu[0];
                const float V = p.sd[1] * p.y[((size_t)n * 3 + 1) * plane + (size_t)ii * p.W + jj] + p.mu[1];
                dux += g1w(a, b) * U;
                dvy += g1w(b, a) * V;
            }
        const float raw_d = p.dx * (dvy / p.dy + dux / p.dx);
        p.ustar_out[((size_t)n * p.H + i) * (p.W + 2) + c] = fminf(fmaxf(raw_d, -1.f), 1.f);
    }
}

// ---- per-pixel statistics over the T time-steps of a window ------------------------------------------------------
// y: [B,T,3,H,W]; target_rms: [B,3,H,W].  mean_out / coef_out: [B,3,H,W] (saved for backward):
//   coef = (rms - target_rms) / (T * rms)   so that   d/dy_t (rms - target_rms)^2 = 2 * coef * (y_t - mean)
__global__ __launch_bounds__(256) void phys_rms_kernel(const float* __restrict__ y, const float* __restrict__ trms, int B, int T,
                                                       size_t chw, float* __restrict__ mean_out, float* __restrict__ coef_out,
                                                       float* __restrict__ sum_out) {
    __shared__ float red[4];
    const size_t total = (size_t)B * chw;
    float acc = 0.f;
    for (size_t k = blockIdx.x * (size_t)256 + threadIdx.x; k < total; k += (size_t)gridDim.x * 256) {
        const size_t b = k / chw, r = k - b * chw;
        const float* src = y + b * T * chw + r;
        float m = 0.f;
        for (int t = 0; t < T; ++t) m += src[(size_t)t * chw];
        m /= T;
        float v = 0.f;
        for (int t = 0; t < T; ++t) {
            const float d = src[(size_t)t * chw] - m;
            v += d * d;
        }
        const float rms = sqrtf(v / T);
        const float d = rms - trms[k];
        acc += d * d;
        if (mean_out) {
            mean_out[k] = m;
            coef_out[k] = rms > 0.f ? d / (T * rms) : 0.f;
        }
    }
    const float tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) atomicAdd(sum_out, tot);
}

// ---- backward: d loss / d y for all four data terms ---------------------------------------------------------------
struct PhysBP {
    const float* y; const float* target;   // [N,3,H,W]
    const float* mean; const float* coef;  // [B,3,H,W]
    float* dy;                             // [N,3,H,W]
    int N, T, H, W;
    float sd[3], mu[3];
    float dx, dy_, rho;
    float cp, cd, cl, cr;   // upstream * beta * 2 / count of each term (pressure, divergence, L2, rms)
    const float* gup;       // optional DEVICE scalar multiplied onto the four coefficients (the upstream gradient of the loss
                            // value, read here instead of on the host: no device->host synchronisation in the BPTT window)
};

__global__ __launch_bounds__(256) void phys_bwd_kernel(PhysBP p) {
    __shared__ float f[3][PT + 4][PT + 4];   // un-normalised fields, halo 2, zero outside the image
    __shared__ float S[6][PT + 2][PT + 2];   // adjoint sources on tile + halo 1
    const int n = blockIdx.z;
    const float up_ = p.gup ? *p.gup : 1.f;
    const float cp_ = up_ * p.cp, cd_ = up_ * p.cd, cl_ = up_ * p.cl, cr_ = up_ * p.cr;
    const int i0 = blockIdx.y * PT, j0 = blockIdx.x * PT;
    const int tid = threadIdx.x;
    const size_t plane = (size_t)p.H * p.W;
    for (int k = tid; k < 3 * (PT + 4) * (PT + 4); k += 256) {
        const int c = k / ((PT + 4) * (PT + 4));
        const int r = k - c * (PT + 4) * (PT + 4);
        const int li = r / (PT + 4), lj = r - li * (PT + 4);
        const int i = i0 - 2 + li, j = j0 - 2 + lj;
        float v = 0.f;
        if (i >= 0 && i < p.H && j >= 0 && j < p.W) v = p.sd[c] * p.y[((size_t)n * 3 + c) * plane + (size_t)i * p.W + j] + p.mu[c];
        f[c][li][lj] = v;
    }
    __syncthreads();
    for (int k = tid; k < (PT + 2) * (PT + 2); k += 256) {
        const int li = k / (PT + 2), lj = k - li * (PT + 2);
        const int i = i0 - 1 + li, j = j0 - 1 + lj;   // pixel p of the source field
        float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f;
        if (i >= 0 && i < p.H && j >= 0 && j < p.W) {
            float ux_x = 0.f, ux_y = 0.f, uy_x = 0.f, uy_y = 0.f, pxx = 0.f, pyy = 0.f, dux = 0.f, dvy = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float U = f[0][li + a][lj + b], V = f[1][li + a][lj + b], P = f[2][li + a][lj + b];
                    ux_x += g1w(a, b) * U; ux_y += g1w(b, a) * U;
                    uy_x += g1w(a, b) * V; uy_y += g1w(b, a) * V;
                    pxx += g2w(a, b) * P;  pyy += g2w(b, a) * P;
                    const int jc = min(max(j + b - 1, 0), p.W - 1) - j0 + 2;
                    dux += g1w(a, b) * f[0][li + a][jc];
                    dvy += g1w(b, a) * f[1][li + a][jc];
                }
            ux_x /= p.dx; uy_x /= p.dx; ux_y /= p.dy_; uy_y /= p.dy_;
            const float raw_p = p.dx * p.dy_ * ((pxx / (p.dx * p.dx) + pyy / (p.dy_ * p.dy_)) / p.rho + ux_x * ux_x + 2.f * ux_y * uy_x + uy_y * uy_y);
            const float raw_d = p.dx * (dvy / p.dy_ + dux / p.dx);
            const bool rin = (i >= 1 && i <= p.H - 2);
            if (rin && j >= 1 && j <= p.W - 2 && raw_p >= -1.f && raw_p <= 1.f) {
                const float a_p = cp_ * raw_p * p.dx * p.dy_;
                s1 = a_p * 2.f * ux_x; s2 = a_p * 2.f * uy_x; s3 = a_p * 2.f * ux_y; s4 = a_p * 2.f * uy_y; s5 = a_p / p.rho;
            }
            if (rin && raw_d >= -1.f && raw_d <= 1.f) s6 = cd_ * raw_d * p.dx;
        }
        S[0][li][lj] = s1; S[1][li][lj] = s2; S[2][li][lj] = s3; S[3][li][lj] = s4; S[4][li][lj] = s5; S[5][li][lj] = s6;
    }
    __syncthreads();
    const int ti = tid / PT, tj = tid % PT;
    const int i = i0 + ti, j = j0 + tj;
    if (i >= p.H || j >= p.W) return;
    float gU = 0.f, gV = 0.f, gP = 0.f;
#pragma unroll
    for (int di = -1; di <= 1; ++di)
#pragma unroll
        for (int dj = -1; dj <= 1; ++dj) {
            // source pixel p = q + (di, dj) used q through tap (a, b) = (1 - di, 1 - dj)
            const int a = 1 - di, b = 1 - dj;
            const int li = ti + 1 + di, lj = tj + 1 + dj;
            gU += S[0][li][lj] * g1w(a, b) / p.dx + S[1][li][lj] * g1w(b, a) / p.dy_;
            gV += S[2][li][lj] * g1w(a, b) / p.dx + S[3][li][lj] * g1w(b, a) / p.dy_;
            gP += S[4][li][lj] * (g2w(a, b) / (p.dx * p.dx) + g2w(b, a) / (p.dy_ * p.dy_));
            // divergence: p's tap column pj + bb - 1 is clamped into the image; it reaches q when the clamp lands on j
            const int pj = j + dj;
            if (pj >= 0 && pj < p.W) {
                const float s6 = S[5][li][lj];
#pragma unroll
                for (int bb = 0; bb < 3; ++bb)
                    if (min(max(pj + bb - 1, 0), p.W - 1) == j) {
                        gU += s6 * g1w(a, bb) / p.dx;
                        gV += s6 * g1w(bb, a) / p.dy_;
                    }
            }
        }
    const float g[3] = {gU * p.sd[0], gV * p.sd[1], gP * p.sd[2]};
    const int b_ = n / p.T;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t o = ((size_t)n * 3 + c) * plane + (size_t)i * p.W + j;
        const size_t ob = ((size_t)b_ * 3 + c) * plane + (size_t)i * p.W + j;
        const float yv = p.y[o];
        float v = g[c] + cl_ * (yv - p.target[o]);
        if (p.coef) v += cr_ * p.coef[ob] * (yv - p.mean[ob]);
        p.dy[o] = v;
    }
}

// ---- C ABI -----------------------------------------------------------------------------------------------------------
// dims = {N, H, W}; fl = {sd0,sd1,sd2, mu0,mu1,mu2, dx, dy, rho}
extern "C" int tmg_phys_fwd(const void* y, const void* target, void* sums, void* pstar_out, void* ustar_out, const int64_t* dims,
                            const float* fl, hipStream_t st) {
    PhysP p;
    p.y = (const float*)y; p.target = (const float*)target; p.sums = (float*)sums;
    p.pstar_out = (float*)pstar_out; p.ustar_out = (float*)ustar_out;
    p.N = (int)dims[0]; p.H = (int)dims[1]; p.W = (int)dims[2];
    for (int c = 0; c < 3; ++c) { p.sd[c] = fl[c]; p.mu[c] = fl[3 + c]; }
    p.dx = fl[6]; p.dy = fl[7]; p.rho = fl[8];
    dim3 grid((p.W + PT - 1) / PT, (p.H + PT - 1) / PT, p.N);
    hipLaunchKernelGGL(phys_fwd_kernel, grid, dim3(256), 0, st, p);
    TMG_CHECK_LAUNCH();
    if (p.ustar_out) {
        const size_t total = (size_t)p.N * p.H * 2;
        hipLaunchKernelGGL(phys_div_edge_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, st, p);
        TMG_CHECK_LAUNCH();
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The two residual fields for ANY stencil pair of the reference (3x3 or 5x5 first- and second-derivative stencils,
// pc/grad1Filter.py:37-57, pc/grad2Filter.py:28-49) and with or without the cell-size scaling (physicsConstrained.py:58-59, :90-91):
// the stand-alone API of PhysConstrainedLES.  The trainer's loss uses the 3x3 / scaled form through phys_fwd_kernel; this one is a
// plain one-thread-per-pixel kernel over global memory (fields of a few hundred KB).
//   ustar [N][H][W + 2] = clamp(sd * (d/dy v + d/dx u)) on the field with its first / last column replicated (:54), zero padding beyond
//   pstar [N][H][W]     = clamp(sp * ((p_xx + p_yy) / rho + u_x^2 + 2 u_y v_x + v_y^2)), zero padding
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float g1w5(int a, int b) {   // d/dx, 5x5 (rows a, columns b) before the / (9 * 12)
    const float col[5] = {1.f, -8.f, 0.f, 8.f, -1.f};
    const float row[5] = {1.f, 2.f, 3.f, 2.f, 1.f};
    return row[a] * col[b];
}
__device__ __forceinline__ float g2w5(int a, int b) {   // d2/dx2, 5x5: the reference's last column is -1 in EVERY row (grad2Filter.py:33-37)
    const float col[5] = {-1.f, 16.f, -30.f, 16.f, -1.f};
    const float row[5] = {1.f, 2.f, 3.f, 2.f, 1.f};
    return b == 4 ? -1.f : row[a] * col[b];
}

template <int K1, int K2>
__global__ void phys_fields_kernel(const float* __restrict__ u, const float* __restrict__ pr, float* __restrict__ ustar,
                                   float* __restrict__ pstar, int N, int H, int W, float dx, float dy, float rho, float sd, float sp) {
    constexpr int R1 = K1 / 2, R2 = K2 / 2;
    const size_t plane = (size_t)H * W;
    const size_t total = (size_t)N * H * (W + 2);
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int jw = (int)(idx % (W + 2));
        size_t r = idx / (W + 2);
        const int i = (int)(r % H);
        const int n = (int)(r / H);
        const float* U = u + (size_t)n * 2 * plane;
        const float* V = U + plane;
        auto w1 = [&](int a, int b) { return K1 == 3 ? g1w(a, b) : g1w5(a, b) / 108.f; };
        auto w2 = [&](int a, int b) { return K2 == 3 ? g2w(a, b) : g2w5(a, b) / 108.f; };
        if (ustar) {
            // widened field: column jw of [0, W + 2) reads column clamp(jw - 1) of the original; zero outside the widened field
            float dux = 0.f, dvy = 0.f;
#pragma unroll
            for (int a = 0; a < K1; ++a)
#pragma unroll
                for (int b = 0; b < K1; ++b) {
                    const int ii = i + a - R1, jj = jw + b - R1;
                    if (ii < 0 || ii >= H || jj < 0 || jj >= W + 2) continue;
                    const int jc = min(max(jj - 1, 0), W - 1);
                    dux += w1(a, b) * U[(size_t)ii * W + jc];
                    dvy += w1(b, a) * V[(size_t)ii * W + jc];
                }
            const float raw = sd * (dvy / dy + dux / dx);
            ustar[idx] = fminf(fmaxf(raw, -1.f), 1.f);
        }
        if (pstar && jw < W) {
            const int j = jw;
            const float* P = pr + (size_t)n * plane;
            float ux_x = 0.f, ux_y = 0.f, uy_x = 0.f, uy_y = 0.f, pxx = 0.f, pyy = 0.f;
#pragma unroll
            for (int a = 0; a < K1; ++a)
#pragma unroll
                for (int b = 0; b < K1; ++b) {
                    const int ii = i + a - R1, jj = j + b - R1;
                    if (ii < 0 || ii >= H || jj < 0 || jj >= W) continue;
                    const float uu = U[(size_t)ii * W + jj], vv = V[(size_t)ii * W + jj];
                    ux_x += w1(a, b) * uu; ux_y += w1(b, a) * uu;
                    uy_x += w1(a, b) * vv; uy_y += w1(b, a) * vv;
                }
#pragma unroll
            for (int a = 0; a < K2; ++a)
#pragma unroll
                for (int b = 0; b < K2; ++b) {
                    const int ii = i + a - R2, jj = j + b - R2;
                    if (ii < 0 || ii >= H || jj < 0 || jj >= W) continue;
                    const float pp = P[(size_t)ii * W + jj];
                    pxx += w2(a, b) * pp; pyy += w2(b, a) * pp;
                }
            ux_x /= dx; uy_x /= dx; ux_y /= dy; uy_y /= dy;
            const float raw = sp * ((pxx / (dx * dx) + pyy / (dy * dy)) / rho + ux_x * ux_x + 2.f * ux_y * uy_x + uy_y * uy_y);
            pstar[((size_t)n * H + i) * W + j] = fminf(fmaxf(raw, -1.f), 1.f);
        }
    }
}

// u: [N][2][H][W] planar velocity, p: [N][1][H][W] pressure (may be null when pstar is null); ustar [N][1][H][W + 2] / pstar [N][1][H][W]
// (either may be null).  dims = {N, H, W, k1, k2, scale}; k1, k2 in {3, 5} (else -100, nothing launched: the reference raises
// ValueError for other sizes); fl = {dx, dy, rho}.
extern "C" int tmg_phys_fields(const void* u, const void* p, void* ustar, void* pstar, const int64_t* dims, const float* fl, hipStream_t st) {
    const int N = (int)dims[0], H = (int)dims[1], W = (int)dims[2], k1 = (int)dims[3], k2 = (int)dims[4], scale = (int)dims[5];
    if ((k1 != 3 && k1 != 5) || (k2 != 3 && k2 != 5)) return -100;
    if (pstar && !p) return -3;
    const float dx = fl[0], dy = fl[1], rho = fl[2];
    const float sd = scale ? dx : 1.f, sp = scale ? dx * dy : 1.f;
    const size_t total = (size_t)N * H * (W + 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
#define TMG_PF(K1_, K2_) hipLaunchKernelGGL((phys_fields_kernel<K1_, K2_>), dim3(blocks), dim3(256), 0, st, (const float*)u, (const float*)p, (float*)ustar, (float*)pstar, N, H, W, dx, dy, rho, sd, sp)
    if (k1 == 3 && k2 == 3) TMG_PF(3, 3);
    else if (k1 == 3) TMG_PF(3, 5);
    else if (k2 == 3) TMG_PF(5, 3);
    else TMG_PF(5, 5);
#undef TMG_PF
    TMG_CHECK_LAUNCH();
    return 0;
}

// dims = {B, T, 3*H*W}
extern "C" int tmg_phys_rms(const void* y, const void* trms, void* mean_out, void* coef_out, void* sum_out, const int64_t* dims,
                            hipStream_t st) {
    const size_t total = (size_t)dims[0] * dims[2];
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(phys_rms_kernel, dim3(blocks), dim3(256), 0, st, (const float*)y, (const float*)trms, (int)dims[0], (int)dims[1],
                       (size_t)dims[2], (float*)mean_out, (float*)coef_out, (float*)sum_out);
    TMG_CHECK_LAUNCH();
    return 0;
}

// dims = {N, T, H, W}; fl = {sd0..2, mu0..2, dx, dy, rho, cp, cd, cl, cr}
extern "C" int tmg_phys_bwd_dev(const void* y, const void* target, const void* mean, const void* coef, void* dy, const void* upstream,
                                const int64_t* dims, const float* fl, hipStream_t st);
extern "C" int tmg_phys_bwd(const void* y, const void* target, const void* mean, const void* coef, void* dy, const int64_t* dims,
                            const float* fl, hipStream_t st) {
    return tmg_phys_bwd_dev(y, target, mean, coef, dy, nullptr, dims, fl, st);
}

// as tmg_phys_bwd with the four coefficients additionally multiplied by the device scalar *upstream (NULL: 1)
extern "C" int tmg_phys_bwd_dev(const void* y, const void* target, const void* mean, const void* coef, void* dy, const void* upstream,
                                const int64_t* dims, const float* fl, hipStream_t st) {
    PhysBP p;
    p.gup = (const float*)upstream;
    p.y = (const float*)y; p.target = (const float*)target; p.mean = (const float*)mean; p.coef = (const float*)coef; p.dy = (float*)dy;
    p.N = (int)dims[0]; p.T = (int)dims[1]; p.H = (int)dims[2]; p.W = (int)dims[3];
    for (int c = 0; c < 3; ++c) { p.sd[c] = fl[c]; p.mu[c] = fl[3 + c]; }
    p.dx = fl[6]; p.dy_ = fl[7]; p.rho = fl[8]; p.cp = fl[9]; p.cd = fl[10]; p.cl = fl[11]; p.cr = fl[12];
    dim3 grid((p.W + PT - 1) / PT, (p.H + PT - 1) / PT, p.N);
    hipLaunchKernelGGL(phys_bwd_kernel, grid, dim3(256), 0, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================================================
// Adam / AMSGrad update of ALL parameters in one launch (the trainer's optimizer step, reference main.py:78: Adam, weight decay 1e-8,
// amsgrad): torch's multi-tensor implementations walk ~1 000 small tensors in ~100 launches (1.4 ms per step at config M); the update
// itself moves 36 bytes per parameter (190 MB: ~40 us).  tab: device int64 [n][5] = pointers (param, grad, exp_avg, exp_avg_sq,
// max_exp_avg_sq); chunks: device int32 [nchunks][3] = (tensor, first element, elements) covering every tensor in pieces of <= 4096.
// Same arithmetic and operation order as torch.optim.Adam (single-tensor form):
//   g += wd p;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;  vmax = max(vmax, v);  p -= (lr / bc1) m / (sqrt(vmax) / sqrt(bc2) + eps)
// =================================================================================================================================
__global__ __launch_bounds__(256) void adam_step_kernel(const long long* __restrict__ tab, const int* __restrict__ chunks, int nchunks,
                                                        float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s, float omb1,
                                                        float omb2, int amsgrad) {
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int t = chunks[3 * c], e0 = chunks[3 * c + 1], n = chunks[3 * c + 2];
        float* p = reinterpret_cast<float*>(tab[5 * t]) + e0;
        const float* g = reinterpret_cast<const float*>(tab[5 * t + 1]) + e0;
        float* m = reinterpret_cast<float*>(tab[5 * t + 2]) + e0;
        float* v = reinterpret_cast<float*>(tab[5 * t + 3]) + e0;
        float* vm = reinterpret_cast<float*>(tab[5 * t + 4]) + e0;
        const float step_size = lr / bc1;
        for (int i = threadIdx.x; i < n; i += 256) {
            float gi = g[i];
            const float pi = p[i];
            if (wd != 0.f) gi = fmaf(wd, pi, gi);
            const float mi = m[i] + (gi - m[i]) * omb1;                // torch: exp_avg.lerp_(grad, 1 - beta1); 1 - beta formed in double on the host
            const float vi = b2 * v[i] + omb2 * (gi * gi);             // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
            m[i] = mi; v[i] = vi;
            float den;
            if (amsgrad) {
                const float vx = fmaxf(vm[i], vi);
                vm[i] = vx;
                den = sqrtf(vx) / bc2s + eps;
            } else {
                den = sqrtf(vi) / bc2s + eps;
            }
            p[i] = pi - step_size * (mi / den);
        }
    }
}

// fl = {lr, beta1, beta2, eps, weight_decay, bias_correction1, sqrt(bias_correction2), 1 - beta1, 1 - beta2}; dims = {nchunks, amsgrad}
extern "C" int tmg_adam_step(const void* tab, const void* chunks, const int64_t* dims, const float* fl, hipStream_t st) {
    const int nchunks = (int)dims[0];
    if (nchunks <= 0) return 0;
    const int grid = nchunks < 4096 ? nchunks : 4096;
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid), dim3(256), 0, st, (const long long*)tab, (const int*)chunks, nchunks, fl[0], fl[1], fl[2],
                       fl[3], fl[4], fl[5], fl[6], fl[7], fl[8], (int)dims[1]);
    TMG_CHECK_LAUNCH();
    return 0;
}
