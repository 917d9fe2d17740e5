// Grouped 3x3 weight gradient for FOUR output channels per group: the growth-1 layers of the coupling networks
// (denseBlock.py:18-36 under autograd), all NL layers of a flow level in one launch.
//
//   dW[g][co][ci][ky][kx] += sum_{b,y,x} dy[b,y,x, 4 g + co] * act(X_g[b, y + ky - 1, x + kx - 1, ci]),   zero padding, co < 4
//
// With 4 output channels a 16x16x4 MFMA tile is 2/16 (the two real growth channels) to 4/16 used: the general grouped kernel
// (conv_wgrad_kernel<9,1>) spends 72 matrix-pipe cycles per pixel and 16 input channels there.  v_mfma_f32_4x4x1_16B_f32 is sixteen
// independent 4x4 outer products per instruction at the same flop rate (lane l: block l / 4; A = row l % 4, B = column l % 4;
// D[l][r] = A[4 (l / 4) + r] * B[l], measured with tools/micro/mfma4x4.hip) - a block here is a (tap, input-channel quad)
// pair, its 4 columns are the group's 4 dy channels, and one instruction consumes ONE pixel: 8 cycles per pixel and 16 blocks, i.e.
// 9 Cin / 64 instructions per pixel instead of 9 Cin / 16 / 4 four times as long ones (4.5x fewer pipe cycles).
//
// Layout: a block of 4 waves stages the ReLU'd input patch (TH + 2) x 18 pixels x Cin channels of one 16-wide tile in LDS (NHWC, pixel
// stride Cin words - Cin / 4 is odd for every supported width - and a row pitch of 19 pixels: with these the 16 blocks' 4-word reads of
// one ds_read_b32 fall on distinct banks, checked exhaustively on the host when the table below was chosen) plus the tile's dy quads.
// A wave walks the rows r = wave, wave + 4, ..; per pixel one broadcast read of dy[l % 4] and, per slot (16 blocks), one read of
// x[tap, quad, l % 4] and one MFMA.  A block keeps its SL x 4 accumulator registers over all its tiles (grid = partitions x groups)
// and adds them to dW once at the end (the four waves meet in LDS first).
#include "tmg_common.h"

struct ThinP {
    const long long* gtab;   // [G][16]: 3 input segments {pointer, pixel stride, 0, channels} + {dy pointer or 0, ...}   (tmg_conv_wgrad_grouped)
    const float* dy;         // shared upstream gradient: group g at channels [dyc g, dyc g + dyc)
    int dys, dyc;            // pixel stride; channels per group: 4, or 2 (compact: rows 2, 3 of a group's dW stay zero)
    float* dW;               // [G][4][Cin][3][3], accumulated into
    int B, H, W, relu_in;
    int tiles_x, tiles_y, ntiles;   // per group
    int G, P;                       // groups, tile partitions per group (grid = G * P blocks)
};

// SL slots of 16 blocks cover the 9 * Cin / 4 (tap, channel quad) blocks; CS = Cin; TH tile rows
template <int SL, int CS, int TH>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(ThinP p) {
    constexpr int Q = CS / 4, NB = 9 * Q;
    constexpr int PWp = 19, PH = TH + 2, PW = 18;
    constexpr int XW = PH * PWp * CS;                  // words of the input patch
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* X = lds;
    float* DY = lds + XW;                              // [TH * 16][4]
    static_assert(SL * 16 >= NB, "slots");
    static_assert(SL * 256 * 4 <= XW + TH * 64, "the cross-wave reduction reuses the staging area");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (partition, group): the G blocks of a partition walk the same tiles at the same time and share the dy lines (a group
    // uses 16 bytes of every pixel of the shared dy tensor) - keep them on ONE XCD (consecutive block ids go round the 8 XCDs), so the
    // lines are fetched into one L2 once instead of G times into all of them
    int g, part;
    if ((p.P & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        g = slot % p.G;
        part = xcd + 8 * (slot / p.G);
    } else {
        g = blockIdx.x % p.G;
        part = blockIdx.x / p.G;
    }
    const long long* gt = p.gtab + (size_t)g * 16;
    const float* sp0 = reinterpret_cast<const float*>(gt[0]);
    const float* sp1 = reinterpret_cast<const float*>(gt[4]);
    const float* sp2 = reinterpret_cast<const float*>(gt[8]);
    const int ss0 = (int)gt[1], ss1 = (int)gt[5], ss2 = (int)gt[9];
    const int n0 = (int)gt[3], n1 = (int)gt[7];
    const float* dyb = p.dy + p.dyc * g;

    // lane offsets of the A operand inside the patch, per slot: block beta = 16 t + lane / 4 = tap * Q + quad (blocks past NB repeat
    // the last one; their results are dropped)
    int aoff[SL];
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int beta = min(16 * t + (lane >> 2), NB - 1);
        const int u = beta / Q, s = beta - u * Q;
        const int ky = u / 3, kx = u - ky * 3;
        aoff[t] = (ky * PWp + kx) * CS + 4 * s + (lane & 3);
    }
    f32x4 acc[SL];
#pragma unroll
    for (int t = 0; t < SL; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int tile = part; tile < p.ntiles; tile += p.P) {
        int t_ = tile;
        const int tx = t_ % p.tiles_x; t_ /= p.tiles_x;
        const int ty = t_ % p.tiles_y;
        const int b = t_ / p.tiles_y;
        const int oy0 = ty * TH, ox0 = tx * 16;
        const size_t img = (size_t)b * p.H * p.W;
        __syncthreads();   // the previous tile's readers are done
        for (int it = tid; it < PH * PW * Q; it += 256) {
            const int pp = it / Q, qd = it - pp * Q;
            const int row = pp / PW, col = pp - row * PW;
            const int iy = oy0 - 1 + row, ix = ox0 - 1 + col;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                const size_t pix = img + (size_t)iy * p.W + ix;
                int c = 4 * qd;
                const float* a_;
                if (c < n0) a_ = sp0 + pix * ss0 + c;
                else if (c < n0 + n1) a_ = sp1 + pix * ss1 + (c - n0);
                else a_ = sp2 + pix * ss2 + (c - n0 - n1);
                v = tmg_ldg4(a_);
                if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            *reinterpret_cast<float4*>(X + (row * PWp + col) * CS + 4 * qd) = v;
        }
        for (int it = tid; it < TH * 16; it += 256) {
            const int r = it >> 4, c = it & 15;
            const int oy = oy0 + r, ox = ox0 + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (oy < p.H && ox < p.W) {
                const float* a_ = dyb + (img + (size_t)oy * p.W + ox) * p.dys;
                if (p.dyc == 4) v = *reinterpret_cast<const float4*>(a_);
                else { const float2 v2 = *reinterpret_cast<const float2*>(a_); v = make_float4(v2.x, v2.y, 0.f, 0.f); }
            }
            *reinterpret_cast<float4*>(DY + it * 4) = v;
        }
        __syncthreads();
        for (int r = wave; r < TH; r += 4) {
            const float* xr = X + r * PWp * CS;
            const float* dr = DY + r * 64 + (lane & 3);
#pragma unroll
            for (int px = 0; px < 16; ++px) {
                const float bv = dr[px * 4];
#pragma unroll
                for (int t = 0; t < SL; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(xr[aoff[t] + px * CS], bv, acc[t], 0, 0, 0);
            }
        }
    }
    // ---- the four waves' partial sums meet in LDS, then one atomic per (block, element) ------------------------------------------
    __syncthreads();
#pragma unroll
    for (int t = 0; t < SL; ++t) *reinterpret_cast<f32x4*>(lds + ((wave * SL + t) * 64 + lane) * 4) = acc[t];
    __syncthreads();
    float* dWg = p.dW + (size_t)g * 4 * CS * 9;
    for (int e = tid; e < SL * 256; e += 256) {
        const int t = e >> 8, l = (e >> 2) & 63, r = e & 3;
        const int beta = 16 * t + (l >> 2);
        if (beta < NB) {
            const int idx = (t * 64 + l) * 4 + r;
            const float v = lds[idx] + lds[idx + SL * 256] + lds[idx + 2 * SL * 256] + lds[idx + 3 * SL * 256];
            const int u = beta / Q, s = beta - u * Q;
            unsafeAtomicAdd(dWg + ((size_t)(l & 3) * CS + 4 * s + r) * 9 + u, v);
        }
    }
}

template <int SL, int CS, int TH>
static int launch_thin(ThinP p, int G, hipStream_t st) {
    constexpr int lds_bytes = ((TH + 2) * 19 * CS + TH * 64) * 4;
    p.tiles_x = (p.W + 15) / 16;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    // enough blocks to fill every CU to its LDS-limited occupancy, few enough that the final atomics stay negligible
    int per_cu = 160 * 1024 / lds_bytes;
    per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
    int P = (per_cu * 256 + G - 1) / G;
    P = P > p.ntiles ? p.ntiles : (P < 1 ? 1 : P);
    if (P >= 8) P &= ~7;
    p.G = G; p.P = P;
    static_assert(lds_bytes <= 64 * 1024, "above 64 KB the kernel would need the per-device LDS opt-in (TMG_LDS_OPTIN)");
    hipLaunchKernelGGL((wgrad_thin_kernel<SL, CS, TH>), dim3(P * G), dim3(256), lds_bytes, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

// gtab: device int64 [G][16] as for tmg_conv_wgrad_grouped (input segments of every group; the dy entries are not used: dy is the
// shared tensor, group g at channels [4 g, 4 g + 4)).  dims = {B, H, W, Cin, relu_in}.  dW [G][4][Cin][3][3] is ACCUMULATED into
// (float atomics: zero it first).  Supported: Cin in {12, 20, 36, 68} (channel halves 8 / 16 / 32 / 64 plus the 4 growth channels),
// every segment a multiple of 4 channels with 16-byte aligned pixels; otherwise -100 (nothing launched; use tmg_conv_wgrad_grouped).
extern "C" int tmg_conv_wgrad_thin_grouped(const void* gtab, int64_t G, const int64_t* seg_channels, int64_t nseg, const void* dy,
                                           int64_t dy_stride, void* dW, const int64_t* dims, hipStream_t st) {
    ThinP p;
    p.gtab = (const long long*)gtab; p.dy = (const float*)dy; p.dys = (int)dy_stride; p.dW = (float*)dW;
    p.B = (int)dims[0]; p.H = (int)dims[1]; p.W = (int)dims[2]; p.relu_in = (int)dims[4];
    p.dyc = dims[5] == 2 ? 2 : 4;      // dims[5]: dy channels per group (0 / 4: a float4 per pixel, 2: a float2)
    const int Cin = (int)dims[3];
    if (G < 1 || !gtab || nseg < 1 || nseg > 3 || (dy_stride & (p.dyc - 1)) || ((uintptr_t)dy & (4 * p.dyc - 1))) return -100;
    for (int i = 0; i < nseg; ++i)
        if (seg_channels[i] & 3) return -100;
    switch (Cin) {
        case 12: return launch_thin<2, 12, 16>(p, (int)G, st);
        case 20: return launch_thin<3, 20, 16>(p, (int)G, st);
        case 36: return launch_thin<6, 36, 16>(p, (int)G, st);
        case 68: return launch_thin<10, 68, 8>(p, (int)G, st);
        default: return -100;
    }
}

// =================================================================================================================================
// Grouped weight gradient of the 1x1 channel mixes (ActNorm folded into the invertible 1x1 conv, glowConv.py:193-222 under autograd),
// all layers of a level in one launch:   dW[g][o][i] += sum_px dout_g[px][o] * y_g[px][i],   db[g][o] += sum_px dout_g[px][o].
// A streaming GEMM with the pixels as the contraction index: both operands are read straight from global memory in MFMA fragment
// order - lane (c = l % 16, k = l / 16) loads channel c of pixel p + k, a load instruction covers four whole 64-byte pixels - no LDS
// staging, no packing; one v_mfma_f32_16x16x4_f32 per 16x16 channel tile and 4 pixels.  The general grouped kernel ran this shape
// through its 3x3 machinery (patch staging, tap pairs) at 2.2-2.9 TB/s of the 128 B per pixel and layer it has to read.
struct MixWgP {
    const long long* gtab;   // [G][16]: input segments (the mix input y = (x1 | y2) as <= 3 channel segments) + {dout pointer, pixel stride}
    float* dW;               // [G][C][C]
    float* db;               // [G][C] or null
    long long npix;
    int C, G, P;
};

template <int CT>
__global__ __launch_bounds__(256) void mix_wgrad_kernel(MixWgP p) {
    constexpr int U = CT == 2 ? 4 : 8;     // pixel groups (of 4) in flight per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int g = blockIdx.x % p.G, part = blockIdx.x / p.G;
    const long long* gt = p.gtab + (size_t)g * 16;
    const int n0 = (int)gt[3], n1 = (int)gt[7];
    // upstream gradient: channel halves addressed separately (row entries 12 / 13: half 1, 14 / 15: half 2; a null second pointer =
    // one interleaved tensor, half 2 starting C / 2 channels in)
    const float* ap[CT];
    long long as_[CT];
    {
        const float* dy1 = reinterpret_cast<const float*>(gt[12]);
        const float* dy2 = reinterpret_cast<const float*>(gt[14]);
        const int chh = p.C >> 1;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int c = 16 * t + li;
            const bool two = dy2 != nullptr && c >= chh;
            ap[t] = two ? dy2 + (c - chh) : dy1 + c;
            as_[t] = two ? gt[15] : gt[13];
        }
    }
    // per-lane source of the B operand (mix input channel 16 nt + li): segment pointer and pixel stride
    const float* bp[CT];
    long long bs[CT];
#pragma unroll
    for (int nt = 0; nt < CT; ++nt) {
        const int c = 16 * nt + li;
        const int s = c < n0 ? 0 : (c < n0 + n1 ? 1 : 2);
        const int cl = c - (s == 0 ? 0 : (s == 1 ? n0 : n0 + n1));
        bp[nt] = reinterpret_cast<const float*>(gt[4 * s]) + cl;
        bs[nt] = gt[4 * s + 1];
    }
    f32x4 acc[CT][CT];
#pragma unroll
    for (int mt = 0; mt < CT; ++mt)
#pragma unroll
        for (int nt = 0; nt < CT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dbs[CT];
#pragma unroll
    for (int mt = 0; mt < CT; ++mt) dbs[mt] = 0.f;

    // partition `part` owns a contiguous pixel range (a multiple of 16 U pixels); its 4 waves interleave units of 4 U pixels
    const long long per = ((p.npix + p.P - 1) / p.P + 16 * U - 1) / (16 * U) * (16 * U);
    const long long p0 = (long long)part * per, p1 = min(p0 + per, p.npix);
    for (long long pb = p0 + (long long)wave * 4 * U; pb < p1; pb += 16 * U) {
        float a[U][CT], b[U][CT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long px = pb + 4 * u + q;
            const bool ok = px < p1;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                a[u][t] = tmg_ldg1(ok ? ap[t] + px * as_[t] : tmg_zero_page);
                b[u][t] = tmg_ldg1(ok ? bp[t] + px * bs[t] : tmg_zero_page);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mt = 0; mt < CT; ++mt) {
                dbs[mt] += a[u][mt];
#pragma unroll
                for (int nt = 0; nt < CT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][mt], b[u][nt], acc[mt][nt], 0, 0, 0);
            }
    }
    // ---- four waves meet in LDS; D rows = dout channel 16 mt + 4 q + r, column = input channel 16 nt + li -------------------------
    __shared__ float red[4][CT * CT * 256 + CT * 64];
    float* mine = red[wave];
#pragma unroll
    for (int mt = 0; mt < CT; ++mt)
#pragma unroll
        for (int nt = 0; nt < CT; ++nt) *reinterpret_cast<f32x4*>(mine + ((mt * CT + nt) * 64 + lane) * 4) = acc[mt][nt];
#pragma unroll
    for (int mt = 0; mt < CT; ++mt) mine[CT * CT * 256 + mt * 64 + lane] = dbs[mt];
    __syncthreads();
    float* dWg = p.dW + (size_t)g * p.C * p.C;
    for (int e = tid; e < CT * CT * 256; e += 256) {
        const int t = e >> 8, l = (e >> 2) & 63, r = e & 3;
        const int mt = t / CT, nt = t - mt * CT;
        const float v = red[0][e] + red[1][e] + red[2][e] + red[3][e];
        unsafeAtomicAdd(dWg + (size_t)(16 * mt + 4 * (l >> 4) + r) * p.C + 16 * nt + (l & 15), v);
    }
    if (p.db) {
        for (int e = tid; e < CT * 16; e += 256) {
            const int mt = e >> 4, c = e & 15;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int k = 0; k < 4; ++k) v += red[w][CT * CT * 256 + mt * 64 + 16 * k + c];
            unsafeAtomicAdd(p.db + (size_t)g * p.C + 16 * mt + c, v);
        }
    }
}

// gtab as for tmg_conv_wgrad_grouped with every group's own upstream gradient (row entries 12 / 13: pointer, pixel stride); all
// tensors pixel-linear NHWC (address = pointer + pixel * stride + channel).  dims = {npix, C}.  dW [G][C][C] and db [G][C] (nullable) are
// ACCUMULATED into (zero them first).  C in {16, 32} (the levels whose mixes are bandwidth kernels); otherwise -100 (nothing launched).
extern "C" int tmg_mix_wgrad_grouped(const void* gtab, int64_t G, void* dW, void* db, const int64_t* dims, hipStream_t st) {
    MixWgP p;
    p.gtab = (const long long*)gtab; p.dW = (float*)dW; p.db = (float*)db; p.npix = dims[0]; p.C = (int)dims[1]; p.G = (int)G;
    if (G < 1 || !gtab || p.npix < 1) return -100;
    if (p.C != 16 && p.C != 32) return -100;
    // ~8 blocks per CU in total, each with at least a few hundred pixels
    long long P = (2048 + G - 1) / G;
    const long long maxp = (p.npix + 511) / 512;
    if (P > maxp) P = maxp;
    if (P < 1) P = 1;
    p.P = (int)P;
    if (p.C == 16) hipLaunchKernelGGL(mix_wgrad_kernel<1>, dim3((unsigned)(P * G)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(mix_wgrad_kernel<2>, dim3((unsigned)(P * G)), dim3(256), 0, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================================================
// [npix][2 K] -> [K][npix][2]: the level-wide conditioning addends of the growth-1 convs (one conv over cond for all layers of a level,
// channel 2k / 2k+1 = layer k) re-laid as one pixel-contiguous float2 plane per layer.  Every layer's c1x2_fwd launch reads its two
// addends for every pixel: out of the [npix][2K] tensor that is a full 128-byte line per pixel for 8 useful bytes (134 MB per launch at
// 128 x 128 x 64, more than the kernel's real input), out of its plane 8 bytes.  64 pixels per block through LDS, both sides coalesced.
__global__ __launch_bounds__(256) void layer_planes_kernel(const float* __restrict__ src, float* __restrict__ dst, long long npix, int CP) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int RS = CP + 2;                      // row stride (words): float2 reads of one layer from consecutive pixels stay 2-way at worst
    const long long p0 = (long long)blockIdx.x * 64;
    const int q4 = CP >> 2;
    for (int it = threadIdx.x; it < 64 * q4; it += 256) {
        const int px = it / q4, c4 = it - px * q4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p0 + px < npix) v = *reinterpret_cast<const float4*>(src + (p0 + px) * CP + 4 * c4);
        float* d = lds + px * RS + 4 * c4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int K = CP >> 1;
    for (int it = threadIdx.x; it < 64 * K; it += 256) {
        const int k = it >> 6, px = it & 63;
        if (p0 + px < npix) {
            const float2 v = *reinterpret_cast<const float2*>(lds + px * RS + 2 * k);
            *reinterpret_cast<float2*>(dst + ((size_t)k * npix + p0 + px) * 2) = v;
        }
    }
}

// src [npix][CP] contiguous (CP a multiple of 4), dst [CP / 2][npix][2].
extern "C" int tmg_layer_planes(const void* src, void* dst, int64_t npix, int64_t CP, hipStream_t st) {
    if (npix < 1 || CP < 4 || (CP & 3) || CP > 512) return -1;
    const long long blocks = (npix + 63) / 64;
    hipLaunchKernelGGL(layer_planes_kernel, dim3((unsigned)blocks), dim3(256), (size_t)64 * (CP + 2) * 4, st, (const float*)src, (float*)dst,
                       (long long)npix, (int)CP);
    TMG_CHECK_LAUNCH();
    return 0;
}
