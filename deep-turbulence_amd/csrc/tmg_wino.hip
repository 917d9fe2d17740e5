// tmg_wino.hip -- 3x3 / stride-1 convolution with MANY output channels as Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
// The direct implicit GEMM (tmg_conv.hip) spends 9 multiply-adds per (pixel, input channel, output channel); it runs at ~70 % of
// the fp32 MFMA peak on the widest contractions of the path - the ConvLSTM gate conv (104 -> 256 channels, reference
// convLSTM.py:72-74) and the level-wide conditioning contraction (32 -> 16 x 15 channels, flowAffine.py:74 restructured in
// tmg_ops.LevelCouplingFn) - i.e. it is matrix-pipe bound and only fewer multiplies make it faster.  Winograd's minimal filtering
// computes a 2x2 output tile from a 4x4 input tile with 16 multiplies instead of 36:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray 2016; the (.) is summed over input channels)
// so the contraction becomes 16 independent GEMMs  M_pos[tile][co] = sum_ci V_pos[tile][ci] U_pos[ci][co], 2.25x fewer MFMA
// operations, plus transforms that are a few per cent of the matrix work at these channel counts.  fp32 throughout (G has
// entries 1/2: exact); the result differs from the direct sum by ordinary fp32 rounding (a few ulp of the accumulated magnitude),
// the parity tests hold it to the same tolerance as the direct kernel.
//
// One 512-thread block (8 waves, two per SIMD) owns an 8x16-pixel output tile = 4x8 Winograd tiles (two 16-row m-tiles) and up
// to 256 output channels (wave w: n-tiles 2w, 2w+1), and walks the input channels in chunks of 32:
//   stage      raw input patch (10x18 pixels, halo 1, zero or replicate padding, optional ReLU) global -> registers -> LDS,
//              double-buffered, the loads of stage k+2 in flight while stage k computes (the lean scheme of conv_fwd_kernel);
//   transform  V[pos][tile][ci] = (B^T d B)[pos] from the raw patch, all threads, LDS -> LDS (one float4 of channels per item);
//   multiply   for each of the 16 positions: acc = V_pos U_pos over the chunk (A = U fragment straight from the L2-resident packed
//              operand, B = V fragment from LDS, 32 MFMAs per wave), then Y[o] += a(o, pos) * acc with a in {0, +1, -1}: the output
//              transform is linear, so it is applied to every chunk's partial sum and only the four Y tiles stay in registers;
//   epilogue   after the last chunk: Y + bias -> NHWC float4 stores (lane = 4 consecutive channels of one Winograd tile).
#include "tmg_common.h"
#include <stdlib.h>

struct WinoP {
    TmgSeg in[TMG_MAX_IN_SEG];
    int nseg;
    int B, Hin, Win;                 // stride 1, padding 1: output size == input size
    int Cin, Cin_pad, Cout, Npad;    // Cin_pad: multiple of 16 (K of the packed operand); Npad: Cout rounded to 16
    const float* U;                  // [16 pos][Cin_pad/16][Npad][16]
    const float* bias;               // [Cout] or null
    int relu_in, pad_rep;
    float* out; int ostride, ooff;
    int tiles_x, tiles_y, ntiles;    // 8x16-pixel tiles of the whole batch
    int nchunks;                     // 32-channel chunks per tile
};

// U = G g G^T per (output channel, input channel):  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int Kpad, int Npad) {
    const size_t total = (size_t)Kpad * Npad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c16 = i & 15;
        size_t r = i >> 4;
        const int n = r % Npad;
        const int kb = r / Npad;
        const int k = kb * 16 + c16;
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = (k < Cin && n < Cout) ? w[((size_t)n * Cin + k) * 9 + a * 3 + b] : 0.f;
        float t[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t[0][b] = g[0][b];
            t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            t[3][b] = g[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
            const size_t plane = (size_t)Kpad * Npad;
            U[(size_t)(a * 4 + 0) * plane + i] = u0;
            U[(size_t)(a * 4 + 1) * plane + i] = u1;
            U[(size_t)(a * 4 + 2) * plane + i] = u2;
            U[(size_t)(a * 4 + 3) * plane + i] = u3;
        }
    }
}

// w: torch layout [Cout][Cin][3][3]; U: [16][Cin_pad/16][Npad][16] floats (Cin_pad = Cin rounded to 16, Npad = Cout rounded to 16)
extern "C" int tmg_conv_wino_pack(const void* w, void* U, int64_t Cout, int64_t Cin, hipStream_t st) {
    const int Kpad = ((int)Cin + 15) & ~15, Npad = ((int)Cout + 15) & ~15;
    const size_t total = (size_t)Kpad * Npad;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, st, (const float*)w, (float*)U, (int)Cout, (int)Cin, Kpad, Npad);
    TMG_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(512, 1) void wino_fwd_kernel(WinoP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 512;
    constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, PP = PH * PW;   // output tile, raw patch (halo 1)
    constexpr int KC = 32, CS = KC + 8;        // channels per chunk, raw-patch pixel stride (words)
    constexpr int VS = KC + 8;                 // V row stride (words): a fragment read is a float4 per lane at li * VS + 4 q
    constexpr int RAWW = PP * CS;              // words per raw buffer
    constexpr int VPL = 32 * VS;               // words per V position plane (32 Winograd tiles)
    float* Vb = lds + 2 * RAWW;                // [16][32][VS]
    constexpr int UPI = (PP * (KC / 4) + NT - 1) / NT;   // raw float4 items per thread (3)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int KB = p.Cin_pad >> 4;
    const int ntt = p.Npad >> 4;
    const int ntile0 = (int)blockIdx.y * 16 + 2 * wave;
    const size_t kb_stride = (size_t)p.Npad * 16, pos_stride = (size_t)KB * p.Npad * 16;
    // B-operand (U) lane offsets of this wave's two n-tiles (tiles past the end repeat the last: dropped in the epilogue)
    const int boff[2] = {li * 16 + 4 * q + min(ntile0, ntt - 1) * 256, li * 16 + 4 * q + min(ntile0 + 1, ntt - 1) * 256};

    // ---- lean staging state: a thread owns channel quad pc4 of every 64th patch pixel -------------------------------------
    const int pc4 = tid & 7, ppix0 = tid >> 3;
    unsigned pyx[UPI];
#pragma unroll
    for (int u = 0; u < UPI; ++u) {
        const int pix = min(ppix0 + u * 64, PP - 1);
        const int py = pix / PW, px = pix - py * PW;
        pyx[u] = ((unsigned)py << 16) | (unsigned)px;
    }
    float4 pv[UPI];

    // ---- transform mapping: item = (Winograd tile t, channel quad c4), rows xi = 2 h, 2 h + 1 of the 4x4 result; the two halves
    //      live in different waves (no intra-wave bank conflicts between them)
    const int th = tid >> 8, tc4 = tid & 7, tt = (tid >> 3) & 31;
    const int tty = tt >> 3, ttx = tt & 7;
    const int traw = ((2 * tty) * PW + 2 * ttx) * CS + 4 * tc4;   // word offset of patch pixel (0, 0) of the tile in a raw buffer
    const int tv = tt * VS + 4 * tc4;                             // word offset inside a V plane

    const int G = gridDim.x;
    const int nmine = (int)blockIdx.x < p.ntiles ? (p.ntiles - (int)blockIdx.x + G - 1) / G : 0;
    const int nchunks = p.nchunks, nst = nmine * nchunks;
    // cursors: ci/ti = chunk / tile of the stage being issued, cc = chunk being committed, cm/tm = chunk / tile being computed
    int ci = 0, cc = 0, cm = 0;
    int ti = blockIdx.x, tm = blockIdx.x;

    f32x4 Y[4][2][2];   // [output pixel of the 2x2 tile][m-tile][n-tile]
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) Y[o][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 bfr[4][2][2];   // U fragments [ring][16-channel group][n-tile], three positions ahead of the MFMAs
    for (int k = -2; k < nst; ++k) {
        // ---- commit stage k+1 --------------------------------------------------------------------------------------------------
        if (k >= -1 && k + 1 < nst) {
            float* rb = lds + ((k + 1) & 1) * RAWW;
#pragma unroll
            for (int u = 0; u < UPI; ++u) {
                if (ppix0 + u * 64 < PP) {
                    float4 v = pv[u];
                    if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    *reinterpret_cast<float4*>(rb + (ppix0 + u * 64) * CS + 4 * pc4) = v;
                }
            }
            if (++cc == nchunks) cc = 0;
        }
        // ---- issue the loads of stage k+2 -------------------------------------------------------------------------------------
        if (k + 2 < nst) {
            int t_ = ti;
            const int tx_ = t_ % p.tiles_x; t_ /= p.tiles_x;
            const int ty_ = t_ % p.tiles_y;
            const int b_ = t_ / p.tiles_y;
            const int iy0 = ty_ * TH - 1, ix0 = tx_ * TW - 1;
            const float* tptr = tmg_zero_page;
            int tss = 0;
            {
                int cl = ci * KC + 4 * pc4;
                if (cl < p.Cin) {
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
                    tptr = sp + so + cl;
                    tss = ss;
                }
            }
            const size_t tbv = (size_t)b_ * p.Hin * p.Win;
#pragma unroll
            for (int u = 0; u < UPI; ++u) {
                const int iy = iy0 + (int)(pyx[u] >> 16), ix = ix0 + (int)(pyx[u] & 0xffffu);
                const int iyc = min(max(iy, 0), p.Hin - 1), ixc = min(max(ix, 0), p.Win - 1);
                const bool oob = !p.pad_rep && (iy != iyc || ix != ixc);
                const float* a_ = (oob || ppix0 + u * 64 >= PP) ? tmg_zero_page : tptr + (tbv + (size_t)iyc * p.Win + ixc) * tss;
                pv[u] = *reinterpret_cast<const float4*>(a_);
            }
            if (++ci == nchunks) { ci = 0; ti += G; }
        }
        if (k >= 0) {
            // ---- input transform of stage k: V = B^T d B,  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]] ----------------------
            {
                const float* rb = lds + (k & 1) * RAWW + traw;
#define TMG_W4(OP, A_, B_) make_float4(A_.x OP B_.x, A_.y OP B_.y, A_.z OP B_.z, A_.w OP B_.w)
                // row pass, one patch column at a time (12 registers of raw data live instead of 48):
                // th = 0: xi 0 = row0 - row2, xi 1 = row1 + row2 ; th = 1 (rows 1, 2, 3 loaded): xi 2 = row2 - row1, xi 3 = row1 - row3
                float4 t[2][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 d0 = *reinterpret_cast<const float4*>(rb + ((th + 0) * PW + c) * CS);
                    const float4 d1 = *reinterpret_cast<const float4*>(rb + ((th + 1) * PW + c) * CS);
                    const float4 d2 = *reinterpret_cast<const float4*>(rb + ((th + 2) * PW + c) * CS);
                    if (th == 0) { t[0][c] = TMG_W4(-, d0, d2); t[1][c] = TMG_W4(+, d1, d2); }
                    else         { t[0][c] = TMG_W4(-, d1, d0); t[1][c] = TMG_W4(-, d0, d2); }
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float* vb = Vb + ((2 * th + e) * 4) * VPL + tv;
                    *reinterpret_cast<float4*>(vb) = TMG_W4(-, t[e][0], t[e][2]);
                    *reinterpret_cast<float4*>(vb + VPL) = TMG_W4(+, t[e][1], t[e][2]);
                    *reinterpret_cast<float4*>(vb + 2 * VPL) = TMG_W4(-, t[e][2], t[e][1]);
                    *reinterpret_cast<float4*>(vb + 3 * VPL) = TMG_W4(-, t[e][1], t[e][3]);
                }
#undef TMG_W4
            }
            __syncthreads();
            // ---- 16 position GEMMs over this chunk, output transform folded in ---------------------------------------------------
            {
                const int c0 = cm * KC;
                const int kgn = min(KC, p.Cin_pad - c0) >> 4;   // 16-channel groups in this chunk (1 or 2)
                const float* ub = p.U + (size_t)(c0 >> 4) * kb_stride;
                const float4* v4 = reinterpret_cast<const float4*>(Vb) + (li * VS + 4 * q) / 4;
#define TMG_WN_LOADB(R, UB, KGN, POS)                                                                                 \
                {                                                                                                     \
                    const float* up_ = (UB) + (size_t)(POS) * pos_stride;                                             \
                    _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                   \
                        bfr[R][0][n] = *reinterpret_cast<const float4*>(up_ + boff[n]);                               \
                        bfr[R][1][n] = *reinterpret_cast<const float4*>(up_ + (size_t)((KGN) - 1) * kb_stride + boff[n]); \
                    }                                                                                                 \
                }
                if (k == 0) { TMG_WN_LOADB(0, ub, kgn, 0) TMG_WN_LOADB(1, ub, kgn, 1) TMG_WN_LOADB(2, ub, kgn, 2) }
                // (every later stage gets its first three fragment sets from the previous stage's last positions)
                // operand of the next stage (the next chunk of this tile, or chunk 0 of the next tile: every tile uses the same U)
                const int c0n = (cm + 1 == nchunks) ? 0 : c0 + KC;
                const int kgn_n = min(KC, p.Cin_pad - c0n) >> 4;
                const float* ubn = p.U + (size_t)(c0n >> 4) * kb_stride;
#pragma unroll
                for (int pos = 0; pos < 16; ++pos) {
                    const int R = pos & 3;
                    // (without the scheduling fences the compiler hoists all 16 positions' LDS reads to the top and spills)
                    __builtin_amdgcn_sched_barrier(0);
                    // U fragments three positions ahead (a position is only 32 MFMAs per wave: one position of lead does not cover
                    // the L2 latency), V fragments of this position from LDS
                    if (pos + 3 < 16) TMG_WN_LOADB((pos + 3) & 3, ub, kgn, pos + 3)
                    else TMG_WN_LOADB((pos + 3) & 3, ubn, kgn_n, pos + 3 - 16)
                    float4 af[2][2];   // [16-channel group][m-tile]
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        af[0][m] = v4[(pos * VPL + m * 16 * VS) / 4];
                        af[1][m] = v4[(pos * VPL + m * 16 * VS + 16) / 4];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 acc[2][2];
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define TMG_WN_STEP(KG, E)                                                                                            \
                    _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < 2; ++n)         \
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[R][KG][n].E, af[KG][m].E, acc[m][n], 0, 0, 0);
                    TMG_WN_STEP(0, x) TMG_WN_STEP(0, y) TMG_WN_STEP(0, z) TMG_WN_STEP(0, w)
                    if (kgn == 2) { TMG_WN_STEP(1, x) TMG_WN_STEP(1, y) TMG_WN_STEP(1, z) TMG_WN_STEP(1, w) }
#undef TMG_WN_STEP
                    // Y = A^T M A,  A^T = [[1,1,1,0],[0,1,-1,-1]]:  coefficient of position (xi, nu) in output (oy, ox) = a[oy][xi] a[ox][nu]
                    const int xi = pos >> 2, nu = pos & 3;
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                        for (int ox = 0; ox < 2; ++ox) {
                            const int ay = oy == 0 ? (xi < 3 ? 1 : 0) : (xi == 0 ? 0 : (xi == 1 ? 1 : -1));
                            const int ax = ox == 0 ? (nu < 3 ? 1 : 0) : (nu == 0 ? 0 : (nu == 1 ? 1 : -1));
                            const int cf = ay * ax;
                            if (cf != 0) {
#pragma unroll
                                for (int m = 0; m < 2; ++m)
#pragma unroll
                                    for (int n = 0; n < 2; ++n)
#pragma unroll
                                        for (int r = 0; r < 4; ++r) {
                                            if (cf > 0) Y[oy * 2 + ox][m][n][r] += acc[m][n][r];
                                            else Y[oy * 2 + ox][m][n][r] -= acc[m][n][r];
                                        }
                                // pin the update HERE: left alone the compiler sinks all 16 positions' additions below the last
                                // position and keeps 16 x 16 accumulator registers alive (300 spilled registers)
                                asm volatile("" : "+v"(Y[oy * 2 + ox][0][0]), "+v"(Y[oy * 2 + ox][0][1]), "+v"(Y[oy * 2 + ox][1][0]), "+v"(Y[oy * 2 + ox][1][1]));
                            }
                        }
                }
#undef TMG_WN_LOADB
            }
            if (cm + 1 == nchunks) {
                // ---- epilogue: lane (li, q) holds channels 4 q .. 4 q + 3 (of each n-tile) of Winograd tile 16 m + li -------------
                int t_ = tm;
                const int tx_ = t_ % p.tiles_x; t_ /= p.tiles_x;
                const int ty_ = t_ % p.tiles_y;
                const int b_ = t_ / p.tiles_y;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int wt = 16 * m + li;
                    const int oyb = ty_ * TH + 2 * (wt >> 3), oxb = tx_ * TW + 2 * (wt & 7);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int n0 = (ntile0 + n) * 16 + 4 * q;
                        if (ntile0 + n < ntt && n0 < p.Cout) {
                            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n0);
#pragma unroll
                            for (int o = 0; o < 4; ++o) {
                                const int oy = oyb + (o >> 1), ox = oxb + (o & 1);
                                if (oy < p.Hin && ox < p.Win) {
                                    const size_t opx = ((size_t)b_ * p.Hin + oy) * p.Win + ox;
                                    *reinterpret_cast<float4*>(p.out + opx * p.ostride + p.ooff + n0) =
                                        make_float4(Y[o][m][n][0] + bv.x, Y[o][m][n][1] + bv.y, Y[o][m][n][2] + bv.z, Y[o][m][n][3] + bv.w);
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int o = 0; o < 4; ++o)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) Y[o][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                cm = 0; tm += G;
            } else {
                ++cm;
            }
        }
        __syncthreads();   // V and the raw buffer just read are rewritten next round; the raw buffer just written is complete
    }
}

// out = conv3x3_stride1(pad(act(in))) + bias with the Winograd operand of tmg_conv_wino_pack.
// dims = {B, H, W, Cin, Cout, relu_in, pad_replicate}; in_desc = {stride, off, n} per segment; out_desc = {stride, off}.
// Envelope: float4-addressable segments and output, Cin % 4 == 0, Cout % 4 == 0, Cout >= 64; returns -100 outside it (the caller
// uses tmg_conv_fwd).
extern "C" int tmg_conv_wino_fwd(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* U, const void* bias,
                                 void* out, const int64_t* out_desc, const int64_t* dims, hipStream_t st) {
    WinoP p;
    p.nseg = (int)nseg;
    if (p.nseg < 1 || p.nseg > TMG_MAX_IN_SEG) return -3;
    int csum = 0;
    bool ok = true;
    for (int i = 0; i < TMG_MAX_IN_SEG; ++i) p.in[i] = TmgSeg{nullptr, 0, 0, 0};
    for (int i = 0; i < p.nseg; ++i) {
        p.in[i] = TmgSeg{(const float*)in_ptrs[i], (int)in_desc[3 * i], (int)in_desc[3 * i + 1], (int)in_desc[3 * i + 2]};
        if (((p.in[i].stride | p.in[i].off | p.in[i].n) & 3) || (((uintptr_t)in_ptrs[i]) & 15)) ok = false;
        csum += p.in[i].n;
    }
    p.B = (int)dims[0]; p.Hin = (int)dims[1]; p.Win = (int)dims[2]; p.Cin = (int)dims[3]; p.Cout = (int)dims[4];
    p.relu_in = (int)dims[5]; p.pad_rep = (int)dims[6];
    if (csum != p.Cin) return -3;
    p.out = (float*)out; p.ostride = (int)out_desc[0]; p.ooff = (int)out_desc[1];
    if (((p.ostride | p.ooff) & 3) || (((uintptr_t)out) & 15) || (p.Cout & 3) || (p.Cin & 3) || p.Cout < 64) ok = false;
    if (bias && (((uintptr_t)bias) & 15)) ok = false;
    if (!ok) return -100;
    p.Cin_pad = (p.Cin + 15) & ~15;
    p.Npad = (p.Cout + 15) & ~15;
    p.U = (const float*)U; p.bias = (const float*)bias;
    p.tiles_x = (p.Win + 15) / 16; p.tiles_y = (p.Hin + 7) / 8;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    p.nchunks = (p.Cin_pad + 31) / 32;
    if (p.ntiles <= 0) return 0;
    const int gy = (p.Npad / 16 + 15) / 16;
    int G = 256 / gy;
    if (G < 1) G = 1;
    if (G > p.ntiles) G = p.ntiles;
    const size_t lds_bytes = (size_t)(2 * 180 * 40 + 16 * 32 * 40) * sizeof(float);
    TMG_LDS_OPTIN((&wino_fwd_kernel));
    TmgProf prof(TMG_PROF_WINO, 2.0 * p.B * p.Hin * p.Win * (double)p.Cout * p.Cin * 9, st);   // algorithmic (direct) flops
    hipLaunchKernelGGL(wino_fwd_kernel, dim3(G, gy, 1), dim3(512), lds_bytes, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}
