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
#include <type_traits>

struct WinoP {
    TmgSeg in[TMG_MAX_IN_SEG];
    int nseg;
    int B, Hin, Win;                 // stride 1, padding 1: output size == input size
    int Cin, Cin_pad, Cout, Npad;    // Cin_pad: multiple of 16 (K of the packed operand); Npad: Cout rounded to 16
    const float* U;                  // [16 pos][Cin_pad/16][Npad][16]
    const float* bias;               // [Cout] or null
    int relu_in, pad_rep;
    TmgOSeg out[TMG_MAX_OUT_SEG];    // up to 3 output segments (the concatenated Cout channels)
    int tiles_x, tiles_y, ntiles;    // 8x16-pixel tiles of the whole batch
    int nchunks;                     // 32-channel chunks per tile
};

// U = G g G^T per (output channel, input channel):  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
// mode 0: forward operand, K = Cin, N = Cout, g = w[n][k].   mode 1: input-gradient operand (the transposed conv): K = Cout,
// N = the first `nvalid` input channels, g = w[k][n] with the taps flipped.
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int K, int N, int Kpad, int Npad, int mode) {
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
            for (int b = 0; b < 3; ++b) {
                float v = 0.f;
                if (k < K && n < N) v = mode == 0 ? w[((size_t)n * Cin + k) * 9 + a * 3 + b] : w[((size_t)k * Cin + n) * 9 + (2 - a) * 3 + (2 - b)];
                g[a][b] = v;
            }
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

// w: torch layout [Cout][Cin][3][3]; U: [16][Kpad/16][Npad][16] floats.  mode 0: K = Cin, N = Cout (forward).  mode 1: K = Cout,
// N = nvalid (<= Cin; 0 -> Cin): the operand of the input gradient w.r.t. the first N input channels.  Kpad / Npad: rounded to 16.
extern "C" int tmg_conv_wino_pack(const void* w, void* U, int64_t Cout, int64_t Cin, int64_t mode, int64_t nvalid, hipStream_t st) {
    const int K = mode == 0 ? (int)Cin : (int)Cout;
    const int N = mode == 0 ? (int)Cout : (int)(nvalid > 0 && nvalid < Cin ? nvalid : Cin);
    const int Kpad = (K + 15) & ~15, Npad = (N + 15) & ~15;
    const size_t total = (size_t)Kpad * Npad;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, st, (const float*)w, (float*)U, (int)Cout, (int)Cin, K, N, Kpad, Npad, (int)mode);
    TMG_CHECK_LAUNCH();
    return 0;
}

// Diagnostic build only (-DTMG_WINO_STAMP, tools/scratch/wino_stamps.py): s_memtime stamps at the phase boundaries of wino_fwd_kernel,
// cycles per phase summed over waves into g_wino_stamps (never read by the kernel; the product build contains no stamp).
#ifdef TMG_WINO_STAMP
__device__ unsigned long long g_wino_stamps[16];
#define TMG_STAMP(I) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[I] += (unsigned)(t_ - st_last); st_last = t_; }
extern "C" int tmg_wino_stamps(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wino_stamps), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino_stamps), 16 * sizeof(unsigned long long));
}
#else
#define TMG_STAMP(I)
#endif

// NPW: 16-channel output tiles per wave.  2: a block covers up to 256 output channels (wave w: n-tiles 2w, 2w+1).  1: up to 128 -
// the contractions with 64..128 output channels (the ConvLSTM block's out-conv input gradient, 40 -> 104 at the first level) would leave
// waves 4-7 of the 2-tile form multiplying repeated tiles; with one tile per wave all eight waves carry live work.
// One Winograd position (xi, nu) of a 32-channel chunk in the multiply loops of wino_fwd_kernel / wino_fwdp_kernel: the MFMAs of
// the position and its share of the output transform  Y = A^T M A,  A^T = [[1,1,1,0],[0,1,-1,-1]].  The fp32 MFMA runs on the
// vector ALUs, so every v_add of the transform is matrix time lost (skipping all of them: -6.7 % of the gate conv, round 6).  Adding
// every M_pos into the (up to four) Y tiles it belongs to costs 36 tile additions per chunk; this form costs 22:
//   * the MFMA accumulates by itself: a position that feeds ONE tile with coefficient +-1 is chained straight onto that tile (the
//     sign is folded into V: the input transform writes -V for nu = 3 and for the row xi = 3, at no cost - a swapped subtraction);
//   * rows xi = 0 / 3 feed only the output row oy = 0 / 1: nu = 0 -> chained onto Y[oy][0], nu = 3 -> chained onto Y[oy][1],
//     nu = 1, 2 -> a fresh tile added to both (4 additions per row);
//   * rows xi = 1, 2 feed both output rows: their column sums T0 = M0 + M1 + M2, T1 = M1 - M2 - M3 are built first (chained / 3
//     additions) and added to the four Y tiles once (4 additions).
#define TMG_WN_STEP(D, KG, E)                                                                                         \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < NPW; ++n)                       \
        D[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[R][KG][n].E, af[KG][m].E, D[m][n], 0, 0, 0);
#define TMG_WN_CHAIN(D)                                                                                               \
    {                                                                                                                 \
        TMG_WN_STEP(D, 0, x) TMG_WN_STEP(D, 0, y) TMG_WN_STEP(D, 0, z) TMG_WN_STEP(D, 0, w)                           \
        if (kgn == 2) { TMG_WN_STEP(D, 1, x) TMG_WN_STEP(D, 1, y) TMG_WN_STEP(D, 1, z) TMG_WN_STEP(D, 1, w) }         \
    }
#define TMG_WN_ZERO(D) _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < NPW; ++n) D[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define TMG_WN_ADD(D, S) _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < NPW; ++n) D[m][n] += S[m][n];
#define TMG_WN_SUB(D, S) _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < NPW; ++n) D[m][n] -= S[m][n];
// pin an update HERE: left alone the compiler sinks all 16 positions' additions below the last position and keeps 16 x 16
// accumulator registers alive (300 spilled registers)
#define TMG_WN_PIN(A)                                                                                                 \
    {                                                                                                                 \
        if constexpr (NPW == 2) asm volatile("" : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[1][0]), "+v"(A[1][1]));        \
        else asm volatile("" : "+v"(A[0][0]), "+v"(A[1][0]));                                                         \
    }
// (pos, R, kgn, bfr, af, Y, T0, T1 of the enclosing loop; pos is a constant after unrolling)
#define TMG_WN_POSITION                                                                                               \
    {                                                                                                                 \
        const int xi = pos >> 2, nu = pos & 3;                                                                        \
        if (xi == 0 || xi == 3) {                                                                                     \
            const int o0 = xi == 0 ? 0 : 2;                                                                           \
            if (nu == 0) { TMG_WN_CHAIN(Y[o0]) }                                                                      \
            else if (nu == 3) { TMG_WN_CHAIN(Y[o0 + 1]) }                                                             \
            else {                                                                                                    \
                f32x4 acc[2][NPW];                                                                                    \
                TMG_WN_ZERO(acc) TMG_WN_CHAIN(acc)                                                                    \
                TMG_WN_ADD(Y[o0], acc)                                                                                \
                if (nu == 1) { TMG_WN_ADD(Y[o0 + 1], acc) } else { TMG_WN_SUB(Y[o0 + 1], acc) }                       \
                TMG_WN_PIN(Y[o0]) TMG_WN_PIN(Y[o0 + 1])                                                               \
            }                                                                                                         \
        } else {                                                                                                      \
            if (nu == 0) { TMG_WN_ZERO(T0) TMG_WN_CHAIN(T0) }                                                         \
            else if (nu == 1) { TMG_WN_ZERO(T1) TMG_WN_CHAIN(T1) }                                                    \
            else if (nu == 2) {                                                                                       \
                f32x4 acc[2][NPW];                                                                                    \
                TMG_WN_ZERO(acc) TMG_WN_CHAIN(acc)                                                                    \
                TMG_WN_ADD(T0, T1) TMG_WN_ADD(T0, acc) TMG_WN_SUB(T1, acc)                                            \
                TMG_WN_PIN(T0) TMG_WN_PIN(T1)                                                                         \
            } else {                                                                                                  \
                TMG_WN_CHAIN(T1)                                                                                      \
                TMG_WN_ADD(Y[0], T0) TMG_WN_ADD(Y[1], T1)                                                             \
                if (xi == 1) { TMG_WN_ADD(Y[2], T0) TMG_WN_ADD(Y[3], T1) } else { TMG_WN_SUB(Y[2], T0) TMG_WN_SUB(Y[3], T1) } \
                TMG_WN_PIN(Y[0]) TMG_WN_PIN(Y[1]) TMG_WN_PIN(Y[2]) TMG_WN_PIN(Y[3])                                   \
            }                                                                                                         \
        }                                                                                                             \
    }
#ifdef TMG_WINO_FLAT_U
#define TMG_WN_ULOAD(DST, PTR, BOFF) DST = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(PTR) + (BOFF));
#else
#define TMG_WN_ULOAD(DST, PTR, BOFF) DST = tmg_bload4(urs, BOFF, (unsigned)(((PTR) - p.U) * 4));
#endif
// (TMG_PACKED_F32: this kernel keeps the packed-fp32 instructions the rest of the file is built without - its float4 input transform and
// the tile additions of the output transform halve their instruction count: gate conv 2.045 -> 2.023 ms, conditioning contraction
// 0.663 -> 0.636 ms; wino_fwdp_kernel, wino_nn_kernel (+3 %) and wino_wgrad_kernel (+-0) measured no better with them, round 6)
template <int NPW>
__global__ __launch_bounds__(512, 1) TMG_PACKED_F32 void wino_fwd_kernel(WinoP p) {
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
    const int ntile0 = (int)blockIdx.y * (8 * NPW) + NPW * wave;
    const size_t kb_stride = (size_t)p.Npad * 16, pos_stride = (size_t)KB * p.Npad * 16;
    // B-operand (U) lane offsets of this wave's two n-tiles (tiles past the end repeat the last: dropped in the epilogue)
    unsigned boff[NPW];   // BYTE offsets, unsigned: scalar base + 32-bit lane offset selects the saddr form of global_load (no 64-bit vector add per load)
#pragma unroll
    for (int n = 0; n < NPW; ++n) boff[n] = 4u * (unsigned)(li * 16 + 4 * q + min(ntile0 + n, ntt - 1) * 256);

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

    // U through a buffer descriptor: scalar byte offset (position, channel group) + the lane's constant 32-bit offset - no vector
    // address arithmetic per load (flat global loads cost a 64-bit vector add each: the fp32 MFMA shares the vector ALUs)
    const __amdgpu_buffer_rsrc_t urs = tmg_make_rsrc(p.U, 16u * (unsigned)pos_stride * 4u);
    // The bias of this lane's channels, loaded ONCE and branch-free (no bias / channels past the end: the zero page): a load inside the
    // epilogue's conditional blocks makes the compiler put `s_waitcnt vmcnt(0)` in front of EVERY store there - each of the 16 stores
    // of a tile then waits for the one before it and for the whole prefetched U ring (round 6: 6-14 % of the launch).  Y starts at
    // the bias instead of adding it in the epilogue.
    f32x4 bvr[NPW];
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        const int n0 = (ntile0 + n) * 16 + 4 * q;
        const float* bp = (p.bias != nullptr && ntile0 + n < ntt && n0 < p.Cout) ? p.bias + n0 : tmg_zero_page;
        const float4 b4 = *reinterpret_cast<const float4*>(bp);
        bvr[n] = (f32x4){b4.x, b4.y, b4.z, b4.w};
    }
    f32x4 Y[4][2][NPW];   // [output pixel of the 2x2 tile][m-tile][n-tile]
    f32x4 T0[2][NPW], T1[2][NPW];   // column sums of a middle row (TMG_WN_POSITION)
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NPW; ++n) Y[o][m][n] = bvr[n];

    float4 bfr[4][2][NPW];   // U fragments [ring][16-channel group][n-tile], three positions ahead of the MFMAs
#ifdef TMG_WINO_STAMP
    unsigned st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    for (int k = -2; k < nst; ++k) {
        TMG_STAMP(7)   // loop overhead / previous barrier exit
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
        TMG_STAMP(0)   // commit
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
            const unsigned tbv = (unsigned)b_ * (unsigned)(p.Hin * p.Win);   // 32-bit pixel index: the launcher refuses B H W >= 2^31
#pragma unroll
            for (int u = 0; u < UPI; ++u) {
                const int iy = iy0 + (int)(pyx[u] >> 16), ix = ix0 + (int)(pyx[u] & 0xffffu);
                const int iyc = min(max(iy, 0), p.Hin - 1), ixc = min(max(ix, 0), p.Win - 1);
                const bool oob = !p.pad_rep && (iy != iyc || ix != ixc);
                const float* a_ = (oob || ppix0 + u * 64 >= PP) ? tmg_zero_page : tptr + (size_t)(tbv + (unsigned)iyc * (unsigned)p.Win + (unsigned)ixc) * (unsigned)tss;
                pv[u] = *reinterpret_cast<const float4*>(a_);
            }
            if (++ci == nchunks) { ci = 0; ti += G; }
        }
        TMG_STAMP(1)   // issue
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
                    else         { t[0][c] = TMG_W4(-, d1, d0); t[1][c] = TMG_W4(-, d2, d0); }   // xi 3 with the sign flipped (TMG_WN_POSITION)
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float* vb = Vb + ((2 * th + e) * 4) * VPL + tv;
                    *reinterpret_cast<float4*>(vb) = TMG_W4(-, t[e][0], t[e][2]);
                    *reinterpret_cast<float4*>(vb + VPL) = TMG_W4(+, t[e][1], t[e][2]);
                    *reinterpret_cast<float4*>(vb + 2 * VPL) = TMG_W4(-, t[e][2], t[e][1]);
                    *reinterpret_cast<float4*>(vb + 3 * VPL) = TMG_W4(-, t[e][3], t[e][1]);      // -V at nu = 3 (TMG_WN_POSITION)
                }
#undef TMG_W4
            }
            TMG_STAMP(2)   // transform
            __syncthreads();
            TMG_STAMP(3)   // barrier 1
            // ---- 16 position GEMMs over this chunk, output transform folded in ---------------------------------------------------
            {
                const int c0 = cm * KC;
                const int kgn = min(KC, p.Cin_pad - c0) >> 4;   // 16-channel groups in this chunk (1 or 2)
                const float* ub = p.U + (size_t)(c0 >> 4) * kb_stride;
                const float4* v4 = reinterpret_cast<const float4*>(Vb) + (li * VS + 4 * q) / 4;
#define TMG_WN_LOADB(R, UB, KGN, POS)                                                                                 \
                {                                                                                                     \
                    const float* up_ = (UB) + (size_t)(POS) * pos_stride;                                             \
                    _Pragma("unroll") for (int n = 0; n < NPW; ++n) {                                                 \
                        TMG_WN_ULOAD(bfr[R][0][n], up_, boff[n])                                                    \
                        TMG_WN_ULOAD(bfr[R][1][n], up_ + (size_t)((KGN) - 1) * kb_stride, boff[n])                  \
                    }                                                                                                 \
                }
                // LEAD: how many positions ahead of the MFMAs the U fragments are fetched.  One position is 2 x 1 024 matrix cycles on a
                // SIMD and U is L2-resident.  With two n-tiles per wave a lead of two positions keeps three ring slots live instead of
                // four: 238 registers against 254 with the column sums of TMG_WN_POSITION; measured equal over the step's shapes (gate
                // conv 2.045 / 2.081 ms, conditioning contraction 0.663 / 0.642 ms at a lead of 2 / 3).
                constexpr int LEAD = NPW == 2 ? 2 : 3;
                if (k == 0) { TMG_WN_LOADB(0, ub, kgn, 0) TMG_WN_LOADB(1, ub, kgn, 1) if (LEAD == 3) TMG_WN_LOADB(2, ub, kgn, 2) }
                // (every later stage gets its first three fragment sets from the previous stage's last positions)
                // operand of the next stage (the next chunk of this tile, or chunk 0 of the next tile: every tile uses the same U)
                const int c0n = (cm + 1 == nchunks) ? 0 : c0 + KC;
                const int kgn_n = min(KC, p.Cin_pad - c0n) >> 4;
                const float* ubn = p.U + (size_t)(c0n >> 4) * kb_stride;
                // The 16 positions are written out as 16 calls of one generic lambda (the position is a compile-time constant inside):
                // as an unrolled loop the two-tile instance was too large for the early full-unroll pass, the loop was unrolled after the
                // last scalar-replacement pass and the U ring stayed in scratch memory.
                auto one_position = [&](auto pc_) __attribute__((always_inline)) {
                    constexpr int pos = decltype(pc_)::value;
                    constexpr int R = pos & 3;
                    // (without the scheduling fences the compiler hoists all 16 positions' LDS reads to the top and spills)
                    __builtin_amdgcn_sched_barrier(0);
                    // U fragments three positions ahead (a position is only 32 MFMAs per wave: one position of lead does not cover
                    // the L2 latency), V fragments of this position from LDS
                    if (pos + LEAD < 16) TMG_WN_LOADB((pos + LEAD) & 3, ub, kgn, pos + LEAD)
                    else TMG_WN_LOADB((pos + LEAD) & 3, ubn, kgn_n, pos + LEAD - 16)
                    float4 af[2][2];   // [16-channel group][m-tile]
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        af[0][m] = v4[(pos * VPL + m * 16 * VS) / 4];
                        af[1][m] = v4[(pos * VPL + m * 16 * VS + 16) / 4];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    TMG_WN_POSITION
                };
#define TMG_WN_P(I) one_position(std::integral_constant<int, I>{});
                TMG_WN_P(0) TMG_WN_P(1) TMG_WN_P(2) TMG_WN_P(3) TMG_WN_P(4) TMG_WN_P(5) TMG_WN_P(6) TMG_WN_P(7)
                TMG_WN_P(8) TMG_WN_P(9) TMG_WN_P(10) TMG_WN_P(11) TMG_WN_P(12) TMG_WN_P(13) TMG_WN_P(14) TMG_WN_P(15)
#undef TMG_WN_P
#undef TMG_WN_LOADB
            }
            TMG_STAMP(4)   // MFMA loop
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
                    for (int n = 0; n < NPW; ++n) {
                        const int n0 = (ntile0 + n) * 16 + 4 * q;
                        if (ntile0 + n < ntt && n0 < p.Cout) {
                            int nl = n0;
                            TMG_PICK_OSEG(p.out, nl, optr, ostride, ooff)
#pragma unroll
                            for (int o = 0; o < 4; ++o) {
                                const int oy = oyb + (o >> 1), ox = oxb + (o & 1);
                                if (oy < p.Hin && ox < p.Win) {
                                    const unsigned opx = ((unsigned)b_ * (unsigned)p.Hin + (unsigned)oy) * (unsigned)p.Win + (unsigned)ox;
                                    *reinterpret_cast<float4*>(optr + (size_t)opx * (unsigned)ostride + ooff + nl) =
                                        make_float4(Y[o][m][n][0], Y[o][m][n][1], Y[o][m][n][2], Y[o][m][n][3]);
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
                        for (int n = 0; n < NPW; ++n) Y[o][m][n] = bvr[n];
                cm = 0; tm += G;
            } else {
                ++cm;
            }
        }
        TMG_STAMP(5)   // epilogue
        __syncthreads();   // V and the raw buffer just read are rewritten next round; the raw buffer just written is complete
        TMG_STAMP(6)   // barrier 2
    }
#ifdef TMG_WINO_STAMP
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_wino_stamps[i], (unsigned long long)st_acc[i]);
    if (tid == 0) atomicAdd(&g_wino_stamps[8], 1ull);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6: the same contraction with PRODUCER waves.  wino_fwd_kernel's eight waves stage, transform and multiply in lock step: the
// matrix pipe idles through commit + issue + input transform + two barriers per chunk (24 % of a wave's life, round-3 stamps).  Here
// the block has twelve waves: waves 0-7 only multiply (the position loop, the output transform and the epilogue of wino_fwd_kernel,
// unchanged), waves 8-11 only produce: a producer thread owns (Winograd tile, channel quad) of a chunk, loads its 4x4 input patch
// straight from global memory into registers (16 float4; the overlapping tiles share their pixels through L1 / L2: no raw patch in
// LDS), applies padding rule / ReLU, transforms (B^T d B) in registers and writes the 16 position values into the OTHER of two V
// buffers (2 x 80 KB = all of the CU's LDS).  ONE barrier per chunk: after it the multipliers read the buffer the producers have just
// filled and the producers refill the one the multipliers have just left.  A chunk is 16 positions x 32 MFMAs x 32 cycles x 2 waves per
// SIMD = 33 k cycles of matrix work; a producer needs a few thousand for its loads and ~100 vector instructions, so it always waits
// at the barrier and the multipliers never wait for data after the first stage.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int NPW>
__global__ __launch_bounds__(768, 1) void wino_fwdp_kernel(WinoP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TH = 8, TW = 16;
    constexpr int KC = 32;
    constexpr int VS = KC + 8;                 // V row stride (words): a fragment read is a float4 per lane at li * VS + 4 q
    constexpr int VPL = 32 * VS;               // words per V position plane (32 Winograd tiles)
    constexpr int VBUF = 16 * VPL;             // words per V buffer
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const int nmine = (int)blockIdx.x < p.ntiles ? (p.ntiles - (int)blockIdx.x + G - 1) / G : 0;
    const int nchunks = p.nchunks, nst = nmine * nchunks;
    if (wave >= 8) {
        // ================================================= producers =================================================
        const int ptid = tid - 512;
        const int tc4 = ptid & 7, tt = ptid >> 3;
        const int tty = tt >> 3, ttx = tt & 7;
        const int tv = tt * VS + 4 * tc4;
        int ci = 0, ti = blockIdx.x;
        for (int k = 0; k < nst; ++k) {
            int t_ = ti;
            const int tx_ = t_ % p.tiles_x; t_ /= p.tiles_x;
            const int ty_ = t_ % p.tiles_y;
            const int b_ = t_ / p.tiles_y;
            const int iy0 = ty_ * TH - 1 + 2 * tty, ix0 = tx_ * TW - 1 + 2 * ttx;
            const float* tptr = tmg_zero_page;
            int tss = 0;
            {
                int cl = ci * KC + 4 * tc4;
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
            const unsigned tbv = (unsigned)b_ * (unsigned)(p.Hin * p.Win);
            float4 d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int iy = iy0 + r;
                const int iyc = min(max(iy, 0), p.Hin - 1);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ix = ix0 + c;
                    const int ixc = min(max(ix, 0), p.Win - 1);
                    const bool oob = !p.pad_rep && (iy != iyc || ix != ixc);
                    const float* a_ = oob ? tmg_zero_page : tptr + (size_t)(tbv + (unsigned)iyc * (unsigned)p.Win + (unsigned)ixc) * (unsigned)tss;
                    d[r][c] = *reinterpret_cast<const float4*>(a_);
                }
            }
            if (p.relu_in) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        d[r][c].x = fmaxf(d[r][c].x, 0.f); d[r][c].y = fmaxf(d[r][c].y, 0.f);
                        d[r][c].z = fmaxf(d[r][c].z, 0.f); d[r][c].w = fmaxf(d[r][c].w, 0.f);
                    }
            }
#define TMG_W4(OP, A_, B_) make_float4(A_.x OP B_.x, A_.y OP B_.y, A_.z OP B_.z, A_.w OP B_.w)
            float* vbuf = lds + (k & 1) * VBUF + tv;
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                float4 t[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (xi == 0) t[c] = TMG_W4(-, d[0][c], d[2][c]);
                    else if (xi == 1) t[c] = TMG_W4(+, d[1][c], d[2][c]);
                    else if (xi == 2) t[c] = TMG_W4(-, d[2][c], d[1][c]);
                    else t[c] = TMG_W4(-, d[3][c], d[1][c]);      // xi 3 with the sign flipped (TMG_WN_POSITION)
                }
                float* vb = vbuf + (xi * 4) * VPL;
                *reinterpret_cast<float4*>(vb) = TMG_W4(-, t[0], t[2]);
                *reinterpret_cast<float4*>(vb + VPL) = TMG_W4(+, t[1], t[2]);
                *reinterpret_cast<float4*>(vb + 2 * VPL) = TMG_W4(-, t[2], t[1]);
                *reinterpret_cast<float4*>(vb + 3 * VPL) = TMG_W4(-, t[3], t[1]);                 // -V at nu = 3
            }
#undef TMG_W4
            if (++ci == nchunks) { ci = 0; ti += G; }
            __syncthreads();       // stage k is in V[k & 1]; the multipliers have left V[(k + 1) & 1]
        }
        return;
    }
    // ===================================================== multipliers =====================================================
    const int li = lane & 15, q = lane >> 4;
    const int KB = p.Cin_pad >> 4;
    const int ntt = p.Npad >> 4;
    const int ntile0 = (int)blockIdx.y * (8 * NPW) + NPW * wave;
    const size_t kb_stride = (size_t)p.Npad * 16, pos_stride = (size_t)KB * p.Npad * 16;
    unsigned boff[NPW];   // byte offsets of this lane inside a (position, channel group) slice of U (see wino_fwd_kernel)
#pragma unroll
    for (int n = 0; n < NPW; ++n) boff[n] = 4u * (unsigned)(li * 16 + 4 * q + min(ntile0 + n, ntt - 1) * 256);
    const __amdgpu_buffer_rsrc_t urs = tmg_make_rsrc(p.U, 16u * (unsigned)pos_stride * 4u);
    int cm = 0, tm = blockIdx.x;
    f32x4 bvr[NPW];       // bias in registers, Y starts from it: no load between the epilogue's stores (see wino_fwd_kernel)
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        const int n0 = (ntile0 + n) * 16 + 4 * q;
        const float* bp = (p.bias != nullptr && ntile0 + n < ntt && n0 < p.Cout) ? p.bias + n0 : tmg_zero_page;
        const float4 b4 = *reinterpret_cast<const float4*>(bp);
        bvr[n] = (f32x4){b4.x, b4.y, b4.z, b4.w};
    }
    f32x4 Y[4][2][NPW];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NPW; ++n) Y[o][m][n] = bvr[n];
    f32x4 T0[2][NPW], T1[2][NPW];
    float4 bfr[4][2][NPW];
    for (int k = 0; k < nst; ++k) {
        const int c0 = cm * KC;
        const int kgn = min(KC, p.Cin_pad - c0) >> 4;
        const float* ub = p.U + (size_t)(c0 >> 4) * kb_stride;
#define TMG_WN_LOADB(R, UB, KGN, POS)                                                                                 \
        {                                                                                                             \
            const float* up_ = (UB) + (size_t)(POS) * pos_stride;                                                     \
            _Pragma("unroll") for (int n = 0; n < NPW; ++n) {                                                         \
                TMG_WN_ULOAD(bfr[R][0][n], up_, boff[n])                                                              \
                TMG_WN_ULOAD(bfr[R][1][n], up_ + (size_t)((KGN) - 1) * kb_stride, boff[n])                            \
            }                                                                                                         \
        }
        if (k == 0) { TMG_WN_LOADB(0, ub, kgn, 0) TMG_WN_LOADB(1, ub, kgn, 1) TMG_WN_LOADB(2, ub, kgn, 2) }
        __syncthreads();           // stage k has been produced
        {
            const float4* v4 = reinterpret_cast<const float4*>(lds + (k & 1) * VBUF) + (li * VS + 4 * q) / 4;
            const int c0n = (cm + 1 == nchunks) ? 0 : c0 + KC;
            const int kgn_n = min(KC, p.Cin_pad - c0n) >> 4;
            const float* ubn = p.U + (size_t)(c0n >> 4) * kb_stride;
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) {
                const int R = pos & 3;
                __builtin_amdgcn_sched_barrier(0);
                if (pos + 3 < 16) TMG_WN_LOADB((pos + 3) & 3, ub, kgn, pos + 3)
                else TMG_WN_LOADB((pos + 3) & 3, ubn, kgn_n, pos + 3 - 16)
                float4 af[2][2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    af[0][m] = v4[(pos * VPL + m * 16 * VS) / 4];
                    af[1][m] = v4[(pos * VPL + m * 16 * VS + 16) / 4];
                }
                __builtin_amdgcn_sched_barrier(0);
                TMG_WN_POSITION
            }
        }
#undef TMG_WN_LOADB
        if (cm + 1 == nchunks) {
            int t_ = tm;
            const int tx_ = t_ % p.tiles_x; t_ /= p.tiles_x;
            const int ty_ = t_ % p.tiles_y;
            const int b_ = t_ / p.tiles_y;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int wt = 16 * m + li;
                const int oyb = ty_ * TH + 2 * (wt >> 3), oxb = tx_ * TW + 2 * (wt & 7);
#pragma unroll
                for (int n = 0; n < NPW; ++n) {
                    const int n0 = (ntile0 + n) * 16 + 4 * q;
                    if (ntile0 + n < ntt && n0 < p.Cout) {
                        int nl = n0;
                        TMG_PICK_OSEG(p.out, nl, optr, ostride, ooff)
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            const int oy = oyb + (o >> 1), ox = oxb + (o & 1);
                            if (oy < p.Hin && ox < p.Win) {
                                const unsigned opx = ((unsigned)b_ * (unsigned)p.Hin + (unsigned)oy) * (unsigned)p.Win + (unsigned)ox;
                                *reinterpret_cast<float4*>(optr + (size_t)opx * (unsigned)ostride + ooff + nl) =
                                    make_float4(Y[o][m][n][0], Y[o][m][n][1], Y[o][m][n][2], Y[o][m][n][3]);
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
                    for (int n = 0; n < NPW; ++n) Y[o][m][n] = bvr[n];
            cm = 0; tm += G;
        } else {
            ++cm;
        }
    }
}

// out = conv3x3_stride1(pad(act(in))) + bias with the Winograd operand of tmg_conv_wino_pack.
// dims = {B, H, W, Cin, Cout, relu_in, pad_replicate}; in_desc / out_desc = {stride, off, n} per segment (<= 3 each).
// Envelope: float4-addressable segments, Cin % 4 == 0, Cout % 4 == 0, Cout >= 64; returns -100 outside it (the caller
// uses tmg_conv_fwd).
extern "C" int tmg_conv_wino_fwd(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* U, const void* bias,
                                 void* const* out_ptrs, const int64_t* out_desc, int64_t nout, const int64_t* dims, hipStream_t st) {
    WinoP p;
    p.nseg = (int)nseg;
    if (p.nseg < 1 || p.nseg > TMG_MAX_IN_SEG || nout < 1 || nout > TMG_MAX_OUT_SEG) return -3;
    int csum = 0, osum = 0;
    bool ok = true;
    for (int i = 0; i < TMG_MAX_IN_SEG; ++i) p.in[i] = TmgSeg{nullptr, 0, 0, 0};
    for (int i = 0; i < p.nseg; ++i) {
        p.in[i] = TmgSeg{(const float*)in_ptrs[i], (int)in_desc[3 * i], (int)in_desc[3 * i + 1], (int)in_desc[3 * i + 2]};
        if (((p.in[i].stride | p.in[i].off | p.in[i].n) & 3) || (((uintptr_t)in_ptrs[i]) & 15)) ok = false;
        csum += p.in[i].n;
    }
    for (int i = 0; i < TMG_MAX_OUT_SEG; ++i) p.out[i] = TmgOSeg{nullptr, 0, 0, 0};
    for (int i = 0; i < (int)nout; ++i) {
        p.out[i] = TmgOSeg{(float*)out_ptrs[i], (int)out_desc[3 * i], (int)out_desc[3 * i + 1], (int)out_desc[3 * i + 2]};
        if (((p.out[i].stride | p.out[i].off | p.out[i].n) & 3) || (((uintptr_t)out_ptrs[i]) & 15)) ok = false;
        osum += p.out[i].n;
    }
    p.B = (int)dims[0]; p.Hin = (int)dims[1]; p.Win = (int)dims[2]; p.Cin = (int)dims[3]; p.Cout = (int)dims[4];
    p.relu_in = (int)dims[5]; p.pad_rep = (int)dims[6];
    if (csum != p.Cin || osum != p.Cout) return -3;
    if ((p.Cout & 3) || (p.Cin & 3) || p.Cout < 64) ok = false;
    if (bias && (((uintptr_t)bias) & 15)) ok = false;
    if (!ok) return -100;
    p.Cin_pad = (p.Cin + 15) & ~15;
    p.Npad = (p.Cout + 15) & ~15;
    p.U = (const float*)U; p.bias = (const float*)bias;
    p.tiles_x = (p.Win + 15) / 16; p.tiles_y = (p.Hin + 7) / 8;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    p.nchunks = (p.Cin_pad + 31) / 32;
    if (p.ntiles <= 0) return 0;
    if ((long long)p.B * p.Hin * p.Win >= (1LL << 31)) return -100;      // 32-bit pixel indices in the kernels
    const int ntt = p.Npad / 16;
    const int npw = ntt <= 8 ? 1 : 2;           // output-channel tiles per wave (see the kernel)
    const int gy = (ntt + 8 * npw - 1) / (8 * npw);
    int G = tmg_num_cus() / gy;
    if (G < 1) G = 1;
    if (G > p.ntiles) G = p.ntiles;
    const size_t lds_bytes = (size_t)(2 * 180 * 40 + 16 * 32 * 40) * sizeof(float);
    TmgProf prof(TMG_PROF_WINO, 2.0 * p.B * p.Hin * p.Win * (double)p.Cout * p.Cin * 9, st);   // algorithmic (direct) flops
    // Producer-wave form (round 6) for the contractions with <= 128 output channels (one n-tile per wave: 100 registers, twelve waves
    // fit): 0.592 -> 0.551 ms (40 -> 104 at 128^2), 0.157 -> 0.146, 0.051 -> 0.049 (profiles/r6_ab_wino_producer_waves.txt).  Two
    // n-tiles per wave need 221 registers with the four-deep U ring; at the 168 that twelve waves leave, a two-deep ring measured 2.375 ms
    // against 2.294 for the gate conv and a three-deep one spills: the wide shapes stay on wino_fwd_kernel<2>.  TMG_WINO_PC=0: off,
    // =2: the one-tile producer form for every shape (two blocks per pixel tile above 128 channels: 2.355 ms).
    static const int pc = getenv("TMG_WINO_PC") ? atoi(getenv("TMG_WINO_PC")) : 1;
    if ((pc == 1 && npw == 1) || pc == 2) {
        const size_t ldsp = (size_t)(2 * 16 * 32 * 40) * sizeof(float);      // two V buffers: all 160 KB of the CU
        const int gyp = (ntt + 7) / 8;
        int Gp = tmg_num_cus() / gyp;
        if (Gp < 1) Gp = 1;
        if (Gp > p.ntiles) Gp = p.ntiles;
        TMG_LDS_OPTIN((&wino_fwdp_kernel<1>));
        hipLaunchKernelGGL(wino_fwdp_kernel<1>, dim3(Gp, gyp, 1), dim3(768), ldsp, st, p);
        TMG_CHECK_LAUNCH();
        return 0;
    }
    if (npw == 1) {
        TMG_LDS_OPTIN((&wino_fwd_kernel<1>));
        hipLaunchKernelGGL(wino_fwd_kernel<1>, dim3(G, gy, 1), dim3(512), lds_bytes, st, p);
    } else {
        TMG_LDS_OPTIN((&wino_fwd_kernel<2>));
        hipLaunchKernelGGL(wino_fwd_kernel<2>, dim3(G, gy, 1), dim3(512), lds_bytes, st, p);
    }
    TMG_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================================================
// OPT-IN VARIANT (round 5): the same wide Winograd contraction with the 16 position GEMMs on the BF16 matrix pipe at fp32 accuracy.
// An fp32 value splits exactly into three bf16 values by truncation (8 + 8 + 8 significant bits); of the nine bf16 x bf16 products
// of two split values the six largest - a0 b0, a0 b1, a1 b0, a0 b2, a1 b1, a2 b0 - leave out terms of relative size <= 2^-23, the
// size of fp32 rounding.  Six v_mfma_f32_16x16x32_bf16 (fp32 accumulation, 16 pipe cycles, K = 32) then replace the 32
// v_mfma_f32_16x16x4_f32 (32 cycles each) of a position and 32-channel chunk: 96 pipe cycles instead of 1 024 per accumulator tile.
// Both Winograd transforms stay in fp32; the split happens when V is written to LDS and when U is packed.  Measured against fp64
// (tools/micro/wino_bf16x3.hip, the gate conv's shape): max-abs 6.7e-7 / rel-L2 2.0e-7, the fp32 kernel above 6.7e-7 / 2.2e-7.
// Structure, tile, staging, output transform and epilogue are wino_fwd_kernel's; what differs:
//   V in LDS   [16 pos][3 parts][4 k-quarters][32 tiles][8 bf16]: a B fragment of mfma_f32_16x16x32_bf16 (lane (li, q): tile li,
//              channels 8 q .. 8 q + 7 of the chunk) is one conflict-free ds_read_b128;
//   transform  a thread owns (tile, 8-channel group, row xi of the 4x4 position grid): 16 float4 reads of the raw patch, the
//              row / column passes, 32 values split into 3 parts (5.5 vector instructions per value), 12 ds_write_b128;
//   U operand  [16 pos][Cin_pad32 / 32][Npad / 16][3 parts][64 lanes][8 bf16] (tmg_conv_wino_pack3): an A fragment (lane (li, q): output
//              channel li of the tile, input channels 8 q .. 8 q + 7) is one 16-byte load per lane and part.
// Selected by tmg_ops.set_winograd_precision("bf16x3") / TMG_WINO_BF3=1 (tmg_hip.conv3x3_auto); the default stays the fp32 MFMA kernel.
// =================================================================================================================================
typedef __bf16 tmg_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void tmg_split3(float v, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned b0 = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(b0);
    const unsigned b1 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(b1);
    p0 = b0 >> 16; p1 = b1 >> 16; p2 = __float_as_uint(r2) >> 16;
}

// U = G g G^T per (output channel, input channel) in fp32, then split; same modes as wino_pack_kernel.
// one thread per (chunk, n-tile, lane, j): all 16 positions and 3 parts
__global__ void wino_pack3_kernel(const float* __restrict__ w, unsigned short* __restrict__ U, int Cout, int Cin, int K, int N, int nch, int ntt, int mode) {
    const int total = nch * ntt * 64 * 8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 7, lane = (i >> 3) & 63;
        const int r = i >> 9;
        const int nt = r % ntt, ch = r / ntt;
        const int n = 16 * nt + (lane & 15), k = 32 * ch + 8 * (lane >> 4) + j;
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float v = 0.f;
                if (k < K && n < N) v = mode == 0 ? w[((size_t)n * Cin + k) * 9 + a * 3 + b] : w[((size_t)k * Cin + n) * 9 + (2 - a) * 3 + (2 - b)];
                g[a][b] = v;
            }
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
            const float u[4] = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                unsigned pr[3];
                tmg_split3(u[b], pr[0], pr[1], pr[2]);
#pragma unroll
                for (int part = 0; part < 3; ++part) {
                    const size_t frag = (((size_t)(a * 4 + b) * nch + ch) * ntt + nt) * 3 + part;
                    U[(frag * 64 + lane) * 8 + j] = (unsigned short)pr[part];
                }
            }
        }
    }
}

// w: torch layout [Cout][Cin][3][3]; U: [16][Kpad32 / 32][Npad / 16][3][64][8] bf16 (= 16 * Kpad32 * Npad * 3 two-byte values).  Modes as
// tmg_conv_wino_pack.
extern "C" int tmg_conv_wino_pack3(const void* w, void* U, int64_t Cout, int64_t Cin, int64_t mode, int64_t nvalid, hipStream_t st) {
    const int K = mode == 0 ? (int)Cin : (int)Cout;
    const int N = mode == 0 ? (int)Cout : (int)(nvalid > 0 && nvalid < Cin ? nvalid : Cin);
    const int nch = (K + 31) / 32, ntt = (N + 15) / 16;
    const int total = nch * ntt * 512;
    const int blocks = (total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048;
    hipLaunchKernelGGL(wino_pack3_kernel, dim3(blocks), dim3(256), 0, st, (const float*)w, (unsigned short*)U, (int)Cout, (int)Cin, K, N, nch, ntt, (int)mode);
    TMG_CHECK_LAUNCH();
    return 0;
}

__device__ __forceinline__ unsigned tmg_pack_hi(float hi, float lo) {      // (bf16 trunc(hi) << 16) | bf16 trunc(lo)
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ tmg_bf16x8 tmg_as_bf(uint4 u) {
    union { uint4 u; tmg_bf16x8 b; } c;
    c.u = u;
    return c.b;
}

template <int NPW>
__global__ __launch_bounds__(512, 1) TMG_PACKED_F32 void wino_fwd3_kernel(WinoP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 512;
    constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, PP = PH * PW;   // output tile, raw patch (halo 1)
    constexpr int KC = 32, CS = KC + 8;        // channels per chunk, raw-patch pixel stride (words)
    constexpr int RAWW = PP * CS;              // words per raw buffer
    constexpr int VQ = 32 * 16;                // bytes of one [32 tiles][8 bf16] plane
    constexpr int VPART = 4 * VQ, VPOS = 3 * VPART;   // bytes of one part (4 k-quarters) / one position (3 parts)
    char* Vb = reinterpret_cast<char*>(lds + 2 * RAWW);    // [16][3][4][32][16 bytes]
    constexpr int UPI = (PP * (KC / 4) + NT - 1) / NT;   // raw float4 items per thread (3)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int ntt = p.Npad >> 4;
    const int nch = p.nchunks;
    const int ntile0 = (int)blockIdx.y * (8 * NPW) + NPW * wave;
    // U addressing (bytes): fragment = 1 KB; [pos][chunk][n-tile][part]
    const unsigned upart = 1024u, unt = 3u * upart, uch = (unsigned)ntt * unt, upos = (unsigned)nch * uch;
    unsigned boff[NPW];
#pragma unroll
    for (int n = 0; n < NPW; ++n) boff[n] = (unsigned)min(ntile0 + n, ntt - 1) * unt + (unsigned)lane * 16u;
    const char* Ub = reinterpret_cast<const char*>(p.U);

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

    // ---- transform mapping: item = (Winograd tile tt, channel octet tc8, row xi of the 4x4 position grid) -------------------------------
    const int tc8 = tid & 3, tt = (tid >> 2) & 31, txi = tid >> 7;
    const int tty = tt >> 3, ttx = tt & 7;
    const int traw = ((2 * tty) * PW + 2 * ttx) * CS + 8 * tc8;   // word offset of patch pixel (0, 0) of the tile in a raw buffer
    // rows of the patch combined for row xi of B^T d:  xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const int tra = (txi == 0) ? 0 : (txi == 2 ? 2 : 1), trb = (txi == 0) ? 2 : (txi == 1 ? 2 : (txi == 2 ? 1 : 3));
    const float tsb = (txi == 1) ? 1.f : -1.f;

    const int G = gridDim.x;
    const int nmine = (int)blockIdx.x < p.ntiles ? (p.ntiles - (int)blockIdx.x + G - 1) / G : 0;
    const int nst = nmine * nch;
    int ci = 0, cc = 0, cm = 0;
    int ti = blockIdx.x, tm = blockIdx.x;

    f32x4 bvr[NPW];       // bias in registers, Y starts from it: no load between the epilogue's stores (see wino_fwd_kernel)
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        const int n0 = (ntile0 + n) * 16 + 4 * q;
        const float* bp = (p.bias != nullptr && ntile0 + n < ntt && n0 < p.Cout) ? p.bias + n0 : tmg_zero_page;
        const float4 b4 = *reinterpret_cast<const float4*>(bp);
        bvr[n] = (f32x4){b4.x, b4.y, b4.z, b4.w};
    }
    f32x4 Y[4][2][NPW];   // [output pixel of the 2x2 tile][m-tile][n-tile]
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NPW; ++n) Y[o][m][n] = bvr[n];

    // U fragments run in STEPS of (position, n-tile) - 32 or 16 per stage, 12 MFMAs each - through a ring of four slots, three steps
    // ahead of their MFMAs (slot = step & 3: the stage length is a multiple of four, one unrolled body).  [A ring of whole positions
    // needed 48 registers for ONE position of lead at two n-tiles per wave, 72 for two: spilled.]
    constexpr int NSTEP = 16 * NPW, LD = 3;
    uint4 bfr[4][3];
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
            if (++cc == nch) cc = 0;
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
            const unsigned tbv = (unsigned)b_ * (unsigned)(p.Hin * p.Win);
#pragma unroll
            for (int u = 0; u < UPI; ++u) {
                const int iy = iy0 + (int)(pyx[u] >> 16), ix = ix0 + (int)(pyx[u] & 0xffffu);
                const int iyc = min(max(iy, 0), p.Hin - 1), ixc = min(max(ix, 0), p.Win - 1);
                const bool oob = !p.pad_rep && (iy != iyc || ix != ixc);
                const float* a_ = (oob || ppix0 + u * 64 >= PP) ? tmg_zero_page : tptr + (size_t)(tbv + (unsigned)iyc * (unsigned)p.Win + (unsigned)ixc) * (unsigned)tss;
                // streamed once: non-temporal, so that the activations do not push the U operand (re-read by every tile) out of L2
                {
                    const f32x4 nt_ = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a_));
                    pv[u] = make_float4(nt_[0], nt_[1], nt_[2], nt_[3]);
                }
            }
            if (++ci == nch) { ci = 0; ti += G; }
        }
        if (k >= 0) {
            // ---- input transform of stage k: V = B^T d B in fp32, split into three bf16 parts on the way to LDS -----------------------------
            {
                const float* rb = lds + (k & 1) * RAWW + traw;
                float t[4][8];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float4 da = *reinterpret_cast<const float4*>(rb + (tra * PW + c) * CS + 4 * e);
                        const float4 db = *reinterpret_cast<const float4*>(rb + (trb * PW + c) * CS + 4 * e);
                        t[c][4 * e + 0] = da.x + tsb * db.x; t[c][4 * e + 1] = da.y + tsb * db.y;
                        t[c][4 * e + 2] = da.z + tsb * db.z; t[c][4 * e + 3] = da.w + tsb * db.w;
                    }
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) {
                    float v[8], r1[8], r2[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        v[j] = nu == 0 ? t[0][j] - t[2][j] : (nu == 1 ? t[1][j] + t[2][j] : (nu == 2 ? t[2][j] - t[1][j] : t[1][j] - t[3][j]));
                        r1[j] = v[j] - __uint_as_float(__float_as_uint(v[j]) & 0xffff0000u);
                        r2[j] = r1[j] - __uint_as_float(__float_as_uint(r1[j]) & 0xffff0000u);
                    }
                    char* vp = Vb + (4 * txi + nu) * VPOS + tc8 * VQ + tt * 16;
                    *reinterpret_cast<uint4*>(vp) = make_uint4(tmg_pack_hi(v[1], v[0]), tmg_pack_hi(v[3], v[2]), tmg_pack_hi(v[5], v[4]), tmg_pack_hi(v[7], v[6]));
                    *reinterpret_cast<uint4*>(vp + VPART) = make_uint4(tmg_pack_hi(r1[1], r1[0]), tmg_pack_hi(r1[3], r1[2]), tmg_pack_hi(r1[5], r1[4]), tmg_pack_hi(r1[7], r1[6]));
                    *reinterpret_cast<uint4*>(vp + 2 * VPART) = make_uint4(tmg_pack_hi(r2[1], r2[0]), tmg_pack_hi(r2[3], r2[2]), tmg_pack_hi(r2[5], r2[4]), tmg_pack_hi(r2[7], r2[6]));
                }
            }
            __syncthreads();
            // ---- 16 position GEMMs over this chunk (6 bf16 MFMAs per accumulator tile), output transform folded in ---------------------
            {
                const char* ub = Ub + (size_t)((unsigned)cm * uch);
                const char* vb = Vb + q * VQ + li * 16;
#define TMG_W3_LOADB(ST)                                                                                              \
                {                                                                                                     \
                    const char* up_ = ((ST) < NSTEP ? ub : ubn) + (unsigned)(((ST) % NSTEP) / NPW) * upos + boff[(ST) % NPW]; \
                    bfr[(ST) & 3][0] = *reinterpret_cast<const uint4*>(up_);                                          \
                    bfr[(ST) & 3][1] = *reinterpret_cast<const uint4*>(up_ + 1024);                                   \
                    bfr[(ST) & 3][2] = *reinterpret_cast<const uint4*>(up_ + 2048);                                   \
                }
                // operand of the next stage (the next chunk of this tile, or chunk 0 of the next tile: every tile uses the same U)
                const char* ubn = Ub + (size_t)((unsigned)((cm + 1 == nch) ? 0 : cm + 1) * uch);
                if (k == 0) { TMG_W3_LOADB(0) TMG_W3_LOADB(1) TMG_W3_LOADB(2) }
                // two accumulator sets: the output-transform additions of position p - 1 (vector ALU) sit between the MFMAs of position
                // p - with six 16-cycle MFMAs per accumulator tile a position is only 24 x 16 pipe cycles, and additions that wait for
                // their own position's last MFMA (as in the fp32 kernel, where a position is 1 024 cycles) would be a quarter of it
                f32x4 acc[2][2][NPW];
#define TMG_W3_YUPD(PPOS, M, N)                                                                                       \
                {                                                                                                     \
                    const int xi_ = (PPOS) >> 2, nu_ = (PPOS) & 3;                                                    \
                    _Pragma("unroll") for (int oy = 0; oy < 2; ++oy) _Pragma("unroll") for (int ox = 0; ox < 2; ++ox) { \
                        const int ay = oy == 0 ? (xi_ < 3 ? 1 : 0) : (xi_ == 0 ? 0 : (xi_ == 1 ? 1 : -1));            \
                        const int ax = ox == 0 ? (nu_ < 3 ? 1 : 0) : (nu_ == 0 ? 0 : (nu_ == 1 ? 1 : -1));            \
                        const int cf = ay * ax;                                                                       \
                        if (cf != 0) {                                                                                \
                            f32x4& y_ = Y[oy * 2 + ox][M][N];                                                         \
                            if (cf > 0) y_ += acc[(PPOS) & 1][M][N];                                                  \
                            else y_ -= acc[(PPOS) & 1][M][N];                                                         \
                            /* pin the update HERE (the optimiser sinks the additions below the last position otherwise) */ \
                            asm volatile("" : "+v"(y_));                                                              \
                        }                                                                                             \
                    }                                                                                                 \
                }
#pragma unroll
                for (int pos = 0; pos < 16; ++pos) {
                    __builtin_amdgcn_sched_barrier(0);
                    uint4 af[2][3];
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        af[m][0] = *reinterpret_cast<const uint4*>(vb + pos * VPOS + m * 256);
                        af[m][1] = *reinterpret_cast<const uint4*>(vb + pos * VPOS + VPART + m * 256);
                        af[m][2] = *reinterpret_cast<const uint4*>(vb + pos * VPOS + 2 * VPART + m * 256);
                    }
#pragma unroll
                    for (int n = 0; n < NPW; ++n) {
                        const int st_ = pos * NPW + n;
                        __builtin_amdgcn_sched_barrier(0);
                        TMG_W3_LOADB(st_ + LD)
                        __builtin_amdgcn_sched_barrier(0);
                        const tmg_bf16x8 a0 = tmg_as_bf(bfr[st_ & 3][0]), a1 = tmg_as_bf(bfr[st_ & 3][1]), a2 = tmg_as_bf(bfr[st_ & 3][2]);
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const tmg_bf16x8 b0 = tmg_as_bf(af[m][0]), b1 = tmg_as_bf(af[m][1]), b2 = tmg_as_bf(af[m][2]);
                            f32x4 c_ = (f32x4){0.f, 0.f, 0.f, 0.f};
                            c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, c_, 0, 0, 0);
                            c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2, c_, 0, 0, 0);
                            c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c_, 0, 0, 0);
                            if (pos > 0) TMG_W3_YUPD(pos > 0 ? pos - 1 : 0, m, n)
                            c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, c_, 0, 0, 0);
                            c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, c_, 0, 0, 0);
                            c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, c_, 0, 0, 0);
                            acc[pos & 1][m][n] = c_;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NPW; ++n) TMG_W3_YUPD(15, m, n)
#undef TMG_W3_YUPD
#undef TMG_W3_LOADB
            }
            if (cm + 1 == nch) {
                // ---- epilogue: lane (li, q) holds channels 4 q .. 4 q + 3 (of each n-tile) of Winograd tile 16 m + li -------------
                int t_ = tm;
                const int tx_ = t_ % p.tiles_x; t_ /= p.tiles_x;
                const int ty_ = t_ % p.tiles_y;
                const int b_ = t_ / p.tiles_y;
                // (opaque copies per epilogue: the tile-invariant output offsets built on them are otherwise hoisted out of the stage loop
                // and spilled at two n-tiles per wave)
                int lie = li, qe = q;
                asm volatile("" : "+v"(lie), "+v"(qe));
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int wt = 16 * m + lie;
                    const int oyb = ty_ * TH + 2 * (wt >> 3), oxb = tx_ * TW + 2 * (wt & 7);
#pragma unroll
                    for (int n = 0; n < NPW; ++n) {
                        const int n0 = (ntile0 + n) * 16 + 4 * qe;
                        if (ntile0 + n < ntt && n0 < p.Cout) {
                            int nl = n0;
                            TMG_PICK_OSEG(p.out, nl, optr, ostride, ooff)
#pragma unroll
                            for (int o = 0; o < 4; ++o) {
                                const int oy = oyb + (o >> 1), ox = oxb + (o & 1);
                                if (oy < p.Hin && ox < p.Win) {
                                    const unsigned opx = ((unsigned)b_ * (unsigned)p.Hin + (unsigned)oy) * (unsigned)p.Win + (unsigned)ox;
                                    __builtin_nontemporal_store(Y[o][m][n],
                                                                reinterpret_cast<f32x4*>(optr + (size_t)opx * (unsigned)ostride + ooff + nl));
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
                        for (int n = 0; n < NPW; ++n) Y[o][m][n] = bvr[n];
                cm = 0; tm += G;
            } else {
                ++cm;
            }
        }
        __syncthreads();   // V and the raw buffer just read are rewritten next round; the raw buffer just written is complete
    }
}

// tmg_conv_wino_fwd with the operand of tmg_conv_wino_pack3 (same arguments, same envelope).
extern "C" int tmg_conv_wino_fwd3(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* U, const void* bias,
                                  void* const* out_ptrs, const int64_t* out_desc, int64_t nout, const int64_t* dims, hipStream_t st) {
    WinoP p;
    p.nseg = (int)nseg;
    if (p.nseg < 1 || p.nseg > TMG_MAX_IN_SEG || nout < 1 || nout > TMG_MAX_OUT_SEG) return -3;
    int csum = 0, osum = 0;
    bool ok = true;
    for (int i = 0; i < TMG_MAX_IN_SEG; ++i) p.in[i] = TmgSeg{nullptr, 0, 0, 0};
    for (int i = 0; i < p.nseg; ++i) {
        p.in[i] = TmgSeg{(const float*)in_ptrs[i], (int)in_desc[3 * i], (int)in_desc[3 * i + 1], (int)in_desc[3 * i + 2]};
        if (((p.in[i].stride | p.in[i].off | p.in[i].n) & 3) || (((uintptr_t)in_ptrs[i]) & 15)) ok = false;
        csum += p.in[i].n;
    }
    for (int i = 0; i < TMG_MAX_OUT_SEG; ++i) p.out[i] = TmgOSeg{nullptr, 0, 0, 0};
    for (int i = 0; i < (int)nout; ++i) {
        p.out[i] = TmgOSeg{(float*)out_ptrs[i], (int)out_desc[3 * i], (int)out_desc[3 * i + 1], (int)out_desc[3 * i + 2]};
        if (((p.out[i].stride | p.out[i].off | p.out[i].n) & 3) || (((uintptr_t)out_ptrs[i]) & 15)) ok = false;
        osum += p.out[i].n;
    }
    p.B = (int)dims[0]; p.Hin = (int)dims[1]; p.Win = (int)dims[2]; p.Cin = (int)dims[3]; p.Cout = (int)dims[4];
    p.relu_in = (int)dims[5]; p.pad_rep = (int)dims[6];
    if (csum != p.Cin || osum != p.Cout) return -3;
    // (FEW output channels - the input-gradient shapes 256 -> 40, 240 -> 32 .. of wino_nn_kernel - were measured on this kernel with
    // three of eight waves live: 0.47-0.69x of wino_nn_kernel, the per-chunk transform / staging / barriers of 8-60 chunks dominate)
    if ((p.Cout & 3) || (p.Cin & 3) || p.Cout < 64) ok = false;
    if (bias && (((uintptr_t)bias) & 15)) ok = false;
    if (!ok) return -100;
    p.Cin_pad = (p.Cin + 31) & ~31;
    p.Npad = (p.Cout + 15) & ~15;
    p.U = (const float*)U; p.bias = (const float*)bias;
    p.tiles_x = (p.Win + 15) / 16; p.tiles_y = (p.Hin + 7) / 8;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    p.nchunks = p.Cin_pad / 32;
    if (p.ntiles <= 0) return 0;
    if ((long long)p.B * p.Hin * p.Win >= (1LL << 31)) return -100;
    const int ntt = p.Npad / 16;
    const int npw = ntt <= 8 ? 1 : 2;
    const int gy = (ntt + 8 * npw - 1) / (8 * npw);
    int G = tmg_num_cus() / gy;
    if (G < 1) G = 1;
    if (G > p.ntiles) G = p.ntiles;
    const size_t lds_bytes = (size_t)(2 * 180 * 40) * sizeof(float) + (size_t)16 * 3 * 4 * 512;
    TmgProf prof(TMG_PROF_WINO, 2.0 * p.B * p.Hin * p.Win * (double)p.Cout * p.Cin * 9, st);   // algorithmic (direct) flops
    if (npw == 1) {
        TMG_LDS_OPTIN((&wino_fwd3_kernel<1>));
        hipLaunchKernelGGL(wino_fwd3_kernel<1>, dim3(G, gy, 1), dim3(512), lds_bytes, st, p);
    } else {
        TMG_LDS_OPTIN((&wino_fwd3_kernel<2>));
        hipLaunchKernelGGL(wino_fwd3_kernel<2>, dim3(G, gy, 1), dim3(512), lds_bytes, st, p);
    }
    TMG_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================================================
// Few output channels (<= 48), many input channels: the input gradients of the contractions above (4R -> Cin gate gradient, 15 C
// -> Cc conditioning gradient) and the ConvLSTM block's narrow convs.  With only NTN <= 4 output-channel tiles there is no work to
// give eight waves along N; instead the 16 Winograd positions are dealt to the waves, two each (wave w: xi = w >> 1, nu = 2 (w & 1)
// and + 1), every wave running the whole input-channel contraction for its positions on the same 32 Winograd tiles:
//   * no V buffer: a lane builds its V fragment (4 channels of one tile at one position) from the raw patch in LDS - a position of
//     B^T d B has four non-zero terms, and the wave's two positions share the row combination: 6 float4 reads + 5 float4 adds;
//   * no per-chunk output transform: the accumulators M_pos run over ALL input channels of the tile;
//   * after the last chunk the waves drop M_pos into LDS and all 512 threads apply A^T M A, bias / ReLU, and store.
// =================================================================================================================================
struct WinoNP {
    TmgSeg in[TMG_MAX_IN_SEG];
    int nseg;
    int B, Hin, Win;
    int Cin, Cin_pad, Cout, Npad;
    const float* U; const float* bias;
    int relu_in, pad_rep, relu_out;
    TmgOSeg out[TMG_MAX_OUT_SEG];
    int tiles_x, tiles_y, ntiles, nchunks;
};

// SKEW: the eight waves of the block run in step (one barrier per round), so a staging phase in front of the MFMA groups is a phase
// during which nobody feeds the matrix pipe (ablation builds, round 4: without commit + issue the 240 -> 32 contraction at 128^2 runs
// in 0.74 ms instead of 0.94).  With SKEW the two waves that share a SIMD (w and w + 4) stage at OPPOSITE ends of the round: waves 0-3
// commit / issue first and multiply afterwards, waves 4-7 multiply first and commit / issue before the barrier - each wave's staging
// instructions issue while its SIMD partner's MFMAs keep the pipe busy.  Same LDS hazards as before: the stage after this one is
// committed to the buffer nobody reads in this round, at any point between the two barriers.  Measured (tools/bench_wino.py, one box,
// TMG_WN_NOSKEW=1 for the form without): 256 -> 40 at 128^2 1.259 -> 1.173 ms, 240 -> 32 0.935 -> 0.863, 480 -> 32 at 64^2 0.428 -> 0.408;
// pairing the waves as (w, w ^ 1) instead: 1.222 / 0.902 (waves w and w + 4 are the ones that share a SIMD).  The same change in
// wino_fwd_kernel, whose rounds also hold the all-thread input transform, lost 2-5 % and is not in the tree.
template <int NTN, int SKEW = 1>
__global__ __launch_bounds__(512, 1) void wino_nn_kernel(WinoNP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 512;
    constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, PP = PH * PW;
    constexpr int KC = 32, CS = KC + 8, RAWW = PP * CS;
    constexpr int NC = NTN * 16 + (NTN < 3 ? 4 : 0);   // row stride (words) of the M buffer [16 pos][32 tiles][NC] (3 tiles: no room to pad)
    float* Mb = lds + 2 * RAWW;
    constexpr int UPI = (PP * (KC / 4) + NT - 1) / NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int KB = p.Cin_pad >> 4;
    const size_t kb_stride = (size_t)p.Npad * 16, pos_stride = (size_t)KB * p.Npad * 16;
    const int xi = wave >> 1, par = wave & 1;
    const int pos0 = xi * 4 + 2 * par;
    const unsigned boff = 4u * (unsigned)(li * 16 + 4 * q);   // lane part (bytes) of a U fragment address; the rest is scalar (tmg_bload4)
    const __amdgpu_buffer_rsrc_t urs = tmg_make_rsrc(p.U, 16u * (unsigned)pos_stride * 4u);
    // row combination of this wave's xi: u = d[ra] + sg d[rb]   (B^T rows: r0 - r2, r1 + r2, r2 - r1, r1 - r3)
    const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1), rb = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    const float sg = xi == 1 ? 1.f : -1.f;
    // column pass without selects: patch columns in the order (c0, c1, c2) for nu = (0, 1), (c2, c3, c1) for nu = (2, 3): both position
    // pairs are (u0 - u2, u1 s1 + u2), s1 = +1 / -1 (wave-uniform offsets and a sign instead of computing both variants)
    const int cc[3] = {par ? 2 : 0, par ? 3 : 1, par ? 1 : 2};
    const float s1 = par ? -1.f : 1.f;
    // per-lane raw-patch word offsets of the tile's pixel (row ra / rb, column 0) for the two m-tiles, + 4 q (channel quad)
    int offa[2], offb[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int wt = 16 * m + li, ty = wt >> 3, tx = wt & 7;
        offa[m] = ((2 * ty + ra) * PW + 2 * tx) * CS + 4 * q;
        offb[m] = ((2 * ty + rb) * PW + 2 * tx) * CS + 4 * q;
    }

    const int pc4 = tid & 7, ppix0 = tid >> 3;
    unsigned pyx[UPI];
#pragma unroll
    for (int u = 0; u < UPI; ++u) {
        const int pix = min(ppix0 + u * 64, PP - 1);
        const int py = pix / PW, px = pix - py * PW;
        pyx[u] = ((unsigned)py << 16) | (unsigned)px;
    }
    float4 pv[UPI];
    // Pixel part of the staging addresses: the same for every 32-channel chunk of a pixel tile (8-60 chunks here), so it is computed when
    // the issue cursor enters a tile and kept (one register per item + a mask of the zero-padded ones): ~9 vector instructions per load
    // and chunk less - this kernel issues only 64-96 MFMAs per wave and chunk, and vector cycles add to matrix cycles.
    // (three n-tiles: the kernel sits at the 256-register limit and measured 1 % slower with the four extra registers live across
    // the chunks - there the offsets are recomputed per chunk as before)
    constexpr bool PXC = NTN <= 2;
    unsigned pxo[UPI];
    unsigned pzm = 0;

    const int G = gridDim.x;
    const int nmine = (int)blockIdx.x < p.ntiles ? (p.ntiles - (int)blockIdx.x + G - 1) / G : 0;
    const int nchunks = p.nchunks, nst = nmine * nchunks;
    int ci = 0, cm = 0;
    int ti = blockIdx.x, tm = blockIdx.x;

    f32x4 acc[2][2][NTN];   // [position of the wave][m-tile][n-tile]
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NTN; ++n) acc[e][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // U fragments (2 positions x NTN tiles per 16-channel group) in a ring over the groups, RD - 1 groups ahead of the MFMAs: a group
    // is only 16 NTN MFMAs per wave, one group of lead does not cover the L2 latency (three tiles: registers allow no more)
    constexpr int RD = NTN <= 2 ? 4 : 2;
    float4 bq[RD][2][NTN];
    const int ngrp = 2 * nchunks;   // groups per tile (those past Cin_pad read a clamped slice against zero V fragments)
#define TMG_WN_LOADG(SLOT, GI)                                                                                        \
    {                                                                                                                 \
        const int kb_ = min((GI), KB - 1);                                                                            \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                               \
            const unsigned so_ = 4u * (unsigned)((size_t)(pos0 + e) * pos_stride + (size_t)kb_ * kb_stride);         \
            _Pragma("unroll") for (int n = 0; n < NTN; ++n) bq[SLOT][e][n] = tmg_bload4(urs, boff, so_ + 1024u * n);  \
        }                                                                                                             \
    }
    if (nst > 0) {
#pragma unroll
        for (int d = 0; d < RD - 1; ++d) TMG_WN_LOADG(d, d % ngrp)
    }

    const bool late = SKEW && wave >= 4;      // stages at the end of the round (see above)
    // The epilogue item of this thread (Winograd tile, channel quad) is the same for every pixel tile (32 QN <= 512 items): its bias quad
    // is loaded once, here - a load inside the epilogue is waited for with vmcnt(0), which also drains the U ring and the patch loads
    // in flight for the next stages.
    constexpr int QN = NTN * 4;   // channel quads
    float4 bvq = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const int n0 = 4 * (tid % QN);
        const float* bp = (p.bias != nullptr && tid < 32 * QN && n0 < p.Cout) ? p.bias + n0 : tmg_zero_page;
        bvq = *reinterpret_cast<const float4*>(bp);
    }
#define TMG_WN_STAGE \
        if (k >= -1 && k + 1 < nst) { \
            float* rb_ = lds + ((k + 1) & 1) * RAWW; \
            _Pragma("unroll") for (int u = 0; u < UPI; ++u) { \
                if (ppix0 + u * 64 < PP) { \
                    float4 v = pv[u]; \
                    if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); } \
                    *reinterpret_cast<float4*>(rb_ + (ppix0 + u * 64) * CS + 4 * pc4) = v; \
                } \
            } \
        } \
        if (k + 2 < nst) { \
            int t_ = ti; \
            const int tx_ = t_ % p.tiles_x; t_ /= p.tiles_x; \
            const int ty_ = t_ % p.tiles_y; \
            const int b_ = t_ / p.tiles_y; \
            const int iy0 = ty_ * TH - 1, ix0 = tx_ * TW - 1; \
            const float* tptr = tmg_zero_page; \
            int tss = 0; \
            { \
                int cl = ci * KC + 4 * pc4; \
                if (cl < p.Cin) { \
                    const float* sp = p.in[0].p; \
                    int ss = p.in[0].stride, so = p.in[0].off; \
                    if (cl >= p.in[0].n) { \
                        cl -= p.in[0].n; \
                        sp = p.in[1].p; ss = p.in[1].stride; so = p.in[1].off; \
                        if (cl >= p.in[1].n) { \
                            cl -= p.in[1].n; \
                            sp = p.in[2].p; ss = p.in[2].stride; so = p.in[2].off; \
                        } \
                    } \
                    tptr = sp + so + cl; \
                    tss = ss; \
                } \
            } \
            if (ci == 0 || !PXC) { \
                const unsigned tbv = (unsigned)b_ * (unsigned)(p.Hin * p.Win); \
                pzm = 0; \
                _Pragma("unroll") for (int u = 0; u < UPI; ++u) { \
                    const int iy = iy0 + (int)(pyx[u] >> 16), ix = ix0 + (int)(pyx[u] & 0xffffu); \
                    const int iyc = min(max(iy, 0), p.Hin - 1), ixc = min(max(ix, 0), p.Win - 1); \
                    const bool oob = !p.pad_rep && (iy != iyc || ix != ixc); \
                    pxo[u] = tbv + (unsigned)iyc * (unsigned)p.Win + (unsigned)ixc; \
                    pzm |= ((oob || ppix0 + u * 64 >= PP) ? 1u : 0u) << u; \
                } \
            } \
            _Pragma("unroll") for (int u = 0; u < UPI; ++u) { \
                const float* a_ = ((pzm >> u) & 1u) ? tmg_zero_page : tptr + (size_t)pxo[u] * (unsigned)tss; \
                pv[u] = *reinterpret_cast<const float4*>(a_); \
            } \
            if (++ci == nchunks) { ci = 0; ti += G; } \
        }
    for (int k = -2; k < nst; ++k) {
        if (!late || k < 0) { TMG_WN_STAGE }
        if (k >= 0) {
            const float* rbuf = lds + (k & 1) * RAWW;
#define TMG_WN_COMPUTE(PH)                                                                                            \
            {                                                                                                         \
                _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                       \
                    const int SL = (2 * (PH) + g) & (RD - 1), SN = (2 * (PH) + g + RD - 1) & (RD - 1);   /* constants after unrolling */ \
                    {                                                                                                 \
                        int gi_ = 2 * cm + g + RD - 1;                                                                \
                        if (gi_ >= ngrp) gi_ -= ngrp;                                                                 \
                        if (gi_ >= ngrp) gi_ -= ngrp;                                                                 \
                        TMG_WN_LOADG(SN, gi_)                                                                         \
                    }                                                                                                 \
                    float4 va[2], vb[2];   /* V fragments of the wave's two positions, per m-tile */                  \
                    _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                                   \
                        float4 u_[3];                                                                                 \
                        _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                               \
                            const float4 da = *reinterpret_cast<const float4*>(rbuf + offa[m] + cc[c] * CS + g * 16); \
                            const float4 db = *reinterpret_cast<const float4*>(rbuf + offb[m] + cc[c] * CS + g * 16); \
                            u_[c] = make_float4(fmaf(db.x, sg, da.x), fmaf(db.y, sg, da.y), fmaf(db.z, sg, da.z), fmaf(db.w, sg, da.w)); \
                        }                                                                                             \
                        va[m] = make_float4(u_[0].x - u_[2].x, u_[0].y - u_[2].y, u_[0].z - u_[2].z, u_[0].w - u_[2].w); \
                        vb[m] = make_float4(fmaf(u_[1].x, s1, u_[2].x), fmaf(u_[1].y, s1, u_[2].y), fmaf(u_[1].z, s1, u_[2].z), fmaf(u_[1].w, s1, u_[2].w)); \
                    }                                                                                                 \
                    __builtin_amdgcn_sched_barrier(0);                                                                \
                    _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < NTN; ++n) {   \
                        acc[0][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[SL][0][n].x, va[m].x, acc[0][m][n], 0, 0, 0); \
                        acc[1][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[SL][1][n].x, vb[m].x, acc[1][m][n], 0, 0, 0); \
                    }                                                                                                 \
                    _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < NTN; ++n) {   \
                        acc[0][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[SL][0][n].y, va[m].y, acc[0][m][n], 0, 0, 0); \
                        acc[1][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[SL][1][n].y, vb[m].y, acc[1][m][n], 0, 0, 0); \
                    }                                                                                                 \
                    _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < NTN; ++n) {   \
                        acc[0][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[SL][0][n].z, va[m].z, acc[0][m][n], 0, 0, 0); \
                        acc[1][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[SL][1][n].z, vb[m].z, acc[1][m][n], 0, 0, 0); \
                    }                                                                                                 \
                    _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < NTN; ++n) {   \
                        acc[0][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[SL][0][n].w, va[m].w, acc[0][m][n], 0, 0, 0); \
                        acc[1][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[SL][1][n].w, vb[m].w, acc[1][m][n], 0, 0, 0); \
                    }                                                                                                 \
                    __builtin_amdgcn_sched_barrier(0);                                                                \
                }                                                                                                     \
            }
            if (k & 1) TMG_WN_COMPUTE(1) else TMG_WN_COMPUTE(0)
#undef TMG_WN_COMPUTE
            if (late) { TMG_WN_STAGE }
            if (cm + 1 == nchunks) {
                // ---- M_pos -> LDS, then A^T M A over the positions by all threads ----------------------------------------------
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < NTN; ++n) {
                            *reinterpret_cast<float4*>(Mb + ((pos0 + e) * 32 + 16 * m + li) * NC + 16 * n + 4 * q) =
                                make_float4(acc[e][m][n][0], acc[e][m][n][1], acc[e][m][n][2], acc[e][m][n][3]);
                            acc[e][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                __syncthreads();
                int t_ = tm;
                const int tx_ = t_ % p.tiles_x; t_ /= p.tiles_x;
                const int ty_ = t_ % p.tiles_y;
                const int b_ = t_ / p.tiles_y;
                static_assert(32 * QN <= NT, "one epilogue item per thread");
                for (int it = tid; it < 32 * QN; it += NT) {
                    const int wt = it / QN, c4 = it - wt * QN;
                    const int n0 = 4 * c4;
                    if (n0 >= p.Cout) continue;
                    const float* mp_ = Mb + wt * NC + n0;
                    float4 z[2][4];   // [oy][nu]: row pass  z0 = M0 + M1 + M2, z1 = M1 - M2 - M3
#pragma unroll
                    for (int nu = 0; nu < 4; ++nu) {
                        const float4 m0 = *reinterpret_cast<const float4*>(mp_ + (0 * 4 + nu) * 32 * NC);
                        const float4 m1 = *reinterpret_cast<const float4*>(mp_ + (1 * 4 + nu) * 32 * NC);
                        const float4 m2 = *reinterpret_cast<const float4*>(mp_ + (2 * 4 + nu) * 32 * NC);
                        const float4 m3 = *reinterpret_cast<const float4*>(mp_ + (3 * 4 + nu) * 32 * NC);
                        z[0][nu] = make_float4(m0.x + m1.x + m2.x, m0.y + m1.y + m2.y, m0.z + m1.z + m2.z, m0.w + m1.w + m2.w);
                        z[1][nu] = make_float4(m1.x - m2.x - m3.x, m1.y - m2.y - m3.y, m1.z - m2.z - m3.z, m1.w - m2.w - m3.w);
                    }
                    const float4 bv = bvq;
                    int nl = n0;
                    TMG_PICK_OSEG(p.out, nl, optr, ostride, ooff)
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                        for (int ox = 0; ox < 2; ++ox) {
                            const float4 za = z[oy][0], zb = z[oy][1], zc = z[oy][2], zd = z[oy][3];
                            float4 y = ox == 0 ? make_float4(za.x + zb.x + zc.x, za.y + zb.y + zc.y, za.z + zb.z + zc.z, za.w + zb.w + zc.w)
                                               : make_float4(zb.x - zc.x - zd.x, zb.y - zc.y - zd.y, zb.z - zc.z - zd.z, zb.w - zc.w - zd.w);
                            y.x += bv.x; y.y += bv.y; y.z += bv.z; y.w += bv.w;
                            if (p.relu_out) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
                            const int py = ty_ * TH + 2 * (wt >> 3) + oy, px = tx_ * TW + 2 * (wt & 7) + ox;
                            if (py < p.Hin && px < p.Win)
                                *reinterpret_cast<float4*>(optr + (size_t)(((unsigned)b_ * (unsigned)p.Hin + (unsigned)py) * (unsigned)p.Win + (unsigned)px) * (unsigned)ostride + ooff + nl) = y;
                        }
                }
                cm = 0; tm += G;
            } else {
                ++cm;
            }
        }
        __syncthreads();
    }
#undef TMG_WN_STAGE
#undef TMG_WN_LOADG
}

template <int NTN>
static int launch_wino_nn(const WinoNP& p, int G, hipStream_t st) {
    const size_t lds_bytes = (size_t)(2 * 180 * 40 + 16 * 32 * (NTN * 16 + (NTN < 3 ? 4 : 0))) * sizeof(float);
    TMG_LDS_OPTIN((&wino_nn_kernel<NTN>));
    TmgProf prof(TMG_PROF_WINO, 2.0 * p.B * p.Hin * p.Win * (double)p.Cout * p.Cin * 9, st);
    hipLaunchKernelGGL((wino_nn_kernel<NTN>), dim3(G), dim3(512), lds_bytes, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

// The Winograd contraction for FEW output channels (Cout <= 64) and many input channels: same operand format and descriptors as
// tmg_conv_wino_fwd, up to 3 output segments (out_desc = {stride, off, n} each), dims = {B,H,W,Cin,Cout,relu_in,pad_replicate,relu_out}.
// Returns -100 outside its envelope (float4-addressable operands, channel counts multiples of 4, Cin >= 64, Cout <= 48).
extern "C" int tmg_conv_wino_narrow(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* U, const void* bias,
                                    void* const* out_ptrs, const int64_t* out_desc, int64_t nout, const int64_t* dims, hipStream_t st) {
    WinoNP p;
    p.nseg = (int)nseg;
    if (p.nseg < 1 || p.nseg > TMG_MAX_IN_SEG || nout < 1 || nout > TMG_MAX_OUT_SEG) return -3;
    int csum = 0, osum = 0;
    bool ok = true;
    for (int i = 0; i < TMG_MAX_IN_SEG; ++i) p.in[i] = TmgSeg{nullptr, 0, 0, 0};
    for (int i = 0; i < p.nseg; ++i) {
        p.in[i] = TmgSeg{(const float*)in_ptrs[i], (int)in_desc[3 * i], (int)in_desc[3 * i + 1], (int)in_desc[3 * i + 2]};
        if (((p.in[i].stride | p.in[i].off | p.in[i].n) & 3) || (((uintptr_t)in_ptrs[i]) & 15)) ok = false;
        csum += p.in[i].n;
    }
    for (int i = 0; i < TMG_MAX_OUT_SEG; ++i) p.out[i] = TmgOSeg{nullptr, 0, 0, 0};
    for (int i = 0; i < (int)nout; ++i) {
        p.out[i] = TmgOSeg{(float*)out_ptrs[i], (int)out_desc[3 * i], (int)out_desc[3 * i + 1], (int)out_desc[3 * i + 2]};
        if (((p.out[i].stride | p.out[i].off | p.out[i].n) & 3) || (((uintptr_t)out_ptrs[i]) & 15)) ok = false;
        osum += p.out[i].n;
    }
    p.B = (int)dims[0]; p.Hin = (int)dims[1]; p.Win = (int)dims[2]; p.Cin = (int)dims[3]; p.Cout = (int)dims[4];
    p.relu_in = (int)dims[5]; p.pad_rep = (int)dims[6]; p.relu_out = (int)dims[7];
    if (csum != p.Cin || osum != p.Cout) return -3;
    if ((p.Cout & 3) || (p.Cin & 3) || p.Cout > 48 || p.Cin < 64) ok = false;   // 48: the M buffer of 4 channel tiles does not fit LDS
    if (bias && (((uintptr_t)bias) & 15)) ok = false;
    if (!ok) return -100;
    p.Cin_pad = (p.Cin + 15) & ~15;
    p.Npad = (p.Cout + 15) & ~15;
    p.U = (const float*)U; p.bias = (const float*)bias;
    p.tiles_x = (p.Win + 15) / 16; p.tiles_y = (p.Hin + 7) / 8;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    p.nchunks = (p.Cin_pad + 31) / 32;
    if (p.ntiles <= 0) return 0;
    if ((long long)p.B * p.Hin * p.Win >= (1LL << 31)) return -100;      // 32-bit pixel indices in the kernels
    const int G = p.ntiles < tmg_num_cus() ? p.ntiles : tmg_num_cus();
    switch (p.Npad >> 4) {
        case 1: return launch_wino_nn<1>(p, G, st);
        case 2: return launch_wino_nn<2>(p, G, st);
        default: return launch_wino_nn<3>(p, G, st);
    }
}

// =================================================================================================================================
// Weight gradient as Winograd F(3x3, 2x2): the 3x3 taps of dW are the "outputs", a 2x2 tile of dy is the "filter":
//     dW[ci][co] = A'^T { sum_tiles (B^T x B)[ci] (.) (G' dy G'^T)[co] } A'
//     G' = [[1,0],[.5,.5],[.5,-.5],[0,1]],  A'^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]],  B^T as in the forward kernels
// (the transposition of F(2x2, 3x3): same interpolation points).  16 contractions over the Winograd tiles per (ci, co) pair instead of
// 9 taps x 4 pixels: 2.25x fewer MFMA operations for the largest kernels of the step (the gate-conv and conditioning weight gradients).
//
// A 512-thread block owns CIT x NCO channel tiles (<= 64 x 64 channels) of all 16 positions and a strided share of the 8x16-pixel
// tiles; wave w owns positions (xi = w >> 1, nu = 2 (w & 1) and + 1) and builds its operand fragments straight from the raw
// activations in LDS (one channel per lane): x^ = 6 reads + 5 vector ops per channel tile, dy^ = 4 reads + 6 ops.  Per k-step (4
// Winograd tiles) a wave issues 2 CIT NCO MFMAs.  Raw tiles are register-prefetched one tile ahead (single LDS buffer).  The
// per-block partial sums go to a slab; wino_wgrad_reduce_kernel folds the slabs, applies A'^T . A' and adds the 9 taps onto dW.
// =================================================================================================================================
struct WinoWP {
    TmgSeg in[TMG_MAX_IN_SEG];
    int nseg;
    int B, Hin, Win;
    int Cin, Cout;
    int relu_in, pad_rep;
    const float* dy; int dy_stride, dy_off;
    float* ws;        // [gx][gy][gz][8 waves][2][CIT][NCO][64 lanes][4] partial sums, then [gx][gy][64] bias partials
    int want_bias;
    // grouped launch (the per-layer zero-conv weight gradients of a wide flow level): group g = blockIdx.y / bpg reads its own input
    // segments from gtab[g] ([ngroups][4][4] int64, rows 0-2 = {pointer, pixel stride, channel offset, channels}) and dy channels
    // [g * dy_goff, + Cout) of the shared tensor; null: one group
    const long long* gtab;
    int bpg, dy_goff;
    int tiles_x, tiles_y, ntiles;
};

// DB (round 6): TWO tile buffers where they fit the CU's 160 KB (every instance except 4 x 3 and 4 x 4).  The single-buffered form commits
// the prefetched tile, meets at a barrier, multiplies, meets again: commit + issue + two barriers are 16 % of a launch during which no wave
// feeds the matrix pipe (round-4 ablation).  With two buffers the next tile is committed into the idle buffer BETWEEN k-steps 3 and 4 of
// the current one - LDS writes and the next global loads beside the other waves' MFMAs - and a tile costs one barrier.
template <int CIT, int NCO, bool DB>
__global__ __launch_bounds__(512, 1) void wino_wgrad_kernel(WinoWP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 512;
    constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, PP = PH * PW;
    constexpr int CSX = CIT * 16 + 8, CSD = NCO * 16 + 8;   // pixel strides (words): 2 CS = 16 (mod 32) -> the two tile columns a
                                                            // half-wave reads fall on disjoint bank halves
    constexpr int BUFW = PP * CSX + 128 * CSD;              // words per tile buffer
    float* XR = lds;                // [PP][CSX] raw (activated) input patch
    float* DR = lds + PP * CSX;     // [128][CSD] raw dy tile
    constexpr int KX = CIT <= 1 ? 4 : (CIT <= 2 ? 8 : 16), KXL = CIT <= 1 ? 2 : (CIT <= 2 ? 3 : 4);   // float4 slots per pixel (2^n)
    constexpr int KD = NCO <= 1 ? 4 : (NCO <= 2 ? 8 : 16), KDL = NCO <= 1 ? 2 : (NCO <= 2 ? 3 : 4);
    constexpr int UX = (PP * KX + NT - 1) / NT, UD = (128 * KD + NT - 1) / NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int xi = wave >> 1, par = wave & 1;
    const int grp = p.gtab ? (int)blockIdx.y / p.bpg : 0;
    const int ci0 = (int)blockIdx.z * CIT * 16, co0 = ((int)blockIdx.y - grp * p.bpg) * NCO * 16;
    // row combination of xi for the input transform: u = x[ra] + sg x[rb]; G' row of xi for dy: e = g0 dy[0] + g1 dy[1]
    const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1), rb = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    const float sg = xi == 1 ? 1.f : -1.f;
    const float g0 = xi == 0 ? 1.f : (xi == 3 ? 0.f : 0.5f), g1 = xi == 0 ? 0.f : (xi == 1 ? 0.5f : (xi == 2 ? -0.5f : 1.f));
    // Column pass without selects: with the patch columns taken in the order (c0, c1, c2) for nu = (0, 1) and (c2, c3, c1) for
    // nu = (2, 3), both position pairs are (u0 - u2, u1 * s1 + u2) with s1 = +1 / -1 - a wave-uniform choice of three LDS base
    // addresses and one sign instead of computing both variants and selecting (the compiler if-converts `par ? a : b`).
    const int cc0 = par ? 2 : 0, cc1 = par ? 3 : 1, cc2 = par ? 1 : 2;
    const float s1 = par ? -1.f : 1.f;
    // per-lane word offsets for k-step 0 (tiles 0..3 of tile-row 0: tile q): x rows ra / rb at columns 2 q + cc*; dy pixel (0, 2 q)
    const int oxa0 = (ra * PW + 2 * q + cc0) * CSX + li, oxa1 = (ra * PW + 2 * q + cc1) * CSX + li, oxa2 = (ra * PW + 2 * q + cc2) * CSX + li;
    const int oxb0 = (rb * PW + 2 * q + cc0) * CSX + li, oxb1 = (rb * PW + 2 * q + cc1) * CSX + li, oxb2 = (rb * PW + 2 * q + cc2) * CSX + li;
    const int od = (2 * q) * CSD + li;
    // dy^ of the wave's two positions as ONE linear form of the 2x2 dy tile: (G' dy G'^T)[xi][nu] = sum_ab G'[xi][a] G'[nu][b] dy[a][b]
    typedef float f2 __attribute__((ext_vector_type(2)));
    const float gn0a = par ? 0.5f : 1.f, gn0b = par ? -0.5f : 0.f;   // G' row of the first nu (0 or 2)
    const float gn1a = par ? 0.f : 0.5f, gn1b = par ? 1.f : 0.5f;    // G' row of the second nu (1 or 3)
    const f2 k00 = {g0 * gn0a, g0 * gn1a}, k01 = {g0 * gn0b, g0 * gn1b}, k10 = {g1 * gn0a, g1 * gn1a}, k11 = {g1 * gn0b, g1 * gn1b};

    // ---- staging: a thread owns one channel quad of every (NT / K)-th pixel --------------------------------------------------------
    const int xc4 = tid & (KX - 1), xpix0 = tid >> KXL;
    constexpr int xstep = NT >> KXL;
    const float* xptr = tmg_zero_page;
    int xss = 0;
    {
        int cl = ci0 + 4 * xc4;
        if (xc4 < CIT * 4 && cl < p.Cin) {
            const float* sp0 = p.in[0].p; int ss0 = p.in[0].stride, so0 = p.in[0].off, sn0 = p.in[0].n;
            const float* sp1 = p.in[1].p; int ss1 = p.in[1].stride, so1 = p.in[1].off, sn1 = p.in[1].n;
            const float* sp2 = p.in[2].p; int ss2 = p.in[2].stride, so2 = p.in[2].off;
            if (p.gtab) {
                const long long* gt = p.gtab + (size_t)grp * 16;
                sp0 = reinterpret_cast<const float*>(gt[0]); ss0 = (int)gt[1]; so0 = (int)gt[2]; sn0 = (int)gt[3];
                sp1 = reinterpret_cast<const float*>(gt[4]); ss1 = (int)gt[5]; so1 = (int)gt[6]; sn1 = (int)gt[7];
                sp2 = reinterpret_cast<const float*>(gt[8]); ss2 = (int)gt[9]; so2 = (int)gt[10];
            }
            const float* sp = sp0;
            int ss = ss0, so = so0;
            if (cl >= sn0) {
                cl -= sn0;
                sp = sp1; ss = ss1; so = so1;
                if (cl >= sn1) {
                    cl -= sn1;
                    sp = sp2; ss = ss2; so = so2;
                }
            }
            xptr = sp + so + cl;
            xss = ss;
        }
    }
    unsigned pyx[UX];
#pragma unroll
    for (int u = 0; u < UX; ++u) {
        const int pix = min(xpix0 + u * xstep, PP - 1);
        const int py = pix / PW, px = pix - py * PW;
        pyx[u] = ((unsigned)py << 16) | (unsigned)px;
    }
    const int dc4 = tid & (KD - 1), dpix0 = tid >> KDL;
    constexpr int dstep = NT >> KDL;
    const bool dcv = dc4 < NCO * 4 && co0 + 4 * dc4 < p.Cout;
    const float* dptr = dcv ? p.dy + p.dy_off + grp * p.dy_goff + co0 + 4 * dc4 : tmg_zero_page;
    const int dss = dcv ? p.dy_stride : 0;
    float4 xv[UX], dv[UD];
    float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f);
#define TMG_WW_ISSUE(TILE)                                                                                            \
    {                                                                                                                 \
        int t_ = (TILE);                                                                                              \
        const int tx_ = t_ % p.tiles_x; t_ /= p.tiles_x;                                                              \
        const int ty_ = t_ % p.tiles_y;                                                                               \
        const int b_ = t_ / p.tiles_y;                                                                                \
        const int iy0 = ty_ * TH - 1, ix0 = tx_ * TW - 1;                                                             \
        const unsigned ib = (unsigned)b_ * (unsigned)(p.Hin * p.Win);                                                 \
        _Pragma("unroll") for (int u = 0; u < UX; ++u) {                                                              \
            const int iy = iy0 + (int)(pyx[u] >> 16), ix = ix0 + (int)(pyx[u] & 0xffffu);                             \
            const int iyc = min(max(iy, 0), p.Hin - 1), ixc = min(max(ix, 0), p.Win - 1);                             \
            const bool oob = !p.pad_rep && (iy != iyc || ix != ixc);                                                  \
            const float* a_ = (oob || xpix0 + u * xstep >= PP) ? tmg_zero_page : xptr + (size_t)(ib + (unsigned)iyc * (unsigned)p.Win + (unsigned)ixc) * (unsigned)xss; \
            xv[u] = tmg_ldg4(a_);      /* (the pointer may come from the grouped launch's table: see tmg_ldg4) */        \
        }                                                                                                             \
        _Pragma("unroll") for (int u = 0; u < UD; ++u) {                                                              \
            const int m = dpix0 + u * dstep;                                                                          \
            const int oy = ty_ * TH + (m >> 4), ox = tx_ * TW + (m & 15);                                             \
            const bool inb = m < 128 && oy < p.Hin && ox < p.Win;                                                     \
            const float* a_ = inb ? dptr + (size_t)(ib + (unsigned)oy * (unsigned)p.Win + (unsigned)ox) * (unsigned)dss : tmg_zero_page; \
            dv[u] = *reinterpret_cast<const float4*>(a_);                                                             \
        }                                                                                                             \
    }

    f32x4 acc[2][CIT][NCO];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < CIT; ++i)
#pragma unroll
            for (int n = 0; n < NCO; ++n) acc[e][i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int G = gridDim.x;
#define TMG_WW_COMMIT(XRB, DRB)                                                                                        \
    {                                                                                                                  \
        _Pragma("unroll") for (int u = 0; u < UX; ++u) {                                                               \
            if (xc4 < CIT * 4 && xpix0 + u * xstep < PP) {                                                             \
                float4 v = xv[u];                                                                                      \
                if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); } \
                *reinterpret_cast<float4*>((XRB) + (xpix0 + u * xstep) * CSX + 4 * xc4) = v;                           \
            }                                                                                                          \
        }                                                                                                              \
        _Pragma("unroll") for (int u = 0; u < UD; ++u) {                                                               \
            if (dc4 < NCO * 4 && dpix0 + u * dstep < 128) {                                                            \
                *reinterpret_cast<float4*>((DRB) + (dpix0 + u * dstep) * CSD + 4 * dc4) = dv[u];                       \
                bacc.x += dv[u].x; bacc.y += dv[u].y; bacc.z += dv[u].z; bacc.w += dv[u].w;                            \
            }                                                                                                          \
        }                                                                                                              \
    }
    // 8 k-steps of 4 Winograd tiles: tile index 4 s + q -> tile row s >> 1, tile column 4 (s & 1) + q
#define TMG_WW_KSTEPS(S0, S1, XRB, DRB)                                                                                \
    _Pragma("unroll") for (int s = (S0); s < (S1); ++s) {                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        const int sx = ((2 * (s >> 1)) * PW + 8 * (s & 1)) * CSX;          /* patch offset of the k-step's first tile */ \
        const int sd = ((2 * (s >> 1)) * 16 + 8 * (s & 1)) * CSD;                                                      \
        float xa[2][CIT], dh[2][NCO];                                                                                  \
        _Pragma("unroll") for (int i = 0; i < CIT; ++i) {                                                              \
            const float u0 = fmaf((XRB)[oxb0 + sx + 16 * i], sg, (XRB)[oxa0 + sx + 16 * i]);                           \
            const float u1 = fmaf((XRB)[oxb1 + sx + 16 * i], sg, (XRB)[oxa1 + sx + 16 * i]);                           \
            const float u2 = fmaf((XRB)[oxb2 + sx + 16 * i], sg, (XRB)[oxa2 + sx + 16 * i]);                           \
            xa[0][i] = u0 - u2;                                                                                        \
            xa[1][i] = fmaf(u1, s1, u2);                                                                               \
        }                                                                                                              \
        _Pragma("unroll") for (int n = 0; n < NCO; ++n) {                                                              \
            const float* dp_ = (DRB) + od + sd + 16 * n;                                                               \
            const f2 d_ = k00 * dp_[0] + k01 * dp_[CSD] + k10 * dp_[16 * CSD] + k11 * dp_[17 * CSD];                   \
            dh[0][n] = d_.x; dh[1][n] = d_.y;                                                                          \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int i = 0; i < CIT; ++i)                                                                \
            _Pragma("unroll") for (int n = 0; n < NCO; ++n) {                                                          \
                acc[0][i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0][i], dh[0][n], acc[0][i][n], 0, 0, 0);         \
                acc[1][i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[1][i], dh[1][n], acc[1][i][n], 0, 0, 0);         \
            }                                                                                                          \
    }
    if constexpr (DB) {
        int cur = 0;
        if ((int)blockIdx.x < p.ntiles) {
            TMG_WW_ISSUE((int)blockIdx.x)
            TMG_WW_COMMIT(XR, DR)
            if ((int)blockIdx.x + G < p.ntiles) TMG_WW_ISSUE((int)blockIdx.x + G)
        }
        __syncthreads();
        for (int tile = blockIdx.x; tile < p.ntiles; tile += G) {
            const float* xrb = XR + cur * BUFW;
            const float* drb = DR + cur * BUFW;
            TMG_WW_KSTEPS(0, 4, xrb, drb)
            __builtin_amdgcn_sched_barrier(0);
            if (tile + G < p.ntiles) {
                // the next tile (in flight since the previous round) -> the idle buffer; the one after it -> the registers
                TMG_WW_COMMIT(XR + (cur ^ 1) * BUFW, DR + (cur ^ 1) * BUFW)
                if (tile + 2 * G < p.ntiles) TMG_WW_ISSUE(tile + 2 * G)
            }
            __builtin_amdgcn_sched_barrier(0);
            TMG_WW_KSTEPS(4, 8, xrb, drb)
            __syncthreads();      // the idle buffer is complete, this one has been read by every wave
            cur ^= 1;
        }
    } else {
        if ((int)blockIdx.x < p.ntiles) TMG_WW_ISSUE((int)blockIdx.x)
        for (int tile = blockIdx.x; tile < p.ntiles; tile += G) {
            // ---- commit the prefetched tile, then put the next one in flight -----------------------------------------------------
            TMG_WW_COMMIT(XR, DR)
            __syncthreads();
            if (tile + G < p.ntiles) TMG_WW_ISSUE(tile + G)
            TMG_WW_KSTEPS(0, 8, XR, DR)
            __syncthreads();
        }
    }
#undef TMG_WW_COMMIT
#undef TMG_WW_KSTEPS
#undef TMG_WW_ISSUE
    // ---- partial sums -> this block's slab (accumulator order, coalesced float4 stores) ---------------------------------------------
    const size_t bl = ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z;
    float4* slab = reinterpret_cast<float4*>(p.ws) + ((bl * 8 + wave) * (2 * CIT * NCO)) * 64 + lane;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < CIT; ++i)
#pragma unroll
            for (int n = 0; n < NCO; ++n)
                slab[((e * CIT + i) * NCO + n) * 64] = make_float4(acc[e][i][n][0], acc[e][i][n][1], acc[e][i][n][2], acc[e][i][n][3]);
    if (p.want_bias && blockIdx.z == 0) {
        // dbias partial of this block: thread t holds the sum of channel quad t % KD over its pixels
        __syncthreads();
        *reinterpret_cast<float4*>(lds + tid * 4) = bacc;
        __syncthreads();
        if (tid < NCO * 16) {
            const int c4 = tid >> 2, e = tid & 3;
            float bs = 0.f;
            for (int t = c4; t < NT; t += KD) bs += lds[t * 4 + e];
            float* wsb = p.ws + (size_t)gridDim.x * gridDim.y * gridDim.z * 8 * (2 * CIT * NCO) * 256;
            wsb[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 64 + tid] = bs;
        }
    }
}

// Fold the slabs of wino_wgrad_kernel over the pixel shares, apply A'^T . A' and add the 9 taps onto dW (+ the bias sums onto dbias).
// One block of NG x 64 threads per pair of channel tiles: thread (lane, group g) sums the slabs bx = g, g + NG, ... for its 4 input
// channels x 1 output channel at all 16 positions, transforms its partial sum (the tap transform is linear), the NG groups meet in LDS
// and group 0 adds the result onto dW in a fixed order.  (A single serial walk over all gx slabs per thread left the chip idle for 120 us
// per launch; with only gy gz CIT NCO = 24..64 blocks per launch the walk of gx / NG slabs per thread is a chain of exposed load
// latencies: NG = 8 for the launches with many slabs.)
template <int NG>
// (a vector-ALU kernel in a file built without the packed-fp32 instructions: it keeps them - without, the 16-group instance spills)
__global__ __launch_bounds__(64 * NG) TMG_PACKED_F32 void wino_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, float* __restrict__ dbias,
                                                                   int gx, int gy, int gz, int CIT, int NCO, int Cin, int Cout, int cin_dst,
                                                                   int cin_valid, int ci_split, int ci_off0, int ci_off1, int bpg,
                                                                   long long dw_gstride, int db_gstride) {
    extern __shared__ __attribute__((aligned(16))) float red_[];
    float (*red)[64][37] = reinterpret_cast<float (*)[64][37]>(red_);   // [NG - 1][64][37]
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    int r_ = blockIdx.x;
    const int n = r_ % NCO; r_ /= NCO;
    const int i = r_ % CIT; r_ /= CIT;
    const int bz = r_ % gz, by = r_ / gz;
    const int li = lane & 15, q = lane >> 4;
    float4 m[16];
#pragma unroll
    for (int pz = 0; pz < 16; ++pz) m[pz] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int bx = grp; bx < gx; bx += NG) {
        const size_t bl = ((size_t)bx * gy + by) * gz + bz;
#pragma unroll
        for (int pz = 0; pz < 16; ++pz) {
            const int w = pz >> 1, e = pz & 1;   // position pz = 4 xi + nu lives in wave (xi, nu >> 1) = pz >> 1, slot nu & 1
            const float4 v = reinterpret_cast<const float4*>(ws)[((bl * 8 + w) * (2 * CIT * NCO) + (e * CIT + i) * NCO + n) * 64 + lane];
            m[pz].x += v.x; m[pz].y += v.y; m[pz].z += v.z; m[pz].w += v.w;
        }
    }
    float tap[4][9];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float M[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float4 v = m[a * 4 + b];
                M[a][b] = r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w));
            }
        // A'^T M: rows  t0 = M0 + M1 + M2, t1 = M1 - M2, t2 = M1 + M2 - M3 ; then the same along the columns
        float T[3][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            T[0][b] = M[0][b] + M[1][b] + M[2][b];
            T[1][b] = M[1][b] - M[2][b];
            T[2][b] = M[1][b] + M[2][b] - M[3][b];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            tap[r][a * 3 + 0] = T[a][0] + T[a][1] + T[a][2];
            tap[r][a * 3 + 1] = T[a][1] - T[a][2];
            tap[r][a * 3 + 2] = T[a][1] + T[a][2] - T[a][3];
        }
    }
    if (grp > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 9; ++k) red[grp - 1][lane][r * 9 + k] = tap[r][k];
    }
    __syncthreads();
    const int group = by / bpg, cb = by - group * bpg;   // grouped launch: block row by = (group, output-channel block)
    dW += (size_t)group * dw_gstride;
    if (grp == 0) {
        const int co = (cb * NCO + n) * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = (bz * CIT + i) * 16 + 4 * q + r;
            if (co >= Cout || ci >= Cin || ci >= cin_valid) continue;
            float* dst = dW + ((size_t)co * cin_dst + ci + (ci < ci_split ? ci_off0 : ci_off1)) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float v = tap[r][k];
#pragma unroll
                for (int g_ = 0; g_ < NG - 1; ++g_) v += red[g_][lane][r * 9 + k];
                dst[k] += v;
            }
        }
    }
    if (dbias && bz == 0 && i == 0 && n == 0 && threadIdx.x < NCO * 16) {
        dbias += (size_t)group * db_gstride;
        const int c = threadIdx.x, co = cb * NCO * 16 + c;
        if (co < Cout) {
            const float* wsb = ws + (size_t)gx * gy * gz * 8 * (2 * CIT * NCO) * 256;
            float s_ = 0.f;
            for (int bx = 0; bx < gx; ++bx) s_ += wsb[((size_t)bx * gy + by) * 64 + c];
            dbias[co] += s_;
        }
    }
}

struct WinoWPlan { int CIT, NCO, gx, gy, gz; size_t ws_floats; };

static int plan_wino_wgrad(int B, int H, int W, int Cin, int Cout, int ngroups, WinoWPlan* pl) {
    const int cit = (Cin + 15) >> 4, cot = (Cout + 15) >> 4;
    if (cit < 2 || cot < 2) return -100;
    pl->gz = (cit + 3) / 4;
    const int citg = (cit + pl->gz - 1) / pl->gz;
    pl->CIT = citg <= 2 ? 2 : (citg == 3 ? 3 : 4);
    pl->gy = (cot + 3) / 4;
    // (4 x 3 register tiles - double-buffered - in place of 4 x 4 for the 64-channel blocks: 3.39 against 2.77 ms at 104 -> 256, round 6)
    const int cotg = (cot + pl->gy - 1) / pl->gy;
    pl->NCO = cotg <= 2 ? 2 : (cotg == 3 ? 3 : 4);
    pl->gy *= ngroups;     // block row = (group, output-channel block)
    const int ntiles = B * ((W + 15) / 16) * ((H + 7) / 8);
    int gx = tmg_num_cus() / (pl->gy * pl->gz);
    if (gx > ntiles / 2) gx = ntiles / 2;
    if (gx < 1) gx = 1;
    pl->gx = gx;
    pl->ws_floats = (size_t)gx * pl->gy * pl->gz * 8 * (2 * pl->CIT * pl->NCO) * 256 + (size_t)gx * pl->gy * 64;
    return 0;
}

// scratch floats tmg_conv_wino_wgrad wants for dims = {B,H,W,Cin,Cout,...} (0: shape not eligible)
extern "C" int64_t tmg_conv_wino_wgrad_ws_floats(const int64_t* dims) {
    WinoWPlan pl;
    if (plan_wino_wgrad((int)dims[0], (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4], 1, &pl) != 0) return 0;
    return (int64_t)pl.ws_floats;
}

template <int CIT, int NCO>
static int launch_wino_wgrad(const WinoWP& p, const WinoWPlan& pl, hipStream_t st) {
    constexpr size_t one = (size_t)(180 * (CIT * 16 + 8) + 128 * (NCO * 16 + 8)) * sizeof(float);
    constexpr bool DB = 2 * one <= 160 * 1024;       // two tile buffers where they fit (see the kernel)
    static const int nodb = getenv("TMG_WW_NO_DB") ? 1 : 0;      // A / B switch
    const double ngr = p.gtab ? (double)(pl.gy / (p.bpg > 0 ? p.bpg : 1)) : 1.0;      // (a grouped launch: every group's flops)
    TmgProf prof(TMG_PROF_WINO_WG, ngr * 2.0 * p.B * p.Hin * p.Win * (double)p.Cout * p.Cin * 9, st);
    if (DB && !nodb) {
        TMG_LDS_OPTIN((&wino_wgrad_kernel<CIT, NCO, DB>));
        hipLaunchKernelGGL((wino_wgrad_kernel<CIT, NCO, DB>), dim3(pl.gx, pl.gy, pl.gz), dim3(512), 2 * one, st, p);
    } else {
        TMG_LDS_OPTIN((&wino_wgrad_kernel<CIT, NCO, false>));
        hipLaunchKernelGGL((wino_wgrad_kernel<CIT, NCO, false>), dim3(pl.gx, pl.gy, pl.gz), dim3(512), one < 8192 ? 8192 : one, st, p);
    }
    TMG_CHECK_LAUNCH();
    return 0;
}

static int wino_wgrad_impl(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* dy, const int64_t* dy_desc,
                           void* dW, void* dbias, void* ws, int64_t ws_floats, const int64_t* dims, hipStream_t st, const long long* gtab,
                           int ngroups, int dy_goff, long long dw_gstride, int db_gstride) {
    WinoWP p;
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
    const int cin_dst = dims[7] > 0 ? (int)dims[7] : p.Cin;
    const int cin_valid = dims[8] > 0 ? (int)dims[8] : (cin_dst < p.Cin ? cin_dst : p.Cin);
    const int ci_split = dims[9] > 0 ? (int)dims[9] : 0x7fffffff;
    const int ci_off0 = (int)dims[10], ci_off1 = (int)dims[11];
    if (csum != p.Cin) return -3;
    p.dy = (const float*)dy; p.dy_stride = (int)dy_desc[0]; p.dy_off = (int)dy_desc[1];
    if (((p.dy_stride | p.dy_off | dy_goff) & 3) || (((uintptr_t)dy) & 15) || (p.Cin & 3) || (p.Cout & 3)) ok = false;
    WinoWPlan pl;
    if (!ok || plan_wino_wgrad(p.B, p.Hin, p.Win, p.Cin, p.Cout, ngroups, &pl) != 0) return -100;
    if (!ws || (size_t)ws_floats < pl.ws_floats || (((uintptr_t)ws) & 15)) return -100;
    p.ws = (float*)ws;
    p.want_bias = dbias ? 1 : 0;
    p.gtab = gtab; p.bpg = pl.gy / ngroups; p.dy_goff = dy_goff;
    p.tiles_x = (p.Win + 15) / 16; p.tiles_y = (p.Hin + 7) / 8;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    if (p.ntiles <= 0) return 0;
    if ((long long)p.B * p.Hin * p.Win >= (1LL << 31)) return -100;      // 32-bit pixel indices in the kernels
    int rc = -7;
#define TMG_WW_CASE(C_, N_) if (pl.CIT == C_ && pl.NCO == N_) rc = launch_wino_wgrad<C_, N_>(p, pl, st);
    TMG_WW_CASE(2, 2) TMG_WW_CASE(2, 3) TMG_WW_CASE(2, 4) TMG_WW_CASE(3, 2) TMG_WW_CASE(3, 3) TMG_WW_CASE(3, 4) TMG_WW_CASE(4, 2) TMG_WW_CASE(4, 3)
    TMG_WW_CASE(4, 4)
#undef TMG_WW_CASE
    if (rc != 0) return rc;
    const int ng = pl.gx >= 64 ? 16 : (pl.gx >= 32 ? 8 : 4);
    if (ng == 16) {
        TMG_LDS_OPTIN((&wino_wgrad_reduce_kernel<16>));
        hipLaunchKernelGGL(wino_wgrad_reduce_kernel<16>, dim3(pl.gy * pl.gz * pl.CIT * pl.NCO), dim3(1024), 15 * 64 * 37 * sizeof(float), st,
                           (const float*)p.ws, (float*)dW, (float*)dbias, pl.gx, pl.gy, pl.gz, pl.CIT, pl.NCO, p.Cin, p.Cout, cin_dst, cin_valid,
                           ci_split, ci_off0, ci_off1, p.bpg, dw_gstride, db_gstride);
    } else if (ng == 8) {
        TMG_LDS_OPTIN((&wino_wgrad_reduce_kernel<8>));
        hipLaunchKernelGGL(wino_wgrad_reduce_kernel<8>, dim3(pl.gy * pl.gz * pl.CIT * pl.NCO), dim3(512), 7 * 64 * 37 * sizeof(float), st,
                           (const float*)p.ws, (float*)dW, (float*)dbias, pl.gx, pl.gy, pl.gz, pl.CIT, pl.NCO, p.Cin, p.Cout, cin_dst, cin_valid,
                           ci_split, ci_off0, ci_off1, p.bpg, dw_gstride, db_gstride);
    } else {
        hipLaunchKernelGGL(wino_wgrad_reduce_kernel<4>, dim3(pl.gy * pl.gz * pl.CIT * pl.NCO), dim3(256), 3 * 64 * 37 * sizeof(float), st,
                           (const float*)p.ws, (float*)dW, (float*)dbias, pl.gx, pl.gy, pl.gz, pl.CIT, pl.NCO, p.Cin, p.Cout, cin_dst, cin_valid,
                           ci_split, ci_off0, ci_off1, p.bpg, dw_gstride, db_gstride);
    }
    TMG_CHECK_LAUNCH();
    return 0;
}

// dW[Cout][cin_dst][3][3] += sum_pixels act(in)(p + tap) (x) dy(p)  and  dbias += sum dy, for 3x3 / stride-1 / padding-1 convs with
// >= 32 input and output channels, as Winograd F(3x3, 2x2).  dims = {B,H,W,Cin,Cout,relu_in,pad_replicate,cin_dst,cin_valid,ci_split,
// ci_off0,ci_off1} (destination mapping as tmg_conv_wgrad); dy_desc = {stride, off}; ws: >= tmg_conv_wino_wgrad_ws_floats(dims) floats.
// Returns -100 outside the envelope (nothing launched).
extern "C" int tmg_conv_wino_wgrad(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* dy, const int64_t* dy_desc,
                                   void* dW, void* dbias, void* ws, int64_t ws_floats, const int64_t* dims, hipStream_t st) {
    return wino_wgrad_impl(in_ptrs, in_desc, nseg, dy, dy_desc, dW, dbias, ws, ws_floats, dims, st, nullptr, 1, 0, 0, 0);
}

// `ngroups` identically shaped weight gradients in one launch (the per-layer zero-conv weight gradients of a wide flow level): arguments
// as tmg_conv_wgrad_grouped without per-group dy tensors - gtab = device table [ngroups][4][4] int64 (rows 0-2: the group's input
// segments), gdims = {dy channels per group, dW floats per group, dbias floats per group}.  ws: tmg_conv_wino_wgrad_grouped_ws_floats.
extern "C" int tmg_conv_wino_wgrad_grouped(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* gtab, int64_t ngroups,
                                           const int64_t* gdims, const void* dy, const int64_t* dy_desc, void* dW, void* dbias, void* ws,
                                           int64_t ws_floats, const int64_t* dims, hipStream_t st) {
    if (ngroups < 1 || !gtab) return -3;
    return wino_wgrad_impl(in_ptrs, in_desc, nseg, dy, dy_desc, dW, dbias, ws, ws_floats, dims, st, (const long long*)gtab, (int)ngroups,
                           (int)gdims[0], (long long)gdims[1], (int)gdims[2]);
}

extern "C" int64_t tmg_conv_wino_wgrad_grouped_ws_floats(const int64_t* dims, int64_t ngroups) {
    WinoWPlan pl;
    if (plan_wino_wgrad((int)dims[0], (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4], (int)ngroups, &pl) != 0) return 0;
    return (int64_t)pl.ws_floats;
}
