// tmg_coupling.hip -- the traffic-heavy part of an affine coupling layer in ONE launch, for the narrow flow levels (C <= 32
// channels) where the per-op chain (zero conv -> coupling -> channel mix) is bound by HBM round trips of hh and y and by launch
// ramps, not by arithmetic.
//
// Reference semantics (paths relative to /root/reference/tmglow/nn/modules/), t2 = cat(x1, cond, d1, d2):
//   hh = (conv3x3_valid(pad_replicate(relu(t2))) + b) * exp(clamp(kappa, -4, ln 4))                  flowUtils.py:246-247
//   shift = hh[0::2], r = hh[1::2], sg = 2 softsign(r)                                               flowAffine.py:76-83, :102-109
//   forward: y2 = (x2 + shift) exp(sg)      reverse: y2 = x2 exp(-sg) - shift      logdet[b] += sum sg
//   then (reverse direction) the folded ActNorm + invertible 1x1 mix: out = Wm [x1; y2] + bm       glowConv.py:207-222, actNorm.py:71-85
// The growth layers d1, d2 (denseBlock.py:135-152) come from tmg_c1x2_fwd (vector-ALU work: fusing them in here as well made the
// kernel VALU-bound - 2 100 vector instructions per thread and tile, 150 us per level-1 layer against 155 us for the unfused
// chain); the conditioning map's share of the zero conv arrives pre-computed per level (hc: a conv is linear in its input
// channels and every layer of a level sees the same map).
//
// One 256-thread block walks 16x16-pixel tiles, the loads of tile k+1 in flight while tile k computes:
//   stage   relu(x1 | d1, d2) on the tile + 1 halo pixel into LDS, replicate padding by clamping the coordinates; raw x1 of the tile
//   zero conv on the matrix cores (v_mfma_f32_16x16x4_f32): A = weights (16 output channels x 4 input channels, registers),
//           B = activations (4 channels x 16 consecutive pixels of a tile row, one ds_read_b32 per fragment at base + immediate,
//           conflict-free because LDS holds channel PAIRS per plane: [c/2][pixel][2]); a lane ends up with 4 consecutive hh
//           channels of a pixel = two (shift, r) pairs, so the affine coupling runs in registers;
//   mix     [x1; y2] is already laid out as the B operand of a second MFMA contraction (k-order chosen to match the
//           accumulator layout, the weight fragments are permuted instead); out = Wm y + bm leaves as one float4 per lane.
// HBM traffic per pixel: reads x (C + halo share of x1), D (4), hc (C); writes out (C), r (C/2), y2 (C/2).
#include "tmg_common.h"
#include <stdlib.h>

// exp and 1/x on the hardware transcendental unit (v_exp_f32 / v_rcp_f32, ~1 ulp): the IEEE-exact library forms cost ~10 vector
// instructions each, and these kernels are bound by vector-instruction issue (1 900 per wave and tile against 180 MFMAs in the
// backward kernel).  The results stay far inside the fp32 parity tolerances (exp feeds a factor in [e^-2, e^2]).
// Element offset of pixel PX in a tensor with pixel stride STRIDE: the pixel index is a 32-bit unsigned (the launchers refuse
// B H W >= 2^31), so the product is ONE v_mad_u64_u32 instead of the multi-instruction 64 x 32-bit multiply a size_t index costs -
// these kernels are bound by vector-instruction issue, and their integer address arithmetic outnumbered the float math 3 : 1
#define TMG_PXO(PX, STRIDE) ((size_t)(PX) * (unsigned)(STRIDE))
__device__ __forceinline__ float cpl_exp(float v) { return __expf(v); }
__device__ __forceinline__ float cpl_rcp(float v) { return __frcp_rn(v); }

struct CplFP {
    // The two channel halves of an activation are addressed separately: half 1 = channels [0, ch) at x + pixel * xs, half 2 =
    // channels [ch, C) at x2 + pixel * x2s.  One interleaved [npix][C] tensor is x2 = x + ch, x2s = xs; the narrow levels keep the
    // halves in tensors of their own (round 4: the kernels that read x1 alone - growth layers, their backward, three weight
    // gradients - then fetch whole cache lines of what they use, and this kernel's x1 patch and x2 epilogue loads stop evicting
    // each other's half-used lines).
    const float* x; int xs;          // layer input, half 1 (x1)
    const float* x2; int x2s;        // layer input, half 2 (x2)
    float* out; int os;              // layer output, half 1
    float* out2; int o2s;            // layer output, half 2
    float* rsave;                    // [npix][ch] softsign arguments r
    float* y2save;                   // [npix][ch] transformed half (null: not stored)
    const float* D;                  // [npix][4] raw (d1, d2, 0, 0)
    const float* hc; int hcs;        // cond part of the zero conv, C channels, before bias / scale
    const float* wz; int wz_rows;    // torch layout [C][wz_rows][3][3]; x1 channel c reads column c
    int wz_d1col;                    // column of d1 (d2 = the next one)
    const float* bz; const float* kappa;
    const float* Wm; const float* bm;   // trailing channel mix [C][C], [C]; null: out = [x1 | y2]
    float* logdet;                   // [B], accumulated
    int B, H, W, C, reverse;
    int tiles_x, tiles_y, ntiles;
    int xmap;                        // XCD-aware tile order (tmg_common.h)
};

// CT: 16-channel output tiles (C <= 16 CT); K4: input-channel quads of the zero conv = ch/4 + 1 (the last quad is d1, d2, 0, 0)
template <int CT, int K4>
__global__ __launch_bounds__(256, CT == 1 ? 3 : 2) void cpl_fwd_kernel(CplFP p) {
    constexpr int CH4 = K4 - 1, CHP = 2 * CH4;        // x1 channel quads / pairs
    constexpr int PW = 18, PP = PW * PW;              // staged patch (tile + halo 1)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // two tile buffers (a tile is committed while the other waves may still read the previous one: ONE barrier per tile), each
    //   XQ [CHP + 1][PP][2]  relu(x1) pairs, then (relu d1, relu d2)       XR [CHP][256][2]  raw x1 of the tile
    constexpr int BUFW = (CHP + 1) * PP * 2 + CHP * 256 * 2;
    constexpr int ZPN = (5 * PW + 4) * 2;
    float* ZP = lds + 2 * BUFW;                       // [ZPN] zeros: channels ch+2, ch+3 of the zero conv input (as large as the largest
                                                      //       fragment offset: those lanes need no special case)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int C = p.C, ch = C >> 1;

    // ---- per-block constants: weight fragments ----------------------------------------------------------------------
    if (tid < ZPN) ZP[tid] = 0.f;
    // zero-conv A fragments: A[i = li][k = q] of k-step (tap, s) = Wz[16 mt + li][column of channel 4 s + q][tap]
    float wa[9][K4][CT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < K4; ++s)
#pragma unroll
            for (int mt = 0; mt < CT; ++mt) {
                const int co = 16 * mt + li, c = 4 * s + q;
                int col = -1;
                if (s < CH4) col = (c < ch) ? c : -1;
                else if (q < 2) col = p.wz_d1col + q;
                // (address select, not a guarded load: a load inside a divergent block is followed by s_waitcnt vmcnt(0), which made
                // these 9 K4 CT loads a chain of L2 latencies - 25 us per block)
                wa[t][s][mt] = *((co < C && col >= 0) ? p.wz + ((size_t)co * p.wz_rows + col) * 9 + t : tmg_zero_page);
            }
    // mix A fragments: k-steps 0 .. CH4-1 carry x1 channels 4 t + q; k-steps CH4 + 2 mt + e carry y2 channel 8 mt + 2 q + e
    // (that is how the coupling leaves y2 in the accumulator registers)
    constexpr int KM = CH4 + 2 * CT;
    float wm[CT][KM];
    float4 bmv[CT], bzv[CT];
    const float osc = out_scale_of(p.kappa);
#pragma unroll
    for (int mo = 0; mo < CT; ++mo) {
        const int co = 16 * mo + li;
#pragma unroll
        for (int t = 0; t < KM; ++t) {
            int k;
            if (t < CH4) k = (4 * t + q < ch) ? 4 * t + q : -1;
            else {
                const int mt = (t - CH4) >> 1, e = (t - CH4) & 1, j = 8 * mt + 2 * q + e;
                k = (j < ch) ? ch + j : -1;
            }
            wm[mo][t] = *((p.Wm && co < C && k >= 0) ? p.Wm + (size_t)co * C + k : tmg_zero_page);
        }
        const int c4 = 16 * mo + 4 * q;
        bmv[mo] = *reinterpret_cast<const float4*>((p.bm && c4 < C) ? p.bm + c4 : tmg_zero_page);
        bzv[mo] = *reinterpret_cast<const float4*>((c4 < C) ? p.bz + c4 : tmg_zero_page);
    }
    __syncthreads();

    // Tile pipeline: the global loads of tile k+1 are issued into registers right after tile k's staging barrier
    constexpr int NIT = PP * K4;                      // staged float4 items of a tile: (patch pixel, channel quad)
    constexpr int NPI = (NIT + 255) / 256;
    float4 pv[NPI];
#define TMG_CPL_ORIGIN(TILE)                                                  \
    int b_, oy0_, ox0_;                                                       \
    {                                                                         \
        int t_ = (TILE);                                                      \
        const int tx_ = t_ % p.tiles_x; t_ /= p.tiles_x;                      \
        const int ty_ = t_ % p.tiles_y;                                       \
        b_ = t_ / p.tiles_y; oy0_ = ty_ * 16; ox0_ = tx_ * 16;                \
    }
#define TMG_CPL_ISSUE(TILE)                                                                                         \
    {                                                                                                               \
        TMG_CPL_ORIGIN(TILE)                                                                                        \
        const unsigned img_ = (unsigned)b_ * (unsigned)(p.H * p.W);                                                 \
        _Pragma("unroll") for (int u = 0; u < NPI; ++u) {                                                           \
            const int i = min(tidl + u * 256, NIT - 1);                                                             \
            const int pp = i / K4, s_ = i - pp * K4;                                                                \
            const int py = pp / PW, px = pp - py * PW;                                                              \
            /* replicate padding: clamp the coordinates (flowUtils.py:246) */                                       \
            const int gy = min(max(oy0_ - 1 + py, 0), p.H - 1), gx = min(max(ox0_ - 1 + px, 0), p.W - 1);           \
            const unsigned gp_ = img_ + (unsigned)gy * (unsigned)p.W + (unsigned)gx;                                \
            const float* a_ = s_ < CH4 ? (4 * s_ < ch ? p.x + TMG_PXO(gp_, p.xs) + 4 * s_ : tmg_zero_page) : p.D + TMG_PXO(gp_, 4); \
            pv[u] = *reinterpret_cast<const float4*>(a_);                                                           \
        }                                                                                                           \
    }
    // a block owns a contiguous range of tiles (neighbouring tiles share halo lines in L2, and the log-det partial sums of an
    // image stay in registers until the range moves on to the next image: one atomic per wave and image, not per tile)
    const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t0 = tmg_xcd_block((int)blockIdx.x, (int)gridDim.x, p.xmap) * per, t1 = min(t0 + per, p.ntiles);
    // The staging items' patch coordinates (i / K4, / PW) depend on the thread alone: left to the compiler they are hoisted out of the
    // tile loop - NPI items x (LDS offsets, patch row / column) - and, at C = 32 where the weight fragments already hold 114 registers,
    // spilled (152 bytes per lane, reloaded in every tile's commit).  An opaque copy of the thread index per tile keeps them where
    // they are used: ~10 integer instructions per item and tile.
    int tidl = tid;
    if (t0 < t1) TMG_CPL_ISSUE(t0)
    int par = 0, ldb = -1;
    float ldacc = 0.f;
    for (int tile = t0; tile < t1; ++tile, par ^= 1) {
        // (C <= 16: no spill to cure, and the recomputed arithmetic measured +3.5 % on the first level's launch - not applied there)
        if (CT == 2) asm volatile("" : "+v"(tidl));
        TMG_CPL_ORIGIN(tile)
        const int b = b_, oy0 = oy0_, ox0 = ox0_;
        const unsigned img = (unsigned)b * (unsigned)(p.H * p.W);
        float* XQ = lds + par * BUFW;
        float* XR = XQ + (CHP + 1) * PP * 2;
        // ---- commit the staged registers: relu(x1 | d1, d2) on the patch, raw x1 of the tile -------------------------------
#pragma unroll
        for (int u = 0; u < NPI; ++u) {
            const int i = tidl + u * 256;
            if (i < NIT) {
                const int pp = i / K4, s = i - pp * K4;
                const int py = pp / PW, px = pp - py * PW;
                const float4 v = pv[u];
                *reinterpret_cast<float2*>(XQ + ((2 * s) * PP + pp) * 2) = make_float2(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f));
                if (s < CH4) {
                    *reinterpret_cast<float2*>(XQ + ((2 * s + 1) * PP + pp) * 2) = make_float2(fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                    const int cy = py - 1, cx = px - 1;
                    if (cy >= 0 && cy < 16 && cx >= 0 && cx < 16) {
                        const int cp = cy * 16 + cx;
                        *reinterpret_cast<float2*>(XR + ((2 * s) * 256 + cp) * 2) = make_float2(v.x, v.y);
                        *reinterpret_cast<float2*>(XR + ((2 * s + 1) * 256 + cp) * 2) = make_float2(v.z, v.w);
                    }
                }
            }
        }
        __syncthreads();
        if (tile + 1 < t1) TMG_CPL_ISSUE(tile + 1)
        // ---- this wave owns tile rows 4 wave .. 4 wave + 3, one 16-pixel n-tile each, walked by a ROLLED loop: one row's
        //      accumulators and epilogue operands live at a time (registers buy occupancy here: the kernel waits on memory, and
        //      with 4 rows unrolled it needed 200 VGPRs = 2 waves per SIMD); the next row's epilogue operands are in flight
        //      while this row runs through the matrix cores
        const int row0 = 4 * wave;
        const float* xb = XQ + (((q >> 1) * PP + row0 * PW + li) * 2 + (q & 1));
        const float* db = (q < 2) ? XQ + ((CHP * PP + row0 * PW + li) * 2 + q) : ZP;
        const float* xrb = XR + (((q >> 1) * 256 + row0 * 16 + li) * 2 + (q & 1));
        const int gx = ox0 + li, gxc = min(gx, p.W - 1);
        float4 hcur[CT];
        float2 xcur[CT];
#define TMG_CPL_EPI_LOAD(HV, XV, NT)                                                                                 \
        {                                                                                                            \
            const int gy_ = min(oy0 + row0 + (NT), p.H - 1);                                                         \
            const unsigned gp_ = img + (unsigned)gy_ * (unsigned)p.W + (unsigned)gxc;                                \
            _Pragma("unroll") for (int mt = 0; mt < CT; ++mt) {                                                      \
                const bool ok_ = 16 * mt + 4 * q < C;                                                                \
                HV[mt] = *reinterpret_cast<const float4*>(ok_ ? p.hc + TMG_PXO(gp_, p.hcs) + 16 * mt + 4 * q : tmg_zero_page); \
                XV[mt] = *reinterpret_cast<const float2*>(ok_ ? p.x2 + TMG_PXO(gp_, p.x2s) + 8 * mt + 2 * q : tmg_zero_page); \
            }                                                                                                        \
        }
        TMG_CPL_EPI_LOAD(hcur, xcur, 0)
        float ld = 0.f;
#pragma unroll 1
        for (int nt = 0; nt < 4; ++nt) {
            float4 hnxt[CT];
            float2 xnxt[CT];
            TMG_CPL_EPI_LOAD(hnxt, xnxt, min(nt + 1, 3))
            // zero conv: 9 taps x K4 quads, fragments at base + immediate
            f32x4 acc[CT];
#pragma unroll
            for (int mt = 0; mt < CT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* xbn = xb + nt * (PW * 2);
            const float* dbn = db + ((q < 2) ? nt * (PW * 2) : 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int tyy = t / 3, txx = t % 3;
#pragma unroll
                for (int s = 0; s < K4; ++s) {
                    const float bf = (s < CH4) ? xbn[((2 * s) * PP + tyy * PW + txx) * 2] : dbn[(tyy * PW + txx) * 2];
#pragma unroll
                    for (int mt = 0; mt < CT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t][s][mt], bf, acc[mt], 0, 0, 0);
                }
            }
            // affine coupling in registers
            const int gy = oy0 + row0 + nt;
            const bool pin = gy < p.H && gx < p.W;
            const unsigned gp = img + (unsigned)min(gy, p.H - 1) * (unsigned)p.W + (unsigned)gxc;
            float y2r[CT][2];
#pragma unroll
            for (int mt = 0; mt < CT; ++mt) {
                const float4 bz4 = bzv[mt], h4 = hcur[mt];
                const float hh0 = (acc[mt][0] + h4.x + bz4.x) * osc, hh1 = (acc[mt][1] + h4.y + bz4.y) * osc;
                const float hh2 = (acc[mt][2] + h4.z + bz4.z) * osc, hh3 = (acc[mt][3] + h4.w + bz4.w) * osc;
                const float sg0 = 2.f * hh1 * cpl_rcp(1.f + fabsf(hh1)), sg1 = 2.f * hh3 * cpl_rcp(1.f + fabsf(hh3));
                const float2 xv = xcur[mt];
                float o0, o1;
                if (p.reverse) { o0 = xv.x * cpl_exp(-sg0) - hh0; o1 = xv.y * cpl_exp(-sg1) - hh2; }
                else           { o0 = (xv.x + hh0) * cpl_exp(sg0); o1 = (xv.y + hh2) * cpl_exp(sg1); }
                const bool ok = pin && 8 * mt + 2 * q < ch;
                y2r[mt][0] = ok ? o0 : 0.f;
                y2r[mt][1] = ok ? o1 : 0.f;
                if (ok) {
                    ld += sg0 + sg1;
                    *reinterpret_cast<float2*>(p.rsave + TMG_PXO(gp, ch) + 8 * mt + 2 * q) = make_float2(hh1, hh3);
                    if (p.y2save) *reinterpret_cast<float2*>(p.y2save + TMG_PXO(gp, ch) + 8 * mt + 2 * q) = make_float2(o0, o1);
                }
            }
            if (p.Wm) {
                // channel mix: [x1; y2] is the B operand as it stands
                f32x4 oacc[CT];
#pragma unroll
                for (int mo = 0; mo < CT; ++mo) oacc[mo] = (f32x4){bmv[mo].x, bmv[mo].y, bmv[mo].z, bmv[mo].w};
                const float* xrn = xrb + nt * 32;
#pragma unroll
                for (int t = 0; t < KM; ++t) {
                    const float bfm = (t < CH4) ? xrn[(2 * t) * 512] : y2r[(t - CH4) >> 1][(t - CH4) & 1];
#pragma unroll
                    for (int mo = 0; mo < CT; ++mo) oacc[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm[mo][t], bfm, oacc[mo], 0, 0, 0);
                }
#pragma unroll
                for (int mo = 0; mo < CT; ++mo) {
                    const int c = 16 * mo + 4 * q;      // a channel quad lies in one half (ch is a multiple of 4)
                    if (pin && c < C)
                        *reinterpret_cast<float4*>(c < ch ? p.out + TMG_PXO(gp, p.os) + c : p.out2 + TMG_PXO(gp, p.o2s) + (c - ch)) =
                            make_float4(oacc[mo][0], oacc[mo][1], oacc[mo][2], oacc[mo][3]);
                }
            } else {
                // no trailing mix: out = [x1 | y2]
                const int cpx = (row0 + nt) * 16 + li;
#pragma unroll
                for (int mt = 0; mt < CT; ++mt)
                    if (pin && 8 * mt + 2 * q < ch) {
                        *reinterpret_cast<float2*>(p.out2 + TMG_PXO(gp, p.o2s) + 8 * mt + 2 * q) = make_float2(y2r[mt][0], y2r[mt][1]);
                        const int c = 8 * mt + 2 * q;
                        *reinterpret_cast<float2*>(p.out + TMG_PXO(gp, p.os) + c) = *reinterpret_cast<const float2*>(XR + ((c >> 1) * 256 + cpx) * 2);
                    }
            }
#pragma unroll
            for (int mt = 0; mt < CT; ++mt) { hcur[mt] = hnxt[mt]; xcur[mt] = xnxt[mt]; }
        }
#undef TMG_CPL_EPI_LOAD
        if (b != ldb) {
            if (ldb >= 0) {
                ldacc = wave_sum(ldacc);
                if (lane == 0) atomicAdd(p.logdet + ldb, ldacc);
            }
            ldb = b;
            ldacc = 0.f;
        }
        ldacc += ld;
    }
    if (ldb >= 0) {
        ldacc = wave_sum(ldacc);
        if (lane == 0) atomicAdd(p.logdet + ldb, ldacc);
    }
#undef TMG_CPL_ISSUE
#undef TMG_CPL_ORIGIN
}

template <int CT, int K4>
static int launch_cpl_fwd(const CplFP& p, hipStream_t st) {
    constexpr int CH4 = K4 - 1, CHP = 2 * CH4;
    const size_t lds = (2 * ((size_t)(CHP + 1) * 324 * 2 + (size_t)CHP * 256 * 2) + (5 * 18 + 4) * 2) * sizeof(float);
    if (lds > 64 * 1024) TMG_LDS_OPTIN((&cpl_fwd_kernel<CT, K4>));
    // blocks: one resident wave (3 per CU at C <= 16, 2 above), every block with the same number of tiles (measured at 64 x 128 x 128 and
    // 64 x 64 x 64: 768 / 512; an uneven 5-or-6 split or a second wave of blocks costs 10-20 %)
    static const int gcap = getenv("TMG_CPL_GRID") ? atoi(getenv("TMG_CPL_GRID")) : (CT == 1 ? 3 : 2) * tmg_num_cus();   // 768 / 512 on an MI355X
    const int per_blk = (p.ntiles + gcap - 1) / gcap;
    const int grid = (p.ntiles + per_blk - 1) / per_blk;
    // algorithmic HBM bytes: x (C), D (4), hc (C) read; out (C), r (C/2), y2 (C/2) written
    TmgProf prof(TMG_PROF_CPL, 4.0 * p.B * (double)p.H * p.W * (3.0 * p.C + 4 + (p.y2save ? 1.0 : 0.5) * p.C), st);
    hipLaunchKernelGGL((cpl_fwd_kernel<CT, K4>), dim3(grid), dim3(256), lds, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

// Zero conv + affine coupling + log-det (+ trailing channel mix) of one coupling layer in one launch (see the file header).
// Tensors are NHWC fp32; every channel count / stride is a multiple of 4.
// dims = {B, H, W, C, reverse, x pixel stride, out pixel stride, hc pixel stride, row length of wz, column of d1 in wz}.
// Returns -100 when the shape is outside the kernel's envelope (C/2 a multiple of 4, 8 <= C <= 32): the caller uses the per-op path.
extern "C" int tmg_coupling_fwd_halves(const void* x1, const void* x2, void* out1, void* out2, void* rsave, void* y2save, const void* D,
                                       const void* hc, const void* wz, const void* bz, const void* kappa, const void* Wm, const void* bm,
                                       void* logdet, const int64_t* dims, hipStream_t st);

extern "C" int tmg_coupling_fwd(const void* x, void* out, void* rsave, void* y2save, const void* D, const void* hc, const void* wz,
                                const void* bz, const void* kappa, const void* Wm, const void* bm, void* logdet, const int64_t* dims,
                                hipStream_t st) {
    // one interleaved [npix][C] tensor each: the second halves start C / 2 channels in, same pixel stride
    const int64_t ch = dims[3] / 2;
    const int64_t d2[12] = {dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], dims[6], dims[7], dims[8], dims[9], dims[5], dims[6]};
    return tmg_coupling_fwd_halves(x, (const float*)x + ch, out, (float*)out + ch, rsave, y2save, D, hc, wz, bz, kappa, Wm, bm, logdet, d2, st);
}

// As tmg_coupling_fwd with the channel halves of the layer input and output addressed separately (see CplFP).
// dims = {B, H, W, C, reverse, x1 pixel stride, out1 pixel stride, hc pixel stride, row length of wz, column of d1 in wz, x2 pixel
// stride, out2 pixel stride}.
extern "C" int tmg_coupling_fwd_halves(const void* x1, const void* x2, void* out1, void* out2, void* rsave, void* y2save, const void* D,
                                       const void* hc, const void* wz, const void* bz, const void* kappa, const void* Wm, const void* bm,
                                       void* logdet, const int64_t* dims, hipStream_t st) {
    CplFP p;
    p.xmap = tmg_xcd_map_on();
    p.B = (int)dims[0]; p.H = (int)dims[1]; p.W = (int)dims[2]; p.C = (int)dims[3]; p.reverse = (int)dims[4];
    p.x = (const float*)x1; p.xs = (int)dims[5];
    p.x2 = (const float*)x2; p.x2s = (int)dims[10];
    p.out = (float*)out1; p.os = (int)dims[6];
    p.out2 = (float*)out2; p.o2s = (int)dims[11];
    p.rsave = (float*)rsave; p.y2save = (float*)y2save; p.D = (const float*)D;
    p.hc = (const float*)hc; p.hcs = (int)dims[7];
    p.wz = (const float*)wz; p.wz_rows = (int)dims[8]; p.wz_d1col = (int)dims[9];
    p.bz = (const float*)bz; p.kappa = (const float*)kappa; p.Wm = (const float*)Wm; p.bm = (const float*)bm;
    p.logdet = (float*)logdet;
    const int ch = p.C / 2;
    if (p.C < 8 || p.C > 32 || (ch & 3) || (p.xs & 3) || (p.os & 3) || (p.hcs & 3) || (p.x2s & 3) || (p.o2s & 3)) return -100;
    if ((((uintptr_t)x1) | ((uintptr_t)x2) | ((uintptr_t)out1) | ((uintptr_t)out2)) & 15) return -100;
    p.tiles_x = (p.W + 15) / 16; p.tiles_y = (p.H + 15) / 16; p.ntiles = p.B * p.tiles_x * p.tiles_y;
    if (p.ntiles <= 0) return 0;
    if ((long long)p.B * p.H * p.W >= (1LL << 31)) return -100;     // 32-bit pixel indices
    switch (ch / 4) {
        case 1: return launch_cpl_fwd<1, 2>(p, st);   // C = 8
        case 2: return launch_cpl_fwd<1, 3>(p, st);   // C = 16
        case 3: return launch_cpl_fwd<2, 4>(p, st);   // C = 24
        case 4: return launch_cpl_fwd<2, 5>(p, st);   // C = 32
    }
    return -100;
}

// =================================================================================================================================
// Backward companion (generative direction, the one the reference trains through): channel-mix input gradient -> affine-coupling
// backward -> zero-conv input gradient with the EXACT adjoint of the replicate padding, one launch per layer instead of
// conv 1x1 (dgrad) + affine_bwd + conv 3x3 (dgrad) + border fix.
//
//   dto  = Wm^T dout                                     (glowConv.py:207-222 + actNorm.py:71-85 under autograd)
//   dtin2 = dto2 e^{-sg};  da = -dto2;  dsg = -2 dto2 (tin2 e^{-sg}) + 2 g_b;  dr = dsg / (1 + |r|)^2      (flowAffine.py:102-109)
//   dhh = e^kappa interleave(da, dr)                     -> DH (kept for the level-wide conditioning contractions and the weight gradients)
//   G   = sum_p sum_t [clamp(p + t - 1) == q] Wz_t^T dhh(p)   over the channels (x1 | d1, d2)               (flowUtils.py:246-247)
//
// Phase A is pointwise: every 16 pixels of the tile + 1 halo pixel (324 pixels = 21 n-tiles dealt to the 4 waves) run through the
// mix MFMA straight from global memory (B operand = one float4 of dout per lane; the k-order is permuted to match, as in
// tmg_mix_f16), the coupling backward runs on the accumulator registers and leaves dhh in LDS as channel pairs.  Phase B is the
// 3x3 transposed contraction over that LDS patch.  Replicate padding: an input position outside the image is a copy of its clamped
// neighbour, so a border pixel q also collects the taps that would have landed on the ring: for the top image row the taps
// (2, ux) applied to the data of row q.y itself (instead of q.y + 1), likewise bottom / left / right, plus one tap for each image
// corner.  These extra MFMA steps are skipped (wave- / block-uniform branches) away from the border.
struct CplBP {
    // channel halves addressed separately, as in CplFP: half 1 = [0, ch) at base + pixel * stride, half 2 = [ch, C) at base2 + ..
    const float* dout; int dos;      // gradient w.r.t. the layer output, half 1
    const float* dout2; int do2s;    //                                   half 2
    const float* x; int xs;          // tin2: the SECOND half of the layer input (the only one read), channel 0 = model channel ch
    const float* r;                  // [npix][ch] saved softsign arguments
    const float* g;                  // [B] gradient arriving on the log-det (null: 0)
    const float* Wm;                 // [C][C] trailing mix of the forward pass
    const float* wz; int wz_rows, wz_d1col;
    const float* kappa;
    float* DH; int dhs;              // [npix] x C slice: e^kappa * dhh
    float* dtin; int dts;            // gradient w.r.t. the layer input, half 1 = dto1 (pass-through gradient, completed by dense2_bwd)
    float* dtin2; int dt2s;          //                                  half 2 = dtin2
    float* G0;                       // [npix][ch]
    float* GD;                       // [npix][4]
    int B, H, W, C;
    int tiles_x, tiles_y, ntiles;
    int xmap;                        // XCD-aware tile order (tmg_common.h)
    // fwd = 1: the DENSITY direction's layer (mix -> coupling, flowAffine.py:76-83): no mix in front of the coupling's backward (dout
    // IS the gradient w.r.t. the coupling output; the mix input gradient is a launch of its own after tmg_dense2_bwd), x = the second
    // half of the coupling OUTPUT y2, and  dtin2 = dy2 e^{sg};  da = dy2 e^{sg};  dsg = 2 dy2 y2 + 2 g_b;  dr = dsg / (1 + |r|)^2
    int fwd;
};

// CT = 16-channel tiles of C; MT = 16-channel tiles of the dgrad output (ch + 2 channels); KS = C / 4 channel quads of dhh
// PERM (C <= 16, generative direction): the rows of the transposed mix are dealt so that EVERY lane's accumulator quad holds two
// pass-through channels (2 q, 2 q + 1 of dto1) and two coupling channels (ch + 2 q, ch + 2 q + 1).  In channel order the lanes q < 2
// hold only pass-through channels and the lanes q >= 2 only coupling channels, so the wave executed BOTH epilogues under masks - the
// coupling arithmetic of four elements per lane for half of the lanes; dealt out, every lane does two elements and the wave issues half
// of those instructions (the kernel is bound by vector-instruction issue: ablation table in DESIGN.md).
template <int CT, int MT, int KS, bool PERM = false>
__global__ __launch_bounds__(256, CT == 1 ? 2 : 2) void cpl_bwd_kernel(CplBP p) {
    static_assert(!PERM || CT == 1, "the dealt row order is built for one 16-row tile");
    constexpr int PW = 18, PP = PW * PW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int C = p.C, ch = C >> 1;
    float* DHL = lds;                                 // [ch][PP][2]: dhh pairs (da_j, dr_j) e^kappa on the tile + 1 halo pixel, zero outside the image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const float osc = out_scale_of(p.kappa);

    // mix A fragments (transposed mix): k-step (h, e) carries dout channel 16 h + 4 q + e
    float wmT[CT][4 * CT];
#pragma unroll
    for (int mo = 0; mo < CT; ++mo)
#pragma unroll
        for (int t = 0; t < 4 * CT; ++t) {
            const int c = 16 * (t >> 2) + 4 * q + (t & 3);
            int i = 16 * mo + li;
            if (PERM) {      // accumulator row li = 4 q' + e of lane group q': e < 2 pass-through channel 2 q' + e, else coupling channel ch + 2 q' + e - 2
                const int qp = li >> 2, e = li & 3, j = 2 * qp + (e & 1);
                i = j < ch ? (e < 2 ? j : ch + j) : C;
            }
            wmT[mo][t] = *((c < C && i < C) ? p.Wm + (size_t)c * C + i : tmg_zero_page);
        }
    // dgrad A fragments: A[i = li][k = q] of (tap u, quad s) = Wz[co = 4 s + q][col(i)][8 - u], rows i = (x1 | d1, d2)
    float wa[9][KS][MT];
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int i = 16 * mt + li, co = 4 * s + q;
                const int col = i < ch ? i : (i < ch + 2 ? p.wz_d1col + (i - ch) : -1);
                wa[u][s][mt] = *((co < C && col >= 0) ? p.wz + ((size_t)co * p.wz_rows + col) * 9 + (8 - u) : tmg_zero_page);
            }

    const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t0 = tmg_xcd_block((int)blockIdx.x, (int)gridDim.x, p.xmap) * per, t1 = min(t0 + per, p.ntiles);
    for (int tile = t0; tile < t1; ++tile) {
        // lane coordinates re-derived from an opaque copy of the thread index per tile: the index arithmetic built on them is
        // tile-invariant, the compiler hoists it out of the loop and - at C = 32, where the weight fragments hold 160 registers - spills
        // it (40 bytes per lane, reloaded in every tile); recomputing it costs a handful of integer instructions per tile
        int tidl = tid;
        asm volatile("" : "+v"(tidl));
        const int lane = tidl & 63, wave = tidl >> 6, li = lane & 15, q = lane >> 4;
        int t_ = tile;
        const int tx = t_ % p.tiles_x; t_ /= p.tiles_x;
        const int ty = t_ % p.tiles_y;
        const int b = t_ / p.tiles_y;
        const int oy0 = ty * 16, ox0 = tx * 16;
        const unsigned img = (unsigned)b * (unsigned)(p.H * p.W);
        const float gb = p.g ? p.g[b] : 0.f;
        // ---- phase A: n-tiles of 16 region pixels (linear index) -------------------------------------------------------------
#define TMG_CPLB_LOAD(DV, RV, TV, NT)                                                                                  \
        {                                                                                                              \
            const int rp_ = min((NT) * 16 + li, PP - 1);                                                               \
            const int ry_ = rp_ / PW, rx_ = rp_ - ry_ * PW;                                                            \
            const int gy_ = oy0 - 1 + ry_, gx_ = ox0 - 1 + rx_;                                                        \
            const bool in_ = gy_ >= 0 && gy_ < p.H && gx_ >= 0 && gx_ < p.W;                                          \
            const unsigned gp_ = img + (unsigned)min(max(gy_, 0), p.H - 1) * (unsigned)p.W + (unsigned)min(max(gx_, 0), p.W - 1); \
            _Pragma("unroll") for (int h = 0; h < CT; ++h) {                                                           \
                const int c_ = 16 * h + 4 * q;                                                                         \
                DV[h] = *reinterpret_cast<const float4*>((in_ && c_ < C) ? (c_ < ch ? p.dout + TMG_PXO(gp_, p.dos) + c_ : p.dout2 + TMG_PXO(gp_, p.do2s) + (c_ - ch)) : tmg_zero_page);  \
                if (PERM) {   /* every lane: r / tin2 of its two coupling channels j0 = 2 q, 2 q + 1 */                \
                    const bool two_ = in_ && 2 * q < ch;                                                               \
                    const float2 r2_ = *reinterpret_cast<const float2*>(two_ ? p.r + TMG_PXO(gp_, ch) + 2 * q : tmg_zero_page);    \
                    const float2 t2_ = *reinterpret_cast<const float2*>(two_ ? p.x + TMG_PXO(gp_, p.xs) + 2 * q : tmg_zero_page);  \
                    RV[h] = make_float4(r2_.x, r2_.y, 0.f, 0.f); TV[h] = make_float4(t2_.x, t2_.y, 0.f, 0.f);         \
                } else {                                                                                               \
                /* the lanes whose accumulator quad of m-tile h is a dto2 quad (channel c_ >= ch) need r / tin2 of j0 = c_ - ch */ \
                const bool two_ = in_ && c_ >= ch && c_ < C;                                                           \
                RV[h] = *reinterpret_cast<const float4*>(two_ ? p.r + TMG_PXO(gp_, ch) + (c_ - ch) : tmg_zero_page);           \
                TV[h] = *reinterpret_cast<const float4*>(two_ ? p.x + TMG_PXO(gp_, p.xs) + (c_ - ch) : tmg_zero_page);         \
                }                                                                                                      \
            }                                                                                                          \
        }
        constexpr int NTA = (PP + 15) / 16;   // 21
        float4 dcu[CT], rcu[CT], tcu[CT];
        TMG_CPLB_LOAD(dcu, rcu, tcu, wave)
#pragma unroll 1
        for (int nt = wave; nt < NTA; nt += 4) {
            float4 dnx[CT], rnx[CT], tnx[CT];
            TMG_CPLB_LOAD(dnx, rnx, tnx, min(nt + 4, NTA - 1))
            const int rp = nt * 16 + li;
            const int ry = rp / PW, rx = rp - ry * PW;
            const int gy = oy0 - 1 + ry, gx = ox0 - 1 + rx;
            const bool inimg = rp < PP && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const bool center = inimg && ry >= 1 && ry <= 16 && rx >= 1 && rx <= 16;
            const unsigned gp = img + (unsigned)min(max(gy, 0), p.H - 1) * (unsigned)p.W + (unsigned)min(max(gx, 0), p.W - 1);
            f32x4 acc[CT];
            if (p.fwd) {
                // density direction: the gradient arrives at the coupling output itself (this lane's quad of m-tile mo is the quad it loaded)
#pragma unroll
                for (int mo = 0; mo < CT; ++mo) acc[mo] = (f32x4){dcu[mo].x, dcu[mo].y, dcu[mo].z, dcu[mo].w};
            } else {
#pragma unroll
                for (int mo = 0; mo < CT; ++mo) acc[mo] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < CT; ++h) {
                    const float dv[4] = {dcu[h].x, dcu[h].y, dcu[h].z, dcu[h].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mo = 0; mo < CT; ++mo)
                            acc[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(wmT[mo][4 * h + e], dv[e], acc[mo], 0, 0, 0);
                }
            }
            if constexpr (PERM) {
                // this lane: acc[0][0..1] = dto1 channels 2 q, 2 q + 1; acc[0][2..3] = dto2 of the coupling channels j0 = 2 q, 2 q + 1
                const int j0 = 2 * q;
                if (j0 < ch) {
                    if (center) *reinterpret_cast<float2*>(p.dtin + TMG_PXO(gp, p.dts) + j0) = make_float2(acc[0][0], acc[0][1]);
                    const float rr[2] = {rcu[0].x, rcu[0].y}, tt[2] = {tcu[0].x, tcu[0].y};
                    float di[2], da[2], dr[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float den = 1.f + fabsf(rr[e]), rden = cpl_rcp(den);
                        const float go = acc[0][2 + e];
                        const float inv = cpl_exp(-2.f * rr[e] * rden);
                        di[e] = go * inv;
                        da[e] = inimg ? -go * osc : 0.f;
                        dr[e] = inimg ? osc * (-2.f * go * (tt[e] * inv) + 2.f * gb) * (rden * rden) : 0.f;
                    }
                    if (rp < PP) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) *reinterpret_cast<float2*>(DHL + ((j0 + e) * PP + rp) * 2) = make_float2(da[e], dr[e]);
                    }
                    if (center) {
                        *reinterpret_cast<float2*>(p.dtin2 + TMG_PXO(gp, p.dt2s) + j0) = make_float2(di[0], di[1]);
                        *reinterpret_cast<float4*>(p.DH + TMG_PXO(gp, p.dhs) + 2 * j0) = make_float4(da[0], dr[0], da[1], dr[1]);
                    }
                }
            } else
#pragma unroll
            for (int mo = 0; mo < CT; ++mo) {
                const int c = 16 * mo + 4 * q;
                if (c < ch) {
                    // pass-through half: dto1 -> first half of dtin (dense2_bwd adds the coupling network's share)
                    if (center) *reinterpret_cast<float4*>(p.dtin + TMG_PXO(gp, p.dts) + c) = make_float4(acc[mo][0], acc[mo][1], acc[mo][2], acc[mo][3]);
                } else if (c < C) {
                    const int j0 = c - ch;
                    const float rr[4] = {rcu[mo].x, rcu[mo].y, rcu[mo].z, rcu[mo].w};
                    const float tt[4] = {tcu[mo].x, tcu[mo].y, tcu[mo].z, tcu[mo].w};
                    float di[4], da[4], dr[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float den = 1.f + fabsf(rr[e]), rden = cpl_rcp(den);
                        const float go = acc[mo][e];
                        if (p.fwd) {
                            const float sc = cpl_exp(2.f * rr[e] * rden);          // e^{sg}
                            di[e] = go * sc;
                            da[e] = inimg ? go * sc * osc : 0.f;
                            dr[e] = inimg ? osc * (2.f * go * tt[e] + 2.f * gb) * (rden * rden) : 0.f;   // tt = y2
                        } else {
                            const float inv = cpl_exp(-2.f * rr[e] * rden);
                            di[e] = go * inv;
                            da[e] = inimg ? -go * osc : 0.f;
                            dr[e] = inimg ? osc * (-2.f * go * (tt[e] * inv) + 2.f * gb) * (rden * rden) : 0.f;
                        }
                    }
                    if (rp < PP) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) *reinterpret_cast<float2*>(DHL + ((j0 + e) * PP + rp) * 2) = make_float2(da[e], dr[e]);
                    }
                    if (center) {
                        *reinterpret_cast<float4*>(p.dtin2 + TMG_PXO(gp, p.dt2s) + j0) = make_float4(di[0], di[1], di[2], di[3]);
                        *reinterpret_cast<float4*>(p.DH + TMG_PXO(gp, p.dhs) + 2 * j0) = make_float4(da[0], dr[0], da[1], dr[1]);
                        *reinterpret_cast<float4*>(p.DH + TMG_PXO(gp, p.dhs) + 2 * j0 + 4) = make_float4(da[2], dr[2], da[3], dr[3]);
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < CT; ++h) { dcu[h] = dnx[h]; rcu[h] = rnx[h]; tcu[h] = tnx[h]; }
        }
#undef TMG_CPLB_LOAD
        __syncthreads();
        // ---- phase B: transposed 3x3 contraction; this wave owns tile rows 4 wave .. 4 wave + 3 ------------------------------------
        const int row0 = 4 * wave;
        const float* hb = DHL + (((q >> 1) * PP + row0 * PW + li) * 2 + (q & 1));   // + plane 2 s, + (row + uy) * PW + ux
        const int gx = ox0 + li;
        const bool left = gx == 0, right = gx == p.W - 1;
#pragma unroll 1
        for (int nt = 0; nt < 4; ++nt) {
            const int gy = oy0 + row0 + nt;
            const float* hbn = hb + nt * (PW * 2);
            f32x4 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define TMG_CPLB_TAP(U, DY, DX, MASK)                                                                                  \
            _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                           \
                float bf = hbn[((2 * s) * PP + (1 + (DY)) * PW + 1 + (DX)) * 2];                                       \
                bf = (MASK) ? bf : 0.f;                                                                                \
                _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                      \
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[U][s][mt], bf, acc[mt], 0, 0, 0);               \
            }
            // regular taps: G(q) += W^T_u dhh(q + u - 1)
#pragma unroll
            for (int u = 0; u < 9; ++u) { TMG_CPLB_TAP(u, u / 3 - 1, u % 3 - 1, true) }
            // replicate-padding adjoint (see the header): border rows / columns / corners collect the taps of the ring
            if (gy == 0) {
#pragma unroll
                for (int ux = 0; ux < 3; ++ux) { TMG_CPLB_TAP(6 + ux, 0, ux - 1, true) }
                if (ox0 == 0) { TMG_CPLB_TAP(8, 0, 0, left) }
                if (ox0 + 16 >= p.W) { TMG_CPLB_TAP(6, 0, 0, right) }
            }
            if (gy == p.H - 1) {
#pragma unroll
                for (int ux = 0; ux < 3; ++ux) { TMG_CPLB_TAP(ux, 0, ux - 1, true) }
                if (ox0 == 0) { TMG_CPLB_TAP(2, 0, 0, left) }
                if (ox0 + 16 >= p.W) { TMG_CPLB_TAP(0, 0, 0, right) }
            }
            if (ox0 == 0) {
#pragma unroll
                for (int uy = 0; uy < 3; ++uy) { TMG_CPLB_TAP(3 * uy + 2, uy - 1, 0, left) }
            }
            if (ox0 + 16 >= p.W) {
#pragma unroll
                for (int uy = 0; uy < 3; ++uy) { TMG_CPLB_TAP(3 * uy, uy - 1, 0, right) }
            }
#undef TMG_CPLB_TAP
            if (gy < p.H && gx < p.W) {
                const unsigned gp = img + (unsigned)gy * (unsigned)p.W + (unsigned)gx;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int c = 16 * mt + 4 * q;
                    const float4 v = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
                    if (c < ch) *reinterpret_cast<float4*>(p.G0 + TMG_PXO(gp, ch) + c) = v;
                    else if (c == ch) *reinterpret_cast<float4*>(p.GD + TMG_PXO(gp, 4)) = v;   // (d1, d2, 0, 0): rows past ch + 1 carry zero weights
                }
            }
        }
        __syncthreads();   // the patch is rewritten by the next tile's phase A
    }
}

template <int CT, int MT, int KS>
static int launch_cpl_bwd(const CplBP& p, hipStream_t st) {
    const size_t lds = (size_t)(p.C / 2) * 324 * 2 * sizeof(float);
    if (lds > 64 * 1024) TMG_LDS_OPTIN((&cpl_bwd_kernel<CT, MT, KS>));
    // blocks: measured at 64 x 128 x 128 (C = 16: 4 096 tiles) 256: 193 us, 512: 140, 768: 139, 1 024: 119, 2 048: 124, 4 096: 136 - two
    // resident blocks per CU, twice as many blocks as that with an even tile count each; at C = 32 (1 024 tiles) 512: 87, 768: 90, 1 024: 100
    static const int gcap = getenv("TMG_CPL_GRID") ? atoi(getenv("TMG_CPL_GRID")) : (CT == 1 ? 4 : 2) * tmg_num_cus();   // 1 024 / 512 on an MI355X
    const int per_blk = (p.ntiles + gcap - 1) / gcap;
    const int grid = (p.ntiles + per_blk - 1) / per_blk;
    // algorithmic HBM bytes: dout (C), r (C/2), tin2 (C/2) read; DH (C), dtin (C), G0 (C/2), GD (4) written
    TmgProf prof(TMG_PROF_CPLB, 4.0 * p.B * (double)p.H * p.W * (4.5 * p.C + 4), st);
    if constexpr (CT == 1) {
        if (!p.fwd) {
            hipLaunchKernelGGL((cpl_bwd_kernel<CT, MT, KS, true>), dim3(grid), dim3(256), lds, st, p);
            TMG_CHECK_LAUNCH();
            return 0;
        }
    }
    hipLaunchKernelGGL((cpl_bwd_kernel<CT, MT, KS>), dim3(grid), dim3(256), lds, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

// Backward of tmg_coupling_fwd's generative-direction layer up to the coupling network's input gradients (see above).
// dims = {B, H, W, C, dout pixel stride, x pixel stride, DH pixel stride, dtin pixel stride, row length of wz, column of d1 in wz}.
// Returns -100 outside the envelope (8 <= C <= 32, C/2 a multiple of 4).
extern "C" int tmg_coupling_bwd_halves(const void* dout1, const void* dout2, const void* x2, const void* r, const void* g, const void* Wm,
                                       const void* wz, const void* kappa, void* DH, void* dtin1, void* dtin2, void* G0, void* GD,
                                       const int64_t* dims, hipStream_t st);

extern "C" int tmg_coupling_bwd(const void* dout, const void* x, const void* r, const void* g, const void* Wm, const void* wz,
                                const void* kappa, void* DH, void* dtin, void* G0, void* GD, const int64_t* dims, hipStream_t st) {
    const int64_t ch = dims[3] / 2;
    const int64_t d2[13] = {dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], dims[6], dims[7], dims[8], dims[9], dims[4], dims[7], 0};
    return tmg_coupling_bwd_halves(dout, (const float*)dout + ch, (const float*)x + ch, r, g, Wm, wz, kappa, DH, dtin, (float*)dtin + ch, G0, GD,
                                   d2, st);
}

// As tmg_coupling_bwd with the channel halves of dout / dtin addressed separately and x2 = the second half of the layer input.
// dims = {B, H, W, C, dout1 pixel stride, x2 pixel stride, DH pixel stride, dtin1 pixel stride, row length of wz, column of d1 in wz,
// dout2 pixel stride, dtin2 pixel stride, density direction (0 / 1: see CplBP::fwd - then x2 is the second half of the coupling OUTPUT)}.
extern "C" int tmg_coupling_bwd_halves(const void* dout1, const void* dout2, const void* x2, const void* r, const void* g, const void* Wm,
                                       const void* wz, const void* kappa, void* DH, void* dtin1, void* dtin2, void* G0, void* GD,
                                       const int64_t* dims, hipStream_t st) {
    CplBP p;
    p.xmap = tmg_xcd_map_on();
    p.B = (int)dims[0]; p.H = (int)dims[1]; p.W = (int)dims[2]; p.C = (int)dims[3];
    p.dout = (const float*)dout1; p.dos = (int)dims[4];
    p.dout2 = (const float*)dout2; p.do2s = (int)dims[10];
    p.x = (const float*)x2; p.xs = (int)dims[5];
    p.r = (const float*)r; p.g = (const float*)g; p.Wm = (const float*)Wm;
    p.wz = (const float*)wz; p.wz_rows = (int)dims[8]; p.wz_d1col = (int)dims[9];
    p.kappa = (const float*)kappa;
    p.DH = (float*)DH; p.dhs = (int)dims[6];
    p.dtin = (float*)dtin1; p.dts = (int)dims[7];
    p.dtin2 = (float*)dtin2; p.dt2s = (int)dims[11];
    p.fwd = (int)dims[12];
    p.G0 = (float*)G0; p.GD = (float*)GD;
    const int ch = p.C / 2;
    if (p.C < 8 || p.C > 32 || (ch & 3) || (p.dos & 3) || (p.xs & 3) || (p.dhs & 3) || (p.dts & 3) || (p.do2s & 3) || (p.dt2s & 3)) return -100;
    if ((((uintptr_t)dout1) | ((uintptr_t)dout2) | ((uintptr_t)x2) | ((uintptr_t)dtin1) | ((uintptr_t)dtin2)) & 15) return -100;
    p.tiles_x = (p.W + 15) / 16; p.tiles_y = (p.H + 15) / 16; p.ntiles = p.B * p.tiles_x * p.tiles_y;
    if (p.ntiles <= 0) return 0;
    if ((long long)p.B * p.H * p.W >= (1LL << 31)) return -100;     // 32-bit pixel indices
    switch (ch / 4) {
        case 1: return launch_cpl_bwd<1, 1, 2>(p, st);   // C = 8:  outputs ch + 2 = 6
        case 2: return launch_cpl_bwd<1, 1, 4>(p, st);   // C = 16: 10
        case 3: return launch_cpl_bwd<2, 1, 6>(p, st);   // C = 24: 14
        case 4: return launch_cpl_bwd<2, 2, 8>(p, st);   // C = 32: 18
    }
    return -100;
}
