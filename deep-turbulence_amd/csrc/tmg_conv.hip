// Implicit-GEMM convolution for gfx950 on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// One kernel family serves every dense contraction on the TM-Glow hot path:
//   * Conv2dZeros 3x3 with replicate padding           (reference flowUtils.py:246-247)
//   * ConvLSTM gate conv and residual out-conv 3x3     (reference convLSTM.py:72-74, :150-152)
//   * the encoder's 3x3 / stride-2 convs               (reference tmGlow.py:88-95,148-156,180-182)
//   * the invertible 1x1 channel mix (ksize = 1)       (reference glowConv.py:193-194, :219-220)
//   * every input-gradient of the above (same kernel, transposed + flipped packed weights)
// plus the weight-gradient kernel (contraction over pixels) and the weight packer.
//
// Layout: activations NHWC fp32.  GEMM view: M = output pixels of a TH x TW tile, N = output
// channels, K = taps x input channels.  The input patch (tile + halo) is staged through LDS once per
// channel chunk with padding / ReLU / per-channel affine applied on the way in; A fragments come from
// LDS with ds_read_b128 (4 k-steps per read), B fragments come straight from the L2-resident packed
// weights with 16-byte global loads (4 k-steps per load).  fp32 MFMA issues one 16x16x4 tile per 32
// cycles per SIMD, so the kernel is matrix-pipe bound by construction and LDS traffic is negligible.
#include "tmg_common.h"
#include <stdlib.h>

// Zero-filled global memory: lanes of a staging batch that have nothing to read (zero padding, channels past the end,
// items past the end) load from here, so the batch needs no control flow.  A load inside a divergent block makes the
// compiler drain vmcnt before the next load (the value meets a zero at the join), which serialises the whole batch.
static __device__ float g_tmg_zero_page[64];

struct ConvP {
    TmgSeg in[TMG_MAX_IN_SEG];
    int nseg;
    int vec4;  // every segment offset / width / stride is a multiple of 4 -> float4 staging
    int B, Hin, Win, Hout, Wout;
    int ksize, stride;
    int Cin, Cin_pad, Cout, Cout_pad;
    const float* wpk;       // [taps][Cin_pad/16][Cout_pad][16]
    const float* bias;      // [Cout] or null
    const float* kappa;     // device scalar: output multiplied by exp(clamp(kappa,-4,ln4)); null -> 1
    const float* in_scale;  // optional per-input-channel affine applied before ReLU (BatchNorm fold)
    const float* in_shift;
    int relu_in, pad_rep, relu_out, accumulate;
    TmgOSeg out[TMG_MAX_OUT_SEG];
    int nout;
    TmgSeg add;  // optional tensor added to the accumulator before bias / scale (p == null: none); n == Cout
    int TW_log2, TH;
    int KCH;  // channels staged per LDS chunk (multiple of 16)
    int tiles_x, tiles_y;
    int ntiles, nchunks;  // conv_fwd_kernel only: tiles of the whole batch, channel chunks per tile
    int ovec4;            // conv_fwd_kernel only: outputs / add operand are float4-addressable along channels
};

// MT m-tiles (16 px each) x NTW n-tiles (16 ch each) per wave; WM x WN waves per block (WM*WN == 4).
template <int MT, int NTW, int WM, int WN>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int li = lane & 15, q = lane >> 4;
    constexpr int MBLK = 16 * MT * WM;

    int t = blockIdx.x;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int TWl = p.TW_log2, TW = 1 << TWl, TH = MBLK >> TWl;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int s = p.stride, halo = p.ksize >> 1;
    const int PW = s * (TW - 1) + 1 + 2 * halo, PH = s * (TH - 1) + 1 + 2 * halo;
    const int iy0 = s * oy0 - halo, ix0 = s * ox0 - halo;
    const int ntaps = p.ksize * p.ksize;
    const int KB = p.Cin_pad >> 4;

    const int ntile0 = (blockIdx.y * WN + wn) * NTW;  // first n-tile of this wave
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane LDS base (in floats, without the chunk's channel stride applied) of each m-tile
    int mrow[MT], mcol[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = (wm * MT + i) * 16 + li;
        mrow[i] = (m >> TWl) * s;
        mcol[i] = (m & (TW - 1)) * s;
    }
    const int ntiles_total = p.Cout_pad >> 4;

    // chunk pipeline (async-STAGE split): the next channel chunk's global loads are issued into registers before this
    // chunk's MFMA loop and written to LDS after it, so HBM/L2 latency hides under the matrix work
    StageRegs<6> R;
    {
        const int k0 = min(p.KCH, p.Cin_pad);
        stage_issue(p, R, b, iy0, ix0, PH, PW, 0, k0, k0 + 8);
    }
    for (int c0 = 0; c0 < p.Cin_pad; c0 += p.KCH) {
        const int kch = min(p.KCH, p.Cin_pad - c0);
        const int CS = kch + 8;
        __syncthreads();
        stage_commit(p, R, lds, b, iy0, ix0, PH, PW, c0, kch, CS);
        __syncthreads();
        if (c0 + p.KCH < p.Cin_pad) {
            const int kn = min(p.KCH, p.Cin_pad - c0 - p.KCH);
            stage_issue(p, R, b, iy0, ix0, PH, PW, c0 + p.KCH, kn, kn + 8);
        }
        const int kbn = kch >> 4;
        const int CS4 = CS >> 2;
        const float4* lds4 = reinterpret_cast<const float4*>(lds);
        // flattened (tap, kb) loop with the next iteration's B fragments (L2-resident packed weights) prefetched
        // into registers while the current iteration's MFMAs issue
        const size_t tap_stride = (size_t)KB * p.Cout_pad * 16, kb_stride = (size_t)p.Cout_pad * 16;
        const float* wlane = p.wpk + ((size_t)(c0 >> 4) * p.Cout_pad + li) * 16 + 4 * q;
        float4 bnext[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int nt = ntile0 + j;
            bnext[j] = (nt < ntiles_total) ? *reinterpret_cast<const float4*>(wlane + (size_t)nt * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        int tap = 0, kb = 0, tyy = 0, txx = 0;
        const int niter = ntaps * kbn;
        for (int it = 0; it < niter; ++it) {
            float4 bf[NTW];
#pragma unroll
            for (int j = 0; j < NTW; ++j) bf[j] = bnext[j];
            float4 af[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = lds4[((mrow[i] + tyy) * PW + mcol[i] + txx) * CS4 + kb * 4 + q];  // float4 index: provably 16-byte aligned -> ds_read_b128
            // advance (tap, kb) and prefetch
            ++kb;
            if (kb == kbn) {
                kb = 0;
                ++tap;
                ++txx;
                if (txx == p.ksize) {
                    txx = 0;
                    ++tyy;
                }
            }
            if (it + 1 < niter) {
                const float* wn = wlane + tap * tap_stride + kb * kb_stride;
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const int nt = ntile0 + j;
                    if (nt < ntiles_total) bnext[j] = *reinterpret_cast<const float4*>(wn + (size_t)nt * 256);
                }
            }
            // k-step outermost: consecutive MFMAs hit different accumulators (a 16x16x4 f32 MFMA issues every 32 cycles
            // but a dependent one must wait 40), so no issue slot is lost to the accumulate chain
#define TMG_MFMA_STEP(E)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NTW; ++j)          \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].E, bf[j].E, acc[i][j], 0, 0, 0);
            TMG_MFMA_STEP(x)
            TMG_MFMA_STEP(y)
            TMG_MFMA_STEP(z)
            TMG_MFMA_STEP(w)
#undef TMG_MFMA_STEP
        }
    }

    // epilogue: C/D map of the 16x16 tile: col = lane & 15 (channel), row = (lane >> 4) * 4 + r (pixel)
    const float osc = out_scale_of(p.kappa);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int n = (ntile0 + j) * 16 + li;
        if (n >= p.Cout) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
        int nl = n;
        float* op_ = p.out[0].p;
        int ostride = p.out[0].stride, ooff = p.out[0].off;
        if (nl >= p.out[0].n) {
            nl -= p.out[0].n;
            op_ = p.out[1].p; ostride = p.out[1].stride; ooff = p.out[1].off;
            if (nl >= p.out[1].n) {
                nl -= p.out[1].n;
                op_ = p.out[2].p; ostride = p.out[2].stride; ooff = p.out[2].off;
            }
        }
        float* obase = op_ + ooff + nl;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = (wm * MT + i) * 16 + q * 4 + r;
                const int oy = oy0 + (m >> TWl), ox = ox0 + (m & (TW - 1));
                if (oy < p.Hout && ox < p.Wout) {
                    const size_t opx = ((size_t)b * p.Hout + oy) * p.Wout + ox;
                    float v = acc[i][j][r] + bv;
                    if (p.add.p) v += p.add.p[opx * p.add.stride + p.add.off + n];
                    v *= osc;
                    if (p.relu_out) v = fmaxf(v, 0.f);
                    float* dst = obase + opx * ostride;
                    if (p.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent, software-pipelined variant of the kernel above for stride-1 convolutions over float4-addressable
// segment lists (every conv of the flow levels).  512 threads = WM x WN = 8 waves; a block walks a strided share of
// the tiles and, per tile, the channel chunks; the unit of the pipeline is one (tile, chunk) stage:
//   round k: registers holding stage k+1 -> idle LDS buffer; global loads of stage k+2 -> registers (in flight for a
//   whole round); MFMA loop of stage k from the other buffer; one barrier.
// Staging is the lean scheme of conv_wgrad_kernel: a thread owns one channel quad and every (512/k4p)-th patch pixel,
// its patch coordinates are tile-invariant registers, a batch of loads has no control flow (lanes with nothing to read
// use the zero page), so neither load latency nor per-item index math sits on the MFMA path (VALU cycles add to MFMA
// cycles on a SIMD).  The first B fragments of the next stage are prefetched during the last iteration of this one.
// ---------------------------------------------------------------------------------------------
template <int MT, int NTW, int WM, int WN>
__global__ __launch_bounds__(512, 1) void conv_fwd_kernel(ConvP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 512, UP = 7;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int li = lane & 15, q = lane >> 4;
    constexpr int MBLK = 16 * MT * WM;
    const int TWl = p.TW_log2, TW = 1 << TWl, TH = MBLK >> TWl;
    const int halo = p.ksize >> 1, ntaps = p.ksize * p.ksize;
    const int PW = TW + 2 * halo, PH = TH + 2 * halo, PHPW = PH * PW;
    const int KB = p.Cin_pad >> 4, KCH = p.KCH, CS = KCH + 8, CS4 = CS >> 2;
    const int bufw = PHPW * CS;  // words per LDS buffer
    const int nchunks = p.nchunks;
    const int ntiles_total = p.Cout_pad >> 4;
    const int ntile0 = (blockIdx.y * WN + wn) * NTW;  // first n-tile of this wave
    const float* zero_page = g_tmg_zero_page;

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // per-lane operand bases: A as float4 index into a buffer, B as float offset into a (tap, kb) slice of wpk
    int abase[MT], mrc[MT];  // mrc: (row << 16) | col of this lane's pixel of m-tile i inside the tile (epilogue)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = (wm * MT + i) * 16 + li;
        abase[i] = ((m >> TWl) * PW + (m & (TW - 1))) * CS4 + q;
        mrc[i] = ((m >> TWl) << 16) | (m & (TW - 1));
    }
    // B fragments through a buffer descriptor (tmg_bload4): lane part boff0 (bytes), everything else scalar
    const unsigned boff0 = 4u * (unsigned)(li * 16 + 4 * q);  // + n-tile * 256 words (scalar); tiles past the end repeat the last (dropped in the epilogue)
#define TMG_FW_BOFF(J) ((unsigned)(min(ntile0 + (J), ntiles_total - 1) * 1024))
    const __amdgpu_buffer_rsrc_t wrs = tmg_make_rsrc(p.wpk, (unsigned)(4u * (unsigned)ntaps * (unsigned)KB * (unsigned)p.Cout_pad * 16u));
    const size_t tap_stride = (size_t)KB * p.Cout_pad * 16, kb_stride = (size_t)p.Cout_pad * 16;

    // ---- lean staging state ---------------------------------------------------------------------------------------
    const int k4l = KCH <= 16 ? 2 : (KCH <= 32 ? 3 : 4), k4p = 1 << k4l;  // float4 slots per pixel, padded to 2^n
    const int pc4 = tid & (k4p - 1), ppix0 = tid >> k4l, pstep = NT >> k4l;
    const unsigned mp = 0xFFFFFFFFu / (unsigned)PW + 1u;
    unsigned pyx[UP];  // (py << 16) | px of item u
#pragma unroll
    for (int u = 0; u < UP; ++u) {
        const int pix = min(ppix0 + u * pstep, PHPW - 1);
        const int py = (int)__umulhi((unsigned)pix, mp), px = pix - py * PW;
        pyx[u] = ((unsigned)py << 16) | (unsigned)px;
    }
    const unsigned pdst0 = 4u * (ppix0 * CS + 4 * pc4);
    float4 pv[UP];
    unsigned oobm = 0;

    const int G = gridDim.x;
    const int nmine = (int)blockIdx.x < p.ntiles ? (p.ntiles - (int)blockIdx.x + G - 1) / G : 0;
    const int nst = nmine * nchunks;
    // (tile, chunk) cursors of the stage being issued / committed / computed.  A tile index advances by G per step; its
    // (tile x, tile y, sample) coordinates are carried along instead of being re-derived with two integer divisions at
    // every use (each ~35 VALU instructions, and VALU cycles add to MFMA cycles).
    int ci = 0, cc = 0, cm = 0;
    const int tpi = p.tiles_x * p.tiles_y;
    const int Gb = G / tpi, Gy = (G - Gb * tpi) / p.tiles_x, Gx = G - Gb * tpi - Gy * p.tiles_x;
    int ti_b = (int)blockIdx.x / tpi, ti_y = ((int)blockIdx.x - ti_b * tpi) / p.tiles_x, ti_x = (int)blockIdx.x - ti_b * tpi - ti_y * p.tiles_x;
    int tm_b = ti_b, tm_y = ti_y, tm_x = ti_x;
#define TMG_FW_ADVANCE(X, Y, B_)                                  \
    {                                                             \
        X += Gx;                                                  \
        if (X >= p.tiles_x) { X -= p.tiles_x; ++Y; }              \
        Y += Gy;                                                  \
        if (Y >= p.tiles_y) { Y -= p.tiles_y; ++B_; }             \
        B_ += Gb;                                                 \
    }
    const float osc = out_scale_of(p.kappa);  // launch constant: read once, not in every tile's epilogue (a dependent load there)
    // narrow register tiles keep this lane's bias quads in registers for the whole launch (wide ones have none to spare)
    constexpr bool HB = MT * NTW <= 8;
    float4 biasq[HB ? NTW : 1];
    if (HB) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n0 = (ntile0 + j) * 16 + 4 * q;
            float b4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[r] = (p.bias && n0 + r < p.Cout) ? p.bias[n0 + r] : 0.f;
            biasq[HB ? j : 0] = make_float4(b4[0], b4[1], b4[2], b4[3]);
        }
    }
    // first B fragments of stage 0 (every later stage gets them from the previous stage's last iteration)
    float4 b0[NTW], b1[NTW];
    if (nst > 0) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) b0[j] = tmg_bload4(wrs, boff0, TMG_FW_BOFF(j));
    }

    for (int k = -2; k < nst; ++k) {
        // ---- commit stage k+1 (loaded during the previous round) into the idle buffer -------------------------------
        if (k >= -1 && k + 1 < nst) {
            const int c0 = cc * KCH, kch = min(KCH, p.Cin_pad - c0);
            const bool pcv = 4 * pc4 < kch;
            float4 isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.in_scale) {
                const int c = c0 + 4 * pc4;
                float* fs = reinterpret_cast<float*>(&isc);
                float* fh = reinterpret_cast<float*>(&ish);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < p.Cin) { fs[e] = p.in_scale[c + e]; fh[e] = p.in_shift[c + e]; }
            }
            char* bb = reinterpret_cast<char*>(lds) + 4u * (((k + 1) & 1) * bufw) + pdst0;
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                if (pcv && ppix0 + u * pstep < PHPW) {
                    float4 v = pv[u];
                    if (p.in_scale && !((oobm >> u) & 1u)) {
                        v.x = v.x * isc.x + ish.x; v.y = v.y * isc.y + ish.y;
                        v.z = v.z * isc.z + ish.z; v.w = v.w * isc.w + ish.w;
                    }
                    if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    *reinterpret_cast<float4*>(bb + u * pstep * CS * 4) = v;
                }
            }
            if (++cc == nchunks) cc = 0;
        }
        // ---- issue the loads of stage k+2 ---------------------------------------------------------------------------
        if (k + 2 < nst) {
            const int b_ = ti_b, oy0_ = ti_y * TH, ox0_ = ti_x * TW;
            const int c0 = ci * KCH, kch = min(KCH, p.Cin_pad - c0);
            const float* tptr = zero_page;
            int tss = 0;
            {
                int cl = c0 + 4 * pc4;
                if (4 * pc4 < kch && cl < p.Cin) {
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
            const int iy0 = oy0_ - halo, ix0 = ox0_ - halo;
            const int tbv = b_ * p.Hin * p.Win * tss;
            oobm = 0;
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const int iy = iy0 + (int)(pyx[u] >> 16), ix = ix0 + (int)(pyx[u] & 0xffffu);
                const int iyc = min(max(iy, 0), p.Hin - 1), ixc = min(max(ix, 0), p.Win - 1);
                const bool oob = !p.pad_rep && (iy != iyc || ix != ixc);
                const int elem = (int)__umul24(__umul24(iyc, p.Win) + ixc, tss) + tbv;
                const float* a_ = (oob || ppix0 + u * pstep >= PHPW) ? zero_page : tptr + elem;
                pv[u] = *reinterpret_cast<const float4*>(a_);
                if (p.in_scale) oobm |= (oob ? 1u : 0u) << u;
            }
            if (++ci == nchunks) { ci = 0; TMG_FW_ADVANCE(ti_x, ti_y, ti_b) }
        }
        // ---- MFMA loop of stage k -----------------------------------------------------------------------------------
        if (k >= 0) {
            const int c0 = cm * KCH, kch = min(KCH, p.Cin_pad - c0);
            const int kbn = kch >> 4, niter = ntaps * kbn;
            const float4* lds4 = reinterpret_cast<const float4*>(lds + (k & 1) * bufw);
            const float* wl = p.wpk + (size_t)(c0 >> 4) * kb_stride;
            // where the B prefetch of the last iteration points: first fragments of the next stage (or anything valid)
            const int c0n = (cm + 1 == nchunks) ? 0 : c0 + KCH;
            const float* wl_next = p.wpk + (size_t)(c0n >> 4) * kb_stride;
            int tap = 0, kb = 0, txx = 0;
            // Narrow register tiles: fetch the epilogue's `add` operand now, so that its latency hides under the MFMA loop
            // instead of stalling every tile's epilogue (the per-layer coupling convs are only a few microseconds per tile)
            // The MFMA operands are swapped (weights as A, pixels as B): the 16x16 result tile is transposed, lane (li, q)
            // holds channels 4q..4q+3 of pixel li, so the epilogue moves float4s along the channel axis (4x fewer store /
            // add-load instructions and address computations than one float per lane and row).
            // Narrow register tiles: fetch the epilogue's `add` operand now, so that its latency hides under the MFMA loop
            // instead of stalling every tile's epilogue (the per-layer coupling convs are only a few microseconds per tile)
            constexpr bool PREADD = MT * NTW <= 8;
            float4 addv[PREADD ? MT : 1][PREADD ? NTW : 1];
            if (PREADD && p.add.p && p.ovec4 && cm + 1 == nchunks) {
                const int b_ = tm_b, oy0_ = tm_y * TH, ox0_ = tm_x * TW;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int oy = min(oy0_ + (mrc[i] >> 16), p.Hout - 1), ox = min(ox0_ + (mrc[i] & 0xffff), p.Wout - 1);
                    const unsigned opx = (unsigned)(__umul24(b_ * p.Hout + oy, p.Wout) + ox);     // u32 x u32 -> u64 products below: one v_mad_u64_u32
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        const int n0 = min((ntile0 + j) * 16 + 4 * q, p.Cout - 4);
                        addv[PREADD ? i : 0][PREADD ? j : 0] = *reinterpret_cast<const float4*>(p.add.p + (size_t)opx * (unsigned)p.add.stride + p.add.off + n0);
                    }
                }
            }
            // (tap, kb) bookkeeping is carried as two scalars advanced by constants - the A offset into the patch and the B slice
            // pointer - instead of being recomputed from (tap, kb) with 64-bit multiplies in every iteration (the narrow
            // register tiles issue only 16 MFMAs per iteration, so ~25 scalar instructions of index math per iteration showed)
            const size_t wtap_adv = tap_stride - (size_t)kbn * kb_stride;  // first slice of the next tap from the last of this
            const int a_tap_adv = CS4 - 4 * kbn, a_row_adv = (PW - p.ksize) * CS4;
            const float* wcur = wl;  // slice of the iteration whose B fragments are prefetched next
            int aoffs = 0;
#define TMG_FW_BODY(BC, BN)                                                                                          \
            {                                                                                                        \
                float4 af[MT];                                                                                       \
                _Pragma("unroll") for (int i = 0; i < MT; ++i) af[i] = lds4[abase[i] + aoffs];                       \
                ++kb; aoffs += 4; wcur += kb_stride;                                                                 \
                if (kb == kbn) {                                                                                     \
                    kb = 0; ++tap; ++txx; aoffs += a_tap_adv; wcur += wtap_adv;                                      \
                    if (txx == p.ksize) { txx = 0; aoffs += a_row_adv; }                                             \
                }                                                                                                    \
                const float* wn_ = (tap == ntaps) ? wl_next : wcur;                                                  \
                const unsigned wso_ = (unsigned)((wn_ - p.wpk) * 4);                                               \
                _Pragma("unroll") for (int j = 0; j < NTW; ++j) BN[j] = tmg_bload4(wrs, boff0, wso_ + TMG_FW_BOFF(j)); \
                _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NTW; ++j)       \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(BC[j].x, af[i].x, acc[i][j], 0, 0, 0);          \
                _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NTW; ++j)       \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(BC[j].y, af[i].y, acc[i][j], 0, 0, 0);          \
                _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NTW; ++j)       \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(BC[j].z, af[i].z, acc[i][j], 0, 0, 0);          \
                _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NTW; ++j)       \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(BC[j].w, af[i].w, acc[i][j], 0, 0, 0);          \
            }
            int it = 0;
            for (; it + 1 < niter; it += 2) {
                TMG_FW_BODY(b0, b1)
                TMG_FW_BODY(b1, b0)
            }
            if (it < niter) {
                TMG_FW_BODY(b0, b1)
#pragma unroll
                for (int j = 0; j < NTW; ++j) b0[j] = b1[j];
            }
#undef TMG_FW_BODY
            if (cm + 1 == nchunks) {
                // epilogue: transposed C/D map: col = lane & 15 (pixel of the m-tile), row = (lane >> 4) * 4 + r (channel)
                const int b_ = tm_b, oy0_ = tm_y * TH, ox0_ = tm_x * TW;
                // (opaque copy per epilogue: the channel offsets n0 / nl below are tile-invariant; hoisted out of the stage loop they are
                // one register per n-tile that the widest instance <4,4,*> - 256 registers - spilled: 8 bytes of scratch per lane)
                int qe = q;
                asm volatile("" : "+v"(qe));
                // STORE-ONLY path (round 6): no load may sit in a conditional block between the stores - the compiler answers such a
                // load (`add` fetched here, the old output of `accumulate`, a bias quad) with `s_waitcnt vmcnt(0)` in front of EVERY
                // store, taken or not: each store of the tile then waits for the one before it and for the B fragments / patch loads
                // prefetched for the next stage (tools/isa_audit.py: 39 of 40 stores of <2,4,*>).  Everything this path needs is
                // in registers: the bias quads (HB), the `add` operand fetched before the MFMA loop (PREADD).
                const bool store_only = p.ovec4 && !p.accumulate && (HB || !p.bias) && (PREADD || !p.add.p);
                if (store_only) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int oy = oy0_ + (mrc[i] >> 16), ox = ox0_ + (mrc[i] & 0xffff);
                        if (oy < p.Hout && ox < p.Wout) {
                            const unsigned opx = (unsigned)(__umul24(b_ * p.Hout + oy, p.Wout) + ox);
#pragma unroll
                            for (int j = 0; j < NTW; ++j) {
                                const int n0 = (ntile0 + j) * 16 + 4 * qe;
                                if (n0 < p.Cout) {
                                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                                    if (HB) { v[0] += biasq[HB ? j : 0].x; v[1] += biasq[HB ? j : 0].y; v[2] += biasq[HB ? j : 0].z; v[3] += biasq[HB ? j : 0].w; }
                                    if (PREADD && p.add.p) {
                                        const float4 a4 = addv[PREADD ? i : 0][PREADD ? j : 0];
                                        v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w;
                                    }
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        v[r] *= osc;
                                        if (p.relu_out) v[r] = fmaxf(v[r], 0.f);
                                    }
                                    int nl = n0;
                                    TMG_PICK_OSEG(p.out, nl, op_, ostride, ooff)
                                    *reinterpret_cast<float4*>(op_ + (size_t)opx * (unsigned)ostride + ooff + nl) = make_float4(v[0], v[1], v[2], v[3]);
                                }
                            }
                        }
                    }
                } else
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int oy = oy0_ + (mrc[i] >> 16), ox = ox0_ + (mrc[i] & 0xffff);
                    if (oy < p.Hout && ox < p.Wout) {
                        const unsigned opx = (unsigned)(__umul24(b_ * p.Hout + oy, p.Wout) + ox);  // host: B*H < 2^24, W < 2^24
#pragma unroll
                        for (int j = 0; j < NTW; ++j) {
                            const int n0 = (ntile0 + j) * 16 + 4 * qe;
                            if (n0 < p.Cout) {
                                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                                if (HB) {
                                    v[0] += biasq[HB ? j : 0].x; v[1] += biasq[HB ? j : 0].y; v[2] += biasq[HB ? j : 0].z; v[3] += biasq[HB ? j : 0].w;
                                } else if (p.bias) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) v[r] += (n0 + r < p.Cout) ? p.bias[n0 + r] : 0.f;
                                }
                                if (p.ovec4) {
                                    int nl = n0;
                                    TMG_PICK_OSEG(p.out, nl, op_, ostride, ooff)
                                    float4* dst = reinterpret_cast<float4*>(op_ + (size_t)opx * (unsigned)ostride + ooff + nl);
                                    if (p.add.p) {
                                        const float4 a4 = (PREADD) ? addv[PREADD ? i : 0][PREADD ? j : 0]
                                                                   : *reinterpret_cast<const float4*>(p.add.p + (size_t)opx * (unsigned)p.add.stride + p.add.off + n0);
                                        v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w;
                                    }
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        v[r] *= osc;
                                        if (p.relu_out) v[r] = fmaxf(v[r], 0.f);
                                    }
                                    if (p.accumulate) {
                                        const float4 o4 = *dst;
                                        v[0] += o4.x; v[1] += o4.y; v[2] += o4.z; v[3] += o4.w;
                                    }
                                    *dst = make_float4(v[0], v[1], v[2], v[3]);
                                } else {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const int n = n0 + r;
                                        if (n < p.Cout) {
                                            int nl = n;
                                            TMG_PICK_OSEG(p.out, nl, op_, ostride, ooff)
                                            float x = v[r];
                                            if (p.add.p) x += p.add.p[(size_t)opx * (unsigned)p.add.stride + p.add.off + n];
                                            x *= osc;
                                            if (p.relu_out) x = fmaxf(x, 0.f);
                                            float* dst = op_ + (size_t)opx * (unsigned)ostride + ooff + nl;
                                            if (p.accumulate) x += *dst;
                                            *dst = x;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                cm = 0; TMG_FW_ADVANCE(tm_x, tm_y, tm_b)
            } else {
                ++cm;
            }
        }
        __syncthreads();  // the buffer just read may be overwritten next round; the one just written is complete
    }
#undef TMG_FW_ADVANCE
#undef TMG_FW_BOFF
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[co][ci][tap] (+)= scale * sum_p in(p*s + tap)[ci] * dy(p)[co]
// GEMM view: M = input channels (per tap), N = output channels, K = pixels.
// ---------------------------------------------------------------------------------------------
struct WgradP {
    TmgSeg in[TMG_MAX_IN_SEG];
    int nseg;
    int vec4;
    int B, Hin, Win, Hout, Wout;
    int ksize, stride;
    int Cin, Cin_pad, Cout;
    const float* in_scale;
    const float* in_shift;
    int relu_in, pad_rep;
    TmgSeg dy;  // n == Cout
    int dy_vec4;
    float* dW;     // [Cout][Cin][ksize*ksize], accumulated with atomics (caller zeroes)
    float* dbias;  // [Cout] or null, accumulated with atomics
    // destination layout of dW: row length cin_dst per output channel; source channel ci < cin_valid lands at
    // ci + (ci < ci_split ? ci_off0 : ci_off1); channels >= cin_valid are dropped.  Lets one launch write a channel
    // sub-range of a wider native weight-gradient tensor (or skip padding channels).
    int cin_dst, cin_valid, ci_split, ci_off0, ci_off1;
    float* ws;     // optional partial-sum slabs (see conv_wgrad_reduce_kernel); null -> direct atomics
    const float* kappa;
    int TW_log2, TH;  // pixel tile (TH*TW == MPIX)
    int MPIX;
    int tiles_x, tiles_y, ntiles;
    int CITG;  // input-channel tiles (of 16) a block group stages at most (LDS sizing)
    int PPG;   // (input-channel tile, tap) pairs per block group: grid.z walks consecutive ranges of the tile-major pair list
    int dbg;     // timing experiments (TMG_WG_DBG): 1 = no MFMA loop, 2 = no staging
    int fstage;  // 1: every segment / dy is float4-addressable and offsets fit 24-bit multiplies -> lean staging path
    // grouped launch (tmg_conv_wgrad_grouped): group g = blockIdx.y / bpg reads its own input segments from gtab[g],
    // dy channels [g*dy_goff, +Cout) and writes dW + g*dw_gstride / dbias + g*db_gstride; null: one group
    const long long* gtab;  // device: [ngroups][4][4]: 3 input segments {pointer, pixel stride, channel offset, channels} + {dy pointer (0: shared dy), dy pixel stride, 0, 0}
    int bpg, dy_goff;
    long long dw_gstride;
    int db_gstride;
    int ksplit;  // 1: waves split the pixels of a tile instead of the (tap, channel tile) pairs (see the kernel)
};

// A block (512 threads = 8 waves, two per SIMD) owns (CITG input-channel tiles x all taps) x (NCO output-channel
// tiles) of dW and a strided share of the pixel tiles.  Work split inside the block (p.ksplit, wave-uniform):
//   0: the (tap, channel tile) pairs are dealt round-robin to 4 waves, and the two wave quartets each walk one half of
//      a tile's pixels;
//   1: (<= 9 pairs) every wave owns ALL pairs and walks an eighth of the tile's pixels - keeps narrow layers at
//      9*NCO MFMAs per k-step per wave instead of 2-3, perfectly balanced.
// Partial sums of the waves meet in the reduce kernel (or in the atomics of the direct path).
// LDS layout (channel-tile major): patch [citn][PH*PW][16], dy [NCO][MPIX][16].  A pixel is 16 words apart in every
// configuration, so a fragment read (16 channels x 4 consecutive pixels) covers 64 consecutive words: conflict-free,
// and the four k-steps of a 16-pixel unit are reached with immediates from one address register per pair.
// Pipeline: LDS is double-buffered and tiles are prefetched two rounds ahead - round k stores the registers holding
// tile k+1 into the idle buffer, issues the global loads of tile k+2 into those registers, runs the MFMA loop of tile
// k from the other buffer, and ends with the only barrier of the round.  No load latency is exposed.
template <int NP, int NCO, bool LEAN>
__global__ __launch_bounds__(512, 1) void conv_wgrad_kernel(WgradP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef const __attribute__((address_space(3))) float* lds_cptr;
    constexpr int NT = 512;
    constexpr int UP = 7;    // patch float4 per thread (plan_wgrad keeps PH*PW*k4p <= UP*NT; 3x3 on a 4x32 tile needs 6.4)
    constexpr int UD = NCO == 1 ? 4 : NCO;  // dy float4 per thread: covers MPIX * NCO / 128 (tiles of up to 512 pixels when NCO == 1, else 128)
    const int MPIX = p.MPIX;  // pixels per staged tile (64 or 128)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> scalar branches
    const int w4 = wave & 3;
    const int li = lane & 15, q = lane >> 4;
    const int TWl = p.TW_log2, TW = 1 << TWl, TH = MPIX >> TWl;
    const int s = p.stride, halo = p.ksize >> 1, ntaps = p.ksize * p.ksize;
    const int PW = s * (TW - 1) + 1 + 2 * halo, PH = s * (TH - 1) + 1 + 2 * halo;
    // channel-tile planes are padded by one 64-byte row: consecutive planes then start 64 bytes apart modulo the 256-byte
    // bank row, so the float4 stores of one pixel's quads (which go to different planes) do not collide (27 % of the LDS
    // cycles were bank conflicts with unpadded planes, all from the staging stores)
    const int PHPW = PH * PW, plane = PHPW * 16 + 16, dplane = MPIX * 16 + 16;
    // this group's pairs: [pair0, pair0 + npairs) of the list ordered (channel tile, tap); they touch channel tiles
    // cit0 .. cit0 + citn - 1.  Ranges of equal length (not whole channel tiles) keep every group - hence every CU - equally
    // loaded: 7 channel tiles x 9 taps = 63 pairs split 32 + 31 instead of 36 + 27.
    const int pair0 = (int)blockIdx.z * p.PPG;
    const int npairs = min(p.PPG, ntaps * (p.Cin_pad >> 4) - pair0);
    const int cit0 = pair0 / ntaps;
    const int citn = (pair0 + npairs - 1) / ntaps - cit0 + 1;
    const int k4 = citn * 4;
    const int grp = p.gtab ? (int)blockIdx.y / p.bpg : 0;  // grouped launch: which (input segments, dy slice, dW slice)
    const int co0 = ((int)blockIdx.y - grp * p.bpg) * NCO * 16;
    const int ldy_w = citn * plane;           // word offset of the dy tile inside a buffer
    const int bufw = p.CITG * plane + NCO * dplane;  // words per buffer
    const int ksplit = p.ksplit;

    f32x4 acc[NP][NCO];
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int n = 0; n < NCO; ++n) acc[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // LDS word offset of each owned pair (+ this lane's channel), hoisted out of the pixel loop.  Slots past the last
    // pair repeat it (their sums are dropped later): branch-free k-steps keep the read/MFMA schedule intact, and a
    // block is as slow as its busiest wave anyway.
    int aoffw[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int gp = pair0 + min(ksplit ? j : w4 + 4 * j, npairs - 1);
        const int citg = gp / ntaps, tap = gp - citg * ntaps, cit = citg - cit0;
        const int tyy = tap / p.ksize, txx = tap - tyy * p.ksize;
        aoffw[j] = cit * plane + (tyy * PW + txx) * 16 + li;
    }

    // ---- register-staged tile loads -----------------------------------------------------------------------------
    const int pitems = PHPW * k4, ditems = MPIX * NCO * 4;
    // floor(n / d) == umulhi(n, ceil(2^32 / d)) for n * d < 2^32 (d >= 2)
    const unsigned mk = 0xFFFFFFFFu / (unsigned)k4 + 1u;
    const unsigned mp = 0xFFFFFFFFu / (unsigned)PW + 1u;
    float4 pv[UP], dv[UD];
    float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f);  // dbias partial of this thread's 4 output channels
#define TMG_WG_ORIGIN(TILE)                  \
    int b_, oy0_, ox0_;                      \
    {                                        \
        int t_ = (TILE);                     \
        const int tx_ = t_ % p.tiles_x;      \
        t_ /= p.tiles_x;                     \
        const int ty_ = t_ % p.tiles_y;      \
        b_ = t_ / p.tiles_y;                 \
        oy0_ = ty_ * TH;                     \
        ox0_ = tx_ * TW;                     \
    }
#define TMG_WG_PATCH_ITEM(IT)                                                                 \
    const int pix_ = (int)__umulhi((unsigned)(IT), mk), c4_ = (IT) - pix_ * k4;               \
    const int py_ = (int)__umulhi((unsigned)pix_, mp), px_ = pix_ - py_ * PW;
    // global loads of one tile into registers (no LDS access)
#define TMG_WG_ISSUE(TILE, TID)                                                                                   \
    {                                                                                                             \
        TMG_WG_ORIGIN(TILE)                                                                                       \
        _Pragma("unroll") for (int u = 0; u < UP; ++u) {                                                          \
            const int it = (TID) + u * NT;                                                                        \
            if (it < pitems) {                                                                                    \
                TMG_WG_PATCH_ITEM(it)                                                                             \
                pv[u] = load_in4(p, b_, s * oy0_ - halo + py_, s * ox0_ - halo + px_, cit0 * 16 + 4 * c4_);       \
            }                                                                                                     \
        }                                                                                                         \
        _Pragma("unroll") for (int u = 0; u < UD; ++u) {                                                          \
            const int it = (TID) + u * NT;                                                                        \
            if (it < ditems) {                                                                                    \
                const int m = it / (NCO * 4), c4 = it - m * (NCO * 4);                                            \
                const int oy = oy0_ + (m >> TWl), ox = ox0_ + (m & (TW - 1));                                     \
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
                if (oy < p.Hout && ox < p.Wout) {                                                                 \
                    const float* src = p.dy.p + (((size_t)b_ * p.Hout + oy) * p.Wout + ox) * p.dy.stride + p.dy.off; \
                    const int c = co0 + 4 * c4;                                                                   \
                    if (p.dy_vec4 && c + 3 < p.Cout) {                                                            \
                        v = *reinterpret_cast<const float4*>(src + c);                                            \
                    } else {                                                                                      \
                        if (c < p.Cout) v.x = src[c];                                                             \
                        if (c + 1 < p.Cout) v.y = src[c + 1];                                                     \
                        if (c + 2 < p.Cout) v.z = src[c + 2];                                                     \
                        if (c + 3 < p.Cout) v.w = src[c + 3];                                                     \
                    }                                                                                             \
                }                                                                                                 \
                dv[u] = v;                                                                                        \
            }                                                                                                     \
        }                                                                                                         \
    }
    // registers -> LDS buffer starting at word BW
#define TMG_WG_COMMIT(BW, TID)                                                                                    \
    {                                                                                                             \
        _Pragma("unroll") for (int u = 0; u < UP; ++u) {                                                          \
            const int it = (TID) + u * NT;                                                                        \
            if (it < pitems) {                                                                                    \
                TMG_WG_PATCH_ITEM(it)                                                                             \
                (void)py_; (void)px_;                                                                             \
                *reinterpret_cast<float4*>(lds + (BW) + (c4_ >> 2) * plane + pix_ * 16 + (c4_ & 3) * 4) = pv[u]; \
            }                                                                                                     \
        }                                                                                                         \
        _Pragma("unroll") for (int u = 0; u < UD; ++u) {                                                          \
            const int it = (TID) + u * NT;                                                                        \
            if (it < ditems) {                                                                                    \
                const int m = it / (NCO * 4), c4 = it - m * (NCO * 4);                                            \
                *reinterpret_cast<float4*>(lds + (BW) + ldy_w + (c4 >> 2) * dplane + m * 16 + (c4 & 3) * 4) = dv[u]; \
                /* NT % (NCO*4) == 0: c4 is the same for all of a thread's items -> per-thread bias partial */    \
                bacc.x += dv[u].x; bacc.y += dv[u].y; bacc.z += dv[u].z; bacc.w += dv[u].w;                       \
            }                                                                                                     \
        }                                                                                                         \
    }

#define TMG_WG_MFMA(AV, BF)                                                                             \
    _Pragma("unroll") for (int j = 0; j < NP; ++j) {                                                    \
        _Pragma("unroll") for (int n = 0; n < NCO; ++n)                                                 \
            acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[j], BF[n], acc[j][n], 0, 0, 0);         \
    }
#define TMG_SB __builtin_amdgcn_sched_barrier(0);

    // ---- lean staging (p.fstage): VALU cycles add to MFMA cycles on a SIMD, so the per-item address math is kept to
    // ~a dozen full-rate instructions.  A thread owns one channel quad (its segment is resolved once) and every
    // (NT / k4p)-th patch pixel; the patch coordinates of its items are tile-invariant and live in registers.
    // (up to 8 channel tiles per group: the 1x1 contractions of the 128-channel level stage 8; plan_wgrad's fits() mirrors this)
    const int k4l = citn <= 1 ? 2 : (citn <= 2 ? 3 : (citn <= 4 ? 4 : 5)), k4p = 1 << k4l;  // float4 slots per pixel, padded to 2^n
    const int pc4 = tid & (k4p - 1), ppix0 = tid >> k4l, pstep = NT >> k4l;
    const float* tptr = g_tmg_zero_page;  // this thread's segment base (+ channel); padding channels read zeros
    int tss = 0;
    {
        int cl = cit0 * 16 + 4 * pc4;
        if (cl < p.Cin) {
            // segment descriptors: the launch's own, or this group's row of the device table (scalar loads)
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
            tptr = sp + so + cl;
            tss = ss;
        }
    }
    const bool pcv = pc4 < k4;  // slots past the group's channel tiles do not exist in LDS
    unsigned pyx[UP];           // (py << 16) | px of item u
#pragma unroll
    for (int u = 0; u < UP; ++u) {
        const int pix = min(ppix0 + u * pstep, PHPW - 1);
        const int py = (int)__umulhi((unsigned)pix, mp), px = pix - py * PW;
        pyx[u] = ((unsigned)py << 16) | (unsigned)px;
    }
    const unsigned pdst0 = 4u * ((pc4 >> 2) * plane + ppix0 * 16 + (pc4 & 3) * 4);  // LDS byte offset of item 0
    // dy: thread owns channel quad dc4 of every (NT / (NCO*4))-th pixel
    constexpr int DL = (NCO == 1) ? 2 : (NCO == 2 ? 3 : 4);
    const int dc4 = tid & (NCO * 4 - 1), dm0 = tid >> DL;
    constexpr int dstep = NT >> DL;
    const bool dcv = co0 + 4 * dc4 < p.Cout;
    // dy of this group: a channel slice of the shared tensor, or the group's own tensor (table row 3)
    const float* dyb = p.dy.p + p.dy.off + grp * p.dy_goff;
    int dys = p.dy.stride;
    if (p.gtab && p.gtab[(size_t)grp * 16 + 12]) {
        dyb = reinterpret_cast<const float*>(p.gtab[(size_t)grp * 16 + 12]);
        dys = (int)p.gtab[(size_t)grp * 16 + 13];
    }
    const float* dptr = dcv ? dyb + co0 + 4 * dc4 : g_tmg_zero_page;
    const int dss = dcv ? dys : 0;
    const unsigned ddst0 = 4u * (ldy_w + (dc4 >> 2) * dplane + dm0 * 16 + (dc4 & 3) * 4);
    unsigned oobm = 0;  // per-item out-of-image bits (only maintained when an input affine must not touch padding)
    float4 isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = make_float4(0.f, 0.f, 0.f, 0.f);
    if (LEAN && p.in_scale) {
        const int c = cit0 * 16 + 4 * pc4;
        float* fs = reinterpret_cast<float*>(&isc);
        float* fh = reinterpret_cast<float*>(&ish);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < p.Cin) { fs[e] = p.in_scale[c + e]; fh[e] = p.in_shift[c + e]; }
    }
#define TMG_WG_ISSUE_F(TILE)                                                                                       \
    {                                                                                                             \
        const int b_ = wi_b, oy0_ = wi_y * TH, ox0_ = wi_x * TW;  /* carried coordinates of tile k+2 */           \
        const int iy0_ = s * oy0_ - halo, ix0_ = s * ox0_ - halo;                                                 \
        const int tbv_ = b_ * p.Hin * p.Win * tss;          /* element offset of image b in this thread's segment */ \
        if (p.in_scale) oobm = 0;                                                                                 \
        /* branch-free: a load inside a divergent block makes the compiler drain vmcnt before the next one, which   \
           serialises the whole batch; lanes without an item read the zero page instead */                         \
        _Pragma("unroll") for (int u = 0; u < UP; ++u) {                                                          \
            const int iy = iy0_ + (int)(pyx[u] >> 16), ix = ix0_ + (int)(pyx[u] & 0xffffu);                       \
            const int iyc = min(max(iy, 0), p.Hin - 1), ixc = min(max(ix, 0), p.Win - 1);                         \
            const bool oob = !p.pad_rep && (iy != iyc || ix != ixc);                                              \
            const int elem = (int)__umul24(__umul24(iyc, p.Win) + ixc, tss) + tbv_;                               \
            const float* a_ = (oob || !pcv || ppix0 + u * pstep >= PHPW) ? g_tmg_zero_page : tptr + elem;             \
            pv[u] = tmg_ldg4(a_);   /* (table-derived pointers: see tmg_ldg4) */                                  \
            if (p.in_scale) oobm |= (oob ? 1u : 0u) << u;                                                         \
        }                                                                                                         \
        const int tbd_ = b_ * p.Hout * p.Wout * dss;                                                              \
        _Pragma("unroll") for (int u = 0; u < UD; ++u) {                                                          \
            const int m = dm0 + u * dstep;                                                                        \
            const int oy = oy0_ + (m >> TWl), ox = ox0_ + (m & (TW - 1));                                         \
            const bool inb = oy < p.Hout && ox < p.Wout && m < MPIX;                                              \
            const int elem = (int)__umul24(__umul24(oy, p.Wout) + ox, dss) + tbd_;                                \
            const float* a_ = inb ? dptr + elem : g_tmg_zero_page;                                                    \
            dv[u] = tmg_ldg4(a_);                                                                                 \
        }                                                                                                         \
    }
#define TMG_WG_COMMIT_F(BW)                                                                                       \
    {                                                                                                             \
        _Pragma("unroll") for (int u = 0; u < UP; ++u) {                                                          \
            if (pcv && ppix0 + u * pstep < PHPW) {                                                                \
                float4 v = pv[u];                                                                                 \
                if (p.in_scale && !((oobm >> u) & 1u)) {                                                          \
                    v.x = v.x * isc.x + ish.x; v.y = v.y * isc.y + ish.y;                                         \
                    v.z = v.z * isc.z + ish.z; v.w = v.w * isc.w + ish.w;                                         \
                }                                                                                                 \
                if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); } \
                *reinterpret_cast<float4*>(reinterpret_cast<char*>(lds) + 4u * (BW) + pdst0 + u * pstep * 64) = v; \
            }                                                                                                     \
        }                                                                                                         \
        _Pragma("unroll") for (int u = 0; u < UD; ++u) {                                                          \
            if (dm0 + u * dstep < MPIX) {                                                                         \
                *reinterpret_cast<float4*>(reinterpret_cast<char*>(lds) + 4u * (BW) + ddst0 + u * dstep * 64) = dv[u]; \
                bacc.x += dv[u].x; bacc.y += dv[u].y; bacc.z += dv[u].z; bacc.w += dv[u].w;                       \
            }                                                                                                     \
        }                                                                                                         \
    }

    // pixels of a tile walked by this wave: [px0, px0 + npx)
    const int npx = ksplit ? MPIX >> 3 : MPIX >> 1;
    const int px0 = (ksplit ? wave : (wave >> 2)) * npx;
    const bool fast = (s == 1) && (TWl >= 4) && !(npx & 15);  // whole 16-pixel units, each inside one tile row
    const int G = gridDim.x;
    // Rotated tile loop (one ISSUE / COMMIT site): round k works on tile = blockIdx.x + k*G; rounds -2 and -1 only fill
    // the pipeline.
    // coordinates of the tile whose loads are issued next (tile k+2), carried instead of two integer divisions per tile
    const int tpi = p.tiles_x * p.tiles_y;
    const int Gb = G / tpi, Gy = (G - Gb * tpi) / p.tiles_x, Gx = G - Gb * tpi - Gy * p.tiles_x;
    int wi_b = (int)blockIdx.x / tpi, wi_y = ((int)blockIdx.x - wi_b * tpi) / p.tiles_x, wi_x = (int)blockIdx.x - wi_b * tpi - wi_y * p.tiles_x;
    int k = -2;
    for (int tile = (int)blockIdx.x - 2 * G; tile < p.ntiles; tile += G, ++k) {
        // The item -> (pixel, channel) index math of the staging macros is tile-invariant; left alone the compiler hoists
        // all of it out of this loop and spills.  Laundering the thread id makes it recompute (~200 VALU per tile).
        int tid_c = tid, tid_i = tid;
        asm volatile("" : "+v"(tid_c));
        asm volatile("" : "+v"(tid_i));
        // Staging of a round: tile k+1 (loaded a round ago) goes from registers to the idle LDS buffer, then the loads of tile
        // k+2 are issued into those registers.  Neither touches the buffer the MFMA loop reads, so a wave may do it anywhere
        // inside its round.
#define TMG_WG_STAGE_ROUND                                                                                        \
        if constexpr (LEAN) {                                                                                     \
            if (k >= -1 && tile + G < p.ntiles && p.dbg != 2) TMG_WG_COMMIT_F(((k + 1) & 1) * bufw)              \
            if (tile + 2 * G < p.ntiles && p.dbg != 2) {                                                          \
                TMG_WG_ISSUE_F(tile + 2 * G)                                                                      \
                wi_x += Gx;                                                                                       \
                if (wi_x >= p.tiles_x) { wi_x -= p.tiles_x; ++wi_y; }                                             \
                wi_y += Gy;                                                                                       \
                if (wi_y >= p.tiles_y) { wi_y -= p.tiles_y; ++wi_b; }                                             \
                wi_b += Gb;                                                                                       \
            }                                                                                                     \
        } else {                                                                                                  \
            if (k >= -1 && tile + G < p.ntiles) TMG_WG_COMMIT(((k + 1) & 1) * bufw, tid_c)                        \
            if (tile + 2 * G < p.ntiles) TMG_WG_ISSUE(tile + 2 * G, tid_i)                                        \
        }
#define TMG_WG_LDF(AV, BF, K)                                                                                     \
        {                                                                                                         \
            _Pragma("unroll") for (int n = 0; n < NCO; ++n) BF[n] = *(lds_cptr)(uintptr_t)(ba[n] + (K) * 256);    \
            _Pragma("unroll") for (int j = 0; j < NP; ++j) AV[j] = *(lds_cptr)(uintptr_t)(aa[j] + (K) * 256);     \
        }
#define TMG_WG_LDG(AV, BF, KS)                                                                                    \
        {                                                                                                         \
            const int m_ = (KS) * 4 + q;                                                                          \
            const float* ab_ = lds + cbw + (((m_ >> TWl) * s) * PW + (m_ & (TW - 1)) * s) * 16;                   \
            _Pragma("unroll") for (int n = 0; n < NCO; ++n) BF[n] = lds[cbw + ldy_w + n * dplane + m_ * 16 + li]; \
            _Pragma("unroll") for (int j = 0; j < NP; ++j) AV[j] = ab_[aoffw[j]];                                 \
        }
        // MFMA loop over the 16-pixel units [UA, UB) of this wave's share of tile k (fast path: fragment registers ping-pong
        // between k-steps, LDS byte addresses are bumped in place once per unit, the scheduling barriers keep "issue the next
        // k-step's reads, then this k-step's MFMAs") or over the k-steps [KA, KB) (generic walk for narrow tiles / stride 2:
        // per-k-step addresses, reads one k-step ahead; KB - KA is even)
#define TMG_WG_RUN(UA, UB, KA, KB)                                                                                \
        if (k >= 0 && p.dbg != 1) {                                                                               \
            const int cbw = (k & 1) * bufw;                                                                       \
            float av[NP], bfr[NCO], avn[NP], bfn[NCO];                                                            \
            if (fast) {                                                                                           \
                const int ua_ = (UA), ub_ = (UB), pxa_ = ua_ * 16;                                                \
                unsigned aa[NP], ba[NCO];                                                                         \
                _Pragma("unroll") for (int j = 0; j < NP; ++j)                                                    \
                    aa[j] = 4u * (cbw + aoffw[j] + q * 16 + ((pxa_ >> TWl) * PW + (pxa_ & (TW - 1))) * 16);       \
                _Pragma("unroll") for (int n = 0; n < NCO; ++n)                                                   \
                    ba[n] = 4u * (cbw + ldy_w + n * dplane + (pxa_ + q) * 16 + li);                               \
                TMG_WG_LDF(av, bfr, 0) TMG_SB                                                                     \
                for (int u = ua_; u < ub_; ++u) {                                                                 \
                    TMG_WG_LDF(avn, bfn, 1) TMG_SB                                                                \
                    TMG_WG_MFMA(av, bfr) TMG_SB                                                                   \
                    TMG_WG_LDF(av, bfr, 2) TMG_SB                                                                 \
                    TMG_WG_MFMA(avn, bfn) TMG_SB                                                                  \
                    TMG_WG_LDF(avn, bfn, 3) TMG_SB                                                                \
                    TMG_WG_MFMA(av, bfr) TMG_SB                                                                   \
                    {                                                                                             \
                        const int un = min(u + 1, ub_ - 1); /* the last unit re-reads its own first k-step */     \
                        const int pa = u * 16, pb = un * 16;                                                      \
                        const int d = (((pb >> TWl) * PW + (pb & (TW - 1))) - ((pa >> TWl) * PW + (pa & (TW - 1)))) * 64; \
                        const int db = (pb - pa) * 64;                                                            \
                        _Pragma("unroll") for (int j = 0; j < NP; ++j) aa[j] += d;                                \
                        _Pragma("unroll") for (int n = 0; n < NCO; ++n) ba[n] += db;                              \
                    }                                                                                             \
                    TMG_WG_LDF(av, bfr, 0) TMG_SB                                                                 \
                    TMG_WG_MFMA(avn, bfn) TMG_SB                                                                  \
                }                                                                                                 \
            } else {                                                                                              \
                const int ka_ = (KA), kb_ = (KB);                                                                 \
                if (ka_ < kb_) {                                                                                  \
                    TMG_WG_LDG(av, bfr, ka_)                                                                      \
                    for (int ks = ka_; ks < kb_; ks += 2) {                                                       \
                        TMG_WG_LDG(avn, bfn, ks + 1)                                                              \
                        TMG_WG_MFMA(av, bfr)                                                                      \
                        TMG_WG_LDG(av, bfr, min(ks + 2, kb_ - 1))                                                 \
                        TMG_WG_MFMA(avn, bfn)                                                                     \
                    }                                                                                             \
                }                                                                                                 \
            }                                                                                                     \
        }
        // (Measured dead end: letting the second wave quartet stage in the MIDDLE of its MFMA loop, so that the two waves of a
        // SIMD never stage at the same time, changes nothing - 5.16 against 5.06 ms on the level-1 gate gradient.  VALU work
        // is not hidden by the other wave's MFMAs on the same SIMD; only less staging arithmetic would help.)
        {
            const int u0 = px0 >> 4, u1 = (px0 + npx) >> 4, k0 = px0 >> 2, k1 = (px0 + npx) >> 2;
            TMG_WG_STAGE_ROUND
            TMG_WG_RUN(u0, u1, k0, k1)
        }
#undef TMG_WG_LDF
#undef TMG_WG_LDG
#undef TMG_WG_RUN
#undef TMG_WG_STAGE_ROUND
        __syncthreads();  // the buffer just read may be overwritten next round; the one just written is complete
    }
#undef TMG_WG_MFMA
#undef TMG_SB
#undef TMG_WG_ISSUE
#undef TMG_WG_COMMIT
#undef TMG_WG_ISSUE_F
#undef TMG_WG_COMMIT_F
#undef TMG_WG_PATCH_ITEM
#undef TMG_WG_ORIGIN

    // Waves holding the same pairs (all 8 with ksplit, the two quartets otherwise) fold their partial sums through LDS,
    // upper half into lower half, until one copy per pair is left: 8x / 2x less slab traffic for the reduce kernel.
    const int nws = ksplit ? 1 : 4;  // waves that still hold sums afterwards
    for (int half = 4; half >= nws; half >>= 1) {
        __syncthreads();
        if (wave >= half && wave < 2 * half) {
            float4* dst = reinterpret_cast<float4*>(lds) + ((wave - half) * NP * NCO) * 64 + lane;
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int n = 0; n < NCO; ++n)
                    dst[(j * NCO + n) * 64] = make_float4(acc[j][n][0], acc[j][n][1], acc[j][n][2], acc[j][n][3]);
        }
        __syncthreads();
        if (wave < half) {
            const float4* src = reinterpret_cast<const float4*>(lds) + (wave * NP * NCO) * 64 + lane;
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int n = 0; n < NCO; ++n) {
                    const float4 v = src[(j * NCO + n) * 64];
                    acc[j][n][0] += v.x; acc[j][n][1] += v.y; acc[j][n][2] += v.z; acc[j][n][3] += v.w;
                }
        }
    }
    // dbias: fold the per-thread partials (thread t owns channels 4*(t % (NCO*4)) ..+3) through LDS
    float bsum = 0.f;
    const bool do_bias = p.dbias && blockIdx.z == 0;
    if (do_bias) {
        __syncthreads();
        *reinterpret_cast<float4*>(lds + tid * 4) = bacc;
        __syncthreads();
        if (tid < NCO * 16) {
            const int c4 = tid >> 2, e = tid & 3;
            for (int t = c4; t < NT; t += NCO * 4) bsum += lds[t * 4 + e];
        }
    }
    if (p.ws) {
        // partial sums go to this block's slab in accumulator order (coalesced 16-byte stores); the reduce kernel
        // folds the slabs into dW.  Avoids ~1e7 contended float atomics on a KB-sized dW.
        const size_t bl = ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z;
        if (wave < nws) {
            float4* slab = reinterpret_cast<float4*>(p.ws) + ((bl * nws + wave) * NP * NCO) * 64 + lane;
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int n = 0; n < NCO; ++n)
                    slab[(j * NCO + n) * 64] = make_float4(acc[j][n][0], acc[j][n][1], acc[j][n][2], acc[j][n][3]);
        }
        if (do_bias && tid < NCO * 16) {
            float* wsb = p.ws + (size_t)gridDim.x * gridDim.y * gridDim.z * nws * NP * NCO * 256;
            wsb[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 64 + tid] = bsum;
        }
        return;
    }
    if (wave >= nws) return;
    const float osc = out_scale_of(p.kappa);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int pid = ksplit ? j : w4 + 4 * j;
        if (pid >= npairs) continue;
        const int citg = (pair0 + pid) / ntaps, tap = pair0 + pid - citg * ntaps;
#pragma unroll
        for (int n = 0; n < NCO; ++n) {
            const int co = co0 + n * 16 + li;
            if (co >= p.Cout) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = citg * 16 + q * 4 + r;
                if (ci < p.cin_valid)
                    atomicAdd(p.dW + ((size_t)co * p.cin_dst + ci + (ci < p.ci_split ? p.ci_off0 : p.ci_off1)) * ntaps + tap, acc[j][n][r] * osc);
            }
        }
    }
    if (do_bias && tid < NCO * 16 && co0 + tid < p.Cout) atomicAdd(p.dbias + co0 + tid, bsum * osc);
}

// Fold the per-block slabs of conv_wgrad_kernel into dW / dbias.  grid = (ceil(items/256), xchunks):
// each thread sums one accumulator float4 over a chunk of the pixel-share (x) dimension, then adds it to dW.
__global__ void conv_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, float* __restrict__ dbias,
                                         const float* __restrict__ kappa, int gx, int gy, int gz, int NP, int NCO, int PPG,
                                         int pairs_total, int Cin, int Cout, int ntaps, int xchunk, int cin_valid, int ci_split,
                                         int ci_off0, int ci_off1, int ksplit, int bpg, long long dw_gstride, int db_gstride) {
    // gy counts the blocks in y of the wgrad launch; with groups (bpg < gy) block y belongs to group y / bpg
    const int nws = ksplit ? 1 : 4;  // wave copies per block that survive the in-block fold
    const int items = gy * gz * nws * NP * NCO * 64;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int x0 = blockIdx.y * xchunk, x1 = min(gx, x0 + xchunk);
    const float osc = out_scale_of(kappa);
    if (i < items) {
        int r_ = i;
        const int lane = r_ & 63; r_ >>= 6;
        const int n = r_ % NCO; r_ /= NCO;
        const int j = r_ % NP; r_ /= NP;
        const int wave = r_ % nws; r_ /= nws;
        const int z = r_ % gz;
        const int y = r_ / gz;
        const int pid = ksplit ? j : wave + 4 * j;
        if (pid < min(PPG, pairs_total - z * PPG)) {
            const size_t per_x = (size_t)gy * gz * nws * NP * NCO * 64;
            const float4* src = reinterpret_cast<const float4*>(ws) + (((((size_t)y * gz + z) * nws + wave) * NP + j) * NCO + n) * 64 + lane;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            // batches of 8 independent loads (slab index clamped, surplus weighted 0): a plain loop waits for every load
            for (int xb = x0; xb < x1; xb += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(size_t)min(xb + u, x1 - 1) * per_x];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float wgt = (xb + u < x1) ? 1.f : 0.f;
                    a.x += wgt * v[u].x; a.y += wgt * v[u].y; a.z += wgt * v[u].z; a.w += wgt * v[u].w;
                }
            }
            const int citg = (z * PPG + pid) / ntaps, tap = z * PPG + pid - citg * ntaps;
            const int grp = y / bpg;
            const int co = ((y - grp * bpg) * NCO + n) * 16 + (lane & 15);
            const int ci = citg * 16 + (lane >> 4) * 4;
            dW += (size_t)grp * dw_gstride;
            if (co < Cout) {
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ci + r < cin_valid)   // Cin == destination row length here
                        atomicAdd(dW + ((size_t)co * Cin + ci + r + (ci + r < ci_split ? ci_off0 : ci_off1)) * ntaps + tap, av[r] * osc);
            }
        }
    }
    // bias partials: [gx][gy][64]
    if (dbias && i < gy * 64) {
        const int y = i >> 6, t = i & 63;
        const int grp = y / bpg;
        const int co = (y - grp * bpg) * NCO * 16 + t;
        dbias += (size_t)grp * db_gstride;
        if (t < NCO * 16 && co < Cout) {
            const float* wsb = ws + (size_t)gx * gy * gz * nws * NP * NCO * 256;
            float a = 0.f;
            for (int xb = x0; xb < x1; xb += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = wsb[((size_t)min(xb + u, x1 - 1) * gy + y) * 64 + t];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += (xb + u < x1) ? v[u] : 0.f;
            }
            atomicAdd(dbias + co, a * osc);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Weight packer: torch layout W[Cout][Cin][k][k]  ->  wpk[tap][Cin_pad/16][Cout_pad][16]
// mode 0: forward operand.   mode 1: input-gradient operand (roles of Cin/Cout swapped, taps flipped):
//         wpk[tap][co/16][ci][co%16] = W[co][ci][ntaps-1-tap]
// ---------------------------------------------------------------------------------------------
__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cout, int Cin, int ntaps,
                                 int Kpad, int Npad, int mode, int cvalid, int csplit, int cgap, size_t w_bstride, size_t wpk_bstride) {
    // blockIdx.y: weight tensor of a batch (the layers of a flow level packed in one launch)
    w += blockIdx.y * w_bstride;
    wpk += blockIdx.y * wpk_bstride;
    const size_t total = (size_t)ntaps * Kpad * Npad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c16 = i & 15;
        size_t r = i >> 4;
        const int n = r % Npad;
        r /= Npad;
        const int kb = r % (Kpad >> 4);
        const int tap = r / (Kpad >> 4);
        const int k = kb * 16 + c16;
        float v = 0.f;
        // operand input channel c maps to source channel c (+ cgap when c >= csplit); channels >= cvalid are zero
        if (mode == 0) {
            if (k < cvalid && n < Cout) v = w[((size_t)n * Cin + k + (k < csplit ? 0 : cgap)) * ntaps + tap];
        } else {
            if (k < Cout && n < cvalid) v = w[((size_t)k * Cin + n + (n < csplit ? 0 : cgap)) * ntaps + (ntaps - 1 - tap)];
        }
        wpk[i] = v;
    }
}

// Up to 16 independent packing jobs (different tensors, shapes and modes) in one launch: the jobs travel in the kernel arguments
// (blockIdx.y selects one: a block-uniform read of the kernarg segment), so there is no device table to upload.  A dense block packs
// the forward and the input-gradient operands of all its layers with one launch, a plain conv node both of its operands.
struct PackJob {
    const float* w; float* wpk;
    int Cout, Cin, ntaps, Kpad, Npad, mode, cvalid, csplit, cgap;
};
constexpr int TMG_PACK_JOBS = 48;     // 56 bytes each: 2.7 KB of kernel arguments
struct PackJobs { PackJob j[TMG_PACK_JOBS]; };

__global__ void conv_pack_many_kernel(PackJobs J) {
    const PackJob& jb = J.j[blockIdx.y];
    const float* __restrict__ w = jb.w;
    float* __restrict__ wpk = jb.wpk;
    const int Cout = jb.Cout, Cin = jb.Cin, ntaps = jb.ntaps, Kpad = jb.Kpad, Npad = jb.Npad, mode = jb.mode;
    const int cvalid = jb.cvalid, csplit = jb.csplit, cgap = jb.cgap;
    const size_t total = (size_t)ntaps * Kpad * Npad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c16 = i & 15;
        size_t r = i >> 4;
        const int n = r % Npad;
        r /= Npad;
        const int kb = r % (Kpad >> 4);
        const int tap = r / (Kpad >> 4);
        const int k = kb * 16 + c16;
        float v = 0.f;
        if (mode == 0) {
            if (k < cvalid && n < Cout) v = w[((size_t)n * Cin + k + (k < csplit ? 0 : cgap)) * ntaps + tap];
        } else {
            if (k < Cout && n < cvalid) v = w[((size_t)k * Cin + n + (n < csplit ? 0 : cgap)) * ntaps + (ntaps - 1 - tap)];
        }
        wpk[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Replicate-padding input-gradient fix-up.  The main kernel computes the gradient w.r.t. the
// in-image part of the padded input; the ring of padded pixels folds onto the border pixels:
//   dx(q) += sum over ring pixels rho with clamp(rho) == q of sum_tap W_tap^T dy(rho - tap + 1)
// Output channel set / segments as in the main kernel.  One thread per (border pixel, out channel).
// ---------------------------------------------------------------------------------------------
struct BorderP {
    const float* dy;  // [B,H,W] x Cdy, stride/off
    int dy_stride, dy_off, Cdy;
    const float* wpk;  // input-gradient packed weights [tap'][Cdy_pad/16][Cx_pad][16] (tmg_conv_pack mode 1)
    int Cx, Npad, KB, dy_vec;
    const float* kappa;
    int B, H, W;
    TmgOSeg out[TMG_MAX_OUT_SEG];
    int nborder;  // border pixels per image
};

__device__ __forceinline__ void border_pixel(int idx, int H, int W, int* y, int* x) {
    // enumerate the image border: top row, bottom row, then left/right columns without corners
    if (idx < W) { *y = 0; *x = idx; return; }
    idx -= W;
    if (H > 1) {
        if (idx < W) { *y = H - 1; *x = idx; return; }
        idx -= W;
    }
    const int rows = H - 2;
    if (idx < rows) { *y = 1 + idx; *x = 0; return; }
    idx -= rows;
    *y = 1 + idx; *x = W - 1;
}

__global__ void conv_rep_border_fix_kernel(BorderP p) {
    const size_t total = (size_t)p.B * p.nborder * p.Cx;
    const float osc = out_scale_of(p.kappa);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = i % p.Cx;
        size_t r = i / p.Cx;
        const int bi = r % p.nborder;
        const int b = r / p.nborder;
        int qy, qx;
        border_pixel(bi, p.H, p.W, &qy, &qx);
        float acc = 0.f;
        // ring pixels that clamp onto (qy,qx): rho = (qy+dy, qx+dx) with dy,dx in {-1,0,1}, outside the image
        for (int dyy = -1; dyy <= 1; ++dyy) {
            const int ry = qy + dyy;
            if (dyy != 0 && !((dyy < 0 && qy == 0) || (dyy > 0 && qy == p.H - 1))) continue;
            for (int dxx = -1; dxx <= 1; ++dxx) {
                const int rx = qx + dxx;
                if (dxx != 0 && !((dxx < 0 && qx == 0) || (dxx > 0 && qx == p.W - 1))) continue;
                if (dyy == 0 && dxx == 0) continue;
                const bool outside = (ry < 0 || ry >= p.H || rx < 0 || rx >= p.W);
                if (!outside) continue;
                // forward: y(pp) uses padded pixel rho = pp + (ky-1, kx-1)  ->  pp = rho - (ky-1, kx-1)
                for (int ky = 0; ky < 3; ++ky) {
                    const int py = ry - (ky - 1);
                    if (py < 0 || py >= p.H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int px = rx - (kx - 1);
                        if (px < 0 || px >= p.W) continue;
                        const float* d = p.dy + (((size_t)b * p.H + py) * p.W + px) * p.dy_stride + p.dy_off;
                        // packed operand: 16 consecutive dy-channels of one ci are contiguous, consecutive ci adjacent
                        const float* wq = p.wpk + (((size_t)(8 - (ky * 3 + kx)) * p.KB) * p.Npad + ci) * 16;
                        float sacc = 0.f;
                        if (p.dy_vec) {
                            for (int kb = 0; kb < p.KB; ++kb) {
                                const float4* w4 = reinterpret_cast<const float4*>(wq + (size_t)kb * p.Npad * 16);
                                const float4* d4 = reinterpret_cast<const float4*>(d + kb * 16);
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float4 a = w4[e], v = d4[e];
                                    sacc += a.x * v.x + a.y * v.y + a.z * v.z + a.w * v.w;
                                }
                            }
                        } else {
                            for (int co = 0; co < p.Cdy; ++co) sacc += d[co] * wq[(size_t)(co >> 4) * p.Npad * 16 + (co & 15)];
                        }
                        acc += sacc;
                    }
                }
            }
        }
        int nl = ci;
        TMG_PICK_OSEG(p.out, nl, optr, ostride, ooff)
        float* dst = optr + (((size_t)b * p.H + qy) * p.W + qx) * ostride + ooff + nl;
        *dst += acc * osc;
    }
}

// The same fold on the matrix cores.  The scalar kernel above re-reads the operand per (pixel, channel): at the level-wide
// conditioning contraction (240..1920 dy channels) that was 3 GB of L2 reads per launch for 1.6 GFLOP of work.  Here the border
// pixels are the GEMM's N dimension: they are sorted into 8 classes (left / right / top / bottom edge without corners, 4 corners),
// a wave owns 16 pixels of ONE class, so its pixels share the class's list of (operand tap, source offset) pairs and every
// operand fragment read from L2 serves 16 pixels.  D = Wt[tap] (16 channels x 4 k) x dy(source pixel)^T (4 k x 16 pixels), both
// operands straight from global memory (16-byte loads, no LDS), accumulated over pairs x dy channels; lane (pixel, q) then
// adds its 4 consecutive channels onto dx with one float4 read-modify-write.
struct BorderMP {
    BorderP b;
    int tile0[17];  // first 16-pixel tile of virtual class v (tile0[16] = total).  v < 4: the edge classes; v = 4 + 3 k + g: corner k
                    // with pair group g (pairs [0,3), [3,6), [6,7) of its list: a corner's 7 pairs as one serial chain of operand
                    // loads made the 16 corner tiles the long pole of the launch)
    int cnt[8];     // pixels of class c per image
    int ksplit;     // waves sharing one tile's dy-channel range (long contractions over few border pixels)
};

// pairs of class c: entry = operand tap | (dy + 1) << 4 | (dx + 1) << 6 ; classes: 0 L, 1 R, 2 T, 3 B, 4 TL, 5 TR, 6 BL, 7 BR
#define TMG_BP(TAP, DY, DX) ((unsigned)(TAP) | ((unsigned)((DY) + 1) << 4) | ((unsigned)((DX) + 1) << 6))
static __device__ const unsigned g_border_pairs[8][8] = {
    {3, TMG_BP(2, -1, 0), TMG_BP(5, 0, 0), TMG_BP(8, 1, 0)},
    {3, TMG_BP(0, -1, 0), TMG_BP(3, 0, 0), TMG_BP(6, 1, 0)},
    {3, TMG_BP(6, 0, -1), TMG_BP(7, 0, 0), TMG_BP(8, 0, 1)},
    {3, TMG_BP(0, 0, -1), TMG_BP(1, 0, 0), TMG_BP(2, 0, 1)},
    {7, TMG_BP(6, 0, -1), TMG_BP(7, 0, 0), TMG_BP(8, 0, 1), TMG_BP(2, -1, 0), TMG_BP(5, 0, 0), TMG_BP(8, 1, 0), TMG_BP(8, 0, 0)},
    {7, TMG_BP(6, 0, -1), TMG_BP(7, 0, 0), TMG_BP(8, 0, 1), TMG_BP(0, -1, 0), TMG_BP(3, 0, 0), TMG_BP(6, 1, 0), TMG_BP(6, 0, 0)},
    {7, TMG_BP(0, 0, -1), TMG_BP(1, 0, 0), TMG_BP(2, 0, 1), TMG_BP(2, -1, 0), TMG_BP(5, 0, 0), TMG_BP(8, 1, 0), TMG_BP(2, 0, 0)},
    {7, TMG_BP(0, 0, -1), TMG_BP(1, 0, 0), TMG_BP(2, 0, 1), TMG_BP(0, -1, 0), TMG_BP(3, 0, 0), TMG_BP(6, 1, 0), TMG_BP(0, 0, 0)}};
#undef TMG_BP

template <int NT>
__global__ __launch_bounds__(256) void conv_rep_border_mfma_kernel(BorderMP m) {
    const BorderP& p = m.b;
    constexpr int U = NT <= 5 ? 4 : 2;   // k-blocks whose operand loads are in flight together (the chain is latency-bound)
    const int lane = threadIdx.x & 63, li = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const float osc = out_scale_of(p.kappa);
    const size_t tap_stride = (size_t)p.KB * p.Npad * 16, kb_stride = (size_t)p.Npad * 16;
    const int S = m.ksplit, KBs = (p.KB + S - 1) / S;
    for (int item = (int)blockIdx.x * 4 + wave; item < m.tile0[16] * S; item += (int)gridDim.x * 4) {
        const int t = item / S, ks = item - t * S;
        const int kb0 = ks * KBs, kb1 = min(p.KB, kb0 + KBs);
        if (kb0 >= kb1) continue;
        int v = 0;
#pragma unroll
        for (int k = 1; k < 16; ++k) v += (t >= m.tile0[k]) ? 1 : 0;
        const int c = v < 4 ? v : 4 + (v - 4) / 3, pg = v < 4 ? 0 : (v - 4) % 3;
        const int cnt = m.cnt[c];
        const int i = (t - m.tile0[v]) * 16 + li;          // this lane's pixel of the class
        const bool pv = i < p.B * cnt;
        const int b = pv ? i / cnt : 0, j = pv ? i - b * cnt : 0;
        int qy, qx;
        if (c == 0) { qy = 1 + j; qx = 0; }
        else if (c == 1) { qy = 1 + j; qx = p.W - 1; }
        else if (c == 2) { qy = 0; qx = 1 + j; }
        else if (c == 3) { qy = p.H - 1; qx = 1 + j; }
        else { qy = (c & 2) ? p.H - 1 : 0; qx = (c & 1) ? p.W - 1 : 0; }
        f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int npairs = min((int)g_border_pairs[c][0], 3 * pg + 3);
        for (int e = 3 * pg; e < npairs; ++e) {
            const unsigned ent = g_border_pairs[c][1 + e];
            const int sy = qy + (int)((ent >> 4) & 3u) - 1, sx = qx + (int)((ent >> 6) & 3u) - 1;
            const bool sv = pv && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
            const float* src = p.dy + (((size_t)b * p.H + (sv ? sy : 0)) * p.W + (sv ? sx : 0)) * p.dy_stride + p.dy_off + 4 * q;
            const float* wt = p.wpk + (size_t)(ent & 15u) * tap_stride + li * 16 + 4 * q;
            for (int kb = kb0; kb < kb1; kb += U) {
                float4 x4[U], w4[U][NT];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kk = kb + u;
                    x4[u] = (sv && kk < kb1 && kk * 16 + 4 * q < p.Cdy) ? *reinterpret_cast<const float4*>(src + kk * 16)
                                                                        : *reinterpret_cast<const float4*>(g_tmg_zero_page);
                    const float* wk = wt + (size_t)min(kk, kb1 - 1) * kb_stride;   // past the range: any valid slice (x4 is zero)
#pragma unroll
                    for (int n = 0; n < NT; ++n) w4[u][n] = *reinterpret_cast<const float4*>(wk + n * 256);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[u][n].x, x4[u].x, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[u][n].y, x4[u].y, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[u][n].z, x4[u].z, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[u][n].w, x4[u].w, acc[n], 0, 0, 0);
                    }
            }
        }
        if (pv) {
            const size_t opx = ((size_t)b * p.H + qy) * p.W + qx;
            float* dst[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                int nl = min(n * 16 + 4 * q, p.Cx - 4);      // (a channel quad past the end: a valid address, never stored)
                TMG_PICK_OSEG(p.out, nl, optr, ostride, ooff)
                dst[n] = optr + opx * ostride + ooff + nl;
            }
            if (S > 1 || v >= 4) {
                // the dy channels (or, at a corner, the pairs) of this pixel are split over several waves: hardware float adds
                // (a few thousand border pixels)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    if (n * 16 + 4 * q < p.Cx) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(dst[n] + r, acc[n][r] * osc);
                    }
            } else {
                // read-modify-write of the NT channel quads: all reads first, then all writes (one read + write per quad in turn is a
                // chain of NT memory round trips - a store behind a load in a conditional block waits for everything in flight)
                float4 o[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) o[n] = *reinterpret_cast<const float4*>(dst[n]);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    if (n * 16 + 4 * q < p.Cx) {
                        o[n].x += acc[n][0] * osc; o[n].y += acc[n][1] * osc; o[n].z += acc[n][2] * osc; o[n].w += acc[n][3] * osc;
                        *reinterpret_cast<float4*>(dst[n]) = o[n];
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Generic direct input-gradient (any stride) for the encoder's two stride-2 convs (tiny FLOPs).
// dx[b,iy,ix,ci] = sum_{ky,kx,co : iy+pad-ky = s*oy, ix+pad-kx = s*ox} W[co][ci][ky][kx] * dy[b,oy,ox,co]
// ---------------------------------------------------------------------------------------------
__global__ void conv_dgrad_direct_kernel(const float* __restrict__ dy, int dy_stride, int dy_off, const float* __restrict__ w,
                                         float* __restrict__ dx, int dx_stride, int dx_off, int B, int Hin, int Win, int Hout,
                                         int Wout, int Cin, int Cout, int ksize, int stride, int accumulate) {
    const int pad = ksize >> 1;
    const size_t total = (size_t)B * Hin * Win * Cin;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = i % Cin;
        size_t r = i / Cin;
        const int ix = r % Win;
        r /= Win;
        const int iy = r % Hin;
        const int b = r / Hin;
        float acc = 0.f;
        for (int ky = 0; ky < ksize; ++ky) {
            const int ny = iy + pad - ky;
            if (ny < 0 || ny % stride) continue;
            const int oy = ny / stride;
            if (oy >= Hout) continue;
            for (int kx = 0; kx < ksize; ++kx) {
                const int nx = ix + pad - kx;
                if (nx < 0 || nx % stride) continue;
                const int ox = nx / stride;
                if (ox >= Wout) continue;
                const float* d = dy + (((size_t)b * Hout + oy) * Wout + ox) * dy_stride + dy_off;
                const float* wp = w + ((size_t)ci * ksize + ky) * ksize + kx;
                for (int co = 0; co < Cout; ++co) acc += d[co] * wp[(size_t)co * Cin * ksize * ksize];
            }
        }
        float* dst = dx + (((size_t)b * Hin + iy) * Win + ix) * dx_stride + dx_off + ci;
        *dst = accumulate ? (*dst + acc) : acc;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>
// Optional per-launch timing with HIP events on the launch stream (bench.py's roofline line).  Events come from a pool that is
// created when profiling is switched on (no hipEventCreate inside a timed region) and the registry is guarded by a mutex (the
// reference drives one replica per host thread, utils/parallel.py:222-231).  Kernel ids < 32: matrix-core kernels, work =
// algorithmic flops; ids >= 32: bandwidth-bound kernel classes, work = algorithmic HBM bytes.
struct ProfRec { hipEvent_t a, b; int kid; double work; };
static std::vector<ProfRec> g_prof;        // records in use: g_prof[0 .. g_prof_n)
static size_t g_prof_n = 0;
static int g_prof_on = 0;
static std::mutex g_prof_mu;
static const char* const g_prof_names[] = {
    "conv_mfma_kernel<4,1,4,1>", "conv_mfma_kernel<4,2,4,1>", "conv_mfma_kernel<4,3,4,1>", "conv_mfma_kernel<4,4,4,1>",
    "conv_mfma_kernel<4,3,2,2>", "conv_mfma_kernel<4,4,2,2>", "conv_mfma_kernel<4,3,1,4>", "conv_mfma_kernel<4,4,1,4>",
    "conv_wgrad_kernel<3,1,*>", "conv_wgrad_kernel<3,2,*>", "conv_wgrad_kernel<3,4,*>",
    "conv_fwd_kernel<*,1,8,1>", "conv_fwd_kernel<*,2,8,1>", "conv_fwd_kernel<*,3,8,1>", "conv_fwd_kernel<*,4,8,1>",
    "conv_fwd_kernel<*,3,4,2>", "conv_fwd_kernel<*,4,4,2>", "conv_fwd_kernel<*,3,2,4>", "conv_fwd_kernel<*,4,2,4>",
    "conv_wgrad_kernel<5,1,*>", "conv_wgrad_kernel<5,2,*>", "conv_wgrad_kernel<5,4,*>",
    "conv_wgrad_kernel<7,1,*>", "conv_wgrad_kernel<7,2,*>", "conv_wgrad_kernel<7,4,*>",
    "conv_wgrad_kernel<9,1,*>", "conv_wgrad_kernel<9,2,*>", "conv_wgrad_kernel<9,4,*>",
    "conv_wgrad_kernel<8,1,*>", "conv_wgrad_kernel<8,2,*>", "conv_wgrad_kernel<8,4,*>",
    "conv 1x1 (invertible channel mix, fp32 MFMA)",
    "hbm: cpl_fwd_kernel (zero conv + coupling + mix)", "hbm: c1x2_fwd_kernel (growth layers)", "hbm: dense2_bwd_kernel",
    "hbm: affine_apply_kernel", "hbm: affine_bwd_kernel", "hbm: lstm_pointwise_fwd_kernel", "hbm: lstm_pointwise_bwd_kernel",
    "hbm: gauss_fwd/bwd_kernel", "hbm: checker / upsample", "hbm: mix16_kernel (fp16-input channel mix)",
    "hbm: cpl_bwd_kernel (mix dgrad + coupling backward + zero-conv dgrad)",
    "wino_fwd / wino_nn_kernel (3x3 conv, Winograd F(2x2,3x3), fp32 MFMA; direct-algorithm flops)",
    "wino_wgrad_kernel (3x3 weight gradient, Winograd F(3x3,2x2), fp32 MFMA; direct-algorithm flops)"};
#define TMG_NPROF 45

// open / close a timed region around a launch; other translation units reach them through tmg_common.h's TmgProf
extern "C" int tmg_prof_open(int kid, double work, hipStream_t st) {
    // mode 1: matrix-core kernels only (ids < 32 and the Winograd classes 43, 44); mode 2: + the bandwidth-bound classes 32..42
    if (!g_prof_on || (g_prof_on == 1 && kid >= 32 && kid < 43) || (g_prof_on >= 100 && kid != g_prof_on - 100)) return -1;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on || g_prof_n >= g_prof.size()) return -1;
    ProfRec& r = g_prof[g_prof_n];
    r.kid = kid; r.work = work;
    (void)hipEventRecord(r.a, st);
    return (int)g_prof_n++;
}
extern "C" void tmg_prof_close(int slot, hipStream_t st) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if ((size_t)slot < g_prof_n) (void)hipEventRecord(g_prof[slot].b, st);
}

struct ProfScope {
    int slot; hipStream_t st;
    ProfScope(int kid, double flops, hipStream_t s) : slot(tmg_prof_open(kid, flops, s)), st(s) {}
    ~ProfScope() { tmg_prof_close(slot, st); }
};

// on = 1: time the matrix-core kernels (ids < 32); on = 2: also the bandwidth-bound classes; on = 100 + k: only kernel id k (what
// bench.py uses inside its timed region: a handful of event pairs per step instead of hundreds); 0: off.  Clears the records and
// (first time) creates the event pool.
extern "C" int tmg_prof_enable(int64_t on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_n = 0;
    if (on) {
        const size_t want = 65536;
        while (g_prof.size() < want) {
            ProfRec r;
            if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) break;
            r.kid = 0; r.work = 0.0;
            g_prof.push_back(r);
        }
    }
    g_prof_on = (int)on;
    return 0;
}

// out[kid*3 + {0,1,2}] = {launch count, total ms, total algorithmic work (flops or bytes)}; returns number of kernel ids
extern "C" int tmg_prof_collect(double* out, int64_t nk) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < (int)nk * 3; ++i) out[i] = 0.0;
    for (size_t i = 0; i < g_prof_n; ++i) {
        ProfRec& r = g_prof[i];
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        if (r.kid < nk) { out[r.kid * 3] += 1.0; out[r.kid * 3 + 1] += ms; out[r.kid * 3 + 2] += r.work; }
    }
    return TMG_NPROF;
}

extern "C" const char* tmg_prof_name(int64_t kid) { return (kid >= 0 && kid < TMG_NPROF) ? g_prof_names[kid] : ""; }

static inline int ilog2_ceil(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

template <int MT, int NTW, int WM, int WN>
static int launch_conv(const ConvP& p, int gy, size_t lds_bytes, hipStream_t st) {
    TMG_LDS_OPTIN((&conv_mfma_kernel<MT, NTW, WM, WN>));
    dim3 grid(p.B * p.tiles_x * p.tiles_y, gy, 1);
    const int kid = (WM == 4 ? NTW - 1 : (WM == 2 ? NTW + 1 : NTW + 3));
    ProfScope prof(kid, 2.0 * p.B * p.Hout * p.Wout * (double)p.Cout * p.Cin * p.ksize * p.ksize, st);
    hipLaunchKernelGGL((conv_mfma_kernel<MT, NTW, WM, WN>), grid, dim3(256), lds_bytes, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

template <int MT, int NTW, int WM, int WN>
static int launch_fwd(const ConvP& p, int G, int gy, size_t lds_bytes, hipStream_t st) {
    TMG_LDS_OPTIN((&conv_fwd_kernel<MT, NTW, WM, WN>));
    const int kid = p.ksize == 1 ? 31 : 11 + (WM == 8 ? NTW - 1 : (WM == 4 ? NTW + 1 : NTW + 3));
    ProfScope prof(kid, 2.0 * p.B * p.Hout * p.Wout * (double)p.Cout * p.Cin * p.ksize * p.ksize, st);
    hipLaunchKernelGGL((conv_fwd_kernel<MT, NTW, WM, WN>), dim3(G, gy, 1), dim3(512), lds_bytes, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

// Tile / chunk plan and launch of conv_fwd_kernel; -100 when the shape is not eligible (caller falls back to conv_mfma_kernel).
static int conv_fwd_lean(ConvP p, hipStream_t st) {
    if (p.stride != 1 || !p.vec4 || (p.Cin & 3) || (long)p.Hin * p.Win >= (1 << 24) || (long)p.B * p.Hout >= (1 << 24) ||
        p.Wout >= (1 << 24) || (long)p.B * p.Hout * p.Wout >= (1L << 32))
        return -100;
    for (int i = 0; i < p.nseg; ++i)
        if (p.in[i].stride >= (1 << 24) || (long)p.B * p.Hin * p.Win * p.in[i].stride >= (1L << 31)) return -100;
    // float4 epilogue: every output segment (and the add operand) addressable in channel quads
    p.ovec4 = (p.Cout & 3) == 0;
    for (int i = 0; i < p.nout; ++i)
        if (((p.out[i].stride | p.out[i].off | p.out[i].n) & 3) || (((uintptr_t)p.out[i].p) & 15)) p.ovec4 = 0;
    if (p.add.p && (((p.add.stride | p.add.off) & 3) || (((uintptr_t)p.add.p) & 15))) p.ovec4 = 0;
    const int ntt = p.Cout_pad >> 4;
    int WM, WN, NTW;
    if (ntt <= 4) { WM = 8; WN = 1; NTW = ntt; }
    else if (ntt <= 8) { WM = 4; WN = 2; NTW = ntt <= 6 ? 3 : 4; }
    else { WM = 2; WN = 4; NTW = ntt <= 12 ? 3 : 4; }
    // TMG_FWD_PLAN=MT,WM,WN,NTW,GMUL,KCHMAX (0 = planner's choice): launch-plan override for measurements (tools/bench_wide.py)
    static int fp[6] = {-1, 0, 0, 0, 0, 0};
    if (fp[0] < 0) {
        fp[0] = 0;
        if (const char* e = getenv("TMG_FWD_PLAN")) sscanf(e, "%d,%d,%d,%d,%d,%d", &fp[0], &fp[1], &fp[2], &fp[3], &fp[4], &fp[5]);
    }
    if (fp[1] > 0 && fp[2] > 0 && fp[3] > 0 && p.ksize == 3) { WM = fp[1]; WN = fp[2]; NTW = fp[3]; }
    const int gy = (ntt + WN * NTW - 1) / (WN * NTW);
    const long npix = (long)p.B * p.Hout * p.Wout;
    const int halo = p.ksize >> 1;
    for (int MT = 4; MT >= 1; MT >>= 1) {
        // m-tiles per wave: 4 when the image is large; fewer when that would leave CUs idle (small levels)
        if (fp[0] > 0 && p.ksize == 3) { if (MT != fp[0]) continue; }
        else if (MT > 1 && (npix / (16 * MT * WM)) * gy < 256) continue;
        const int MBLK = 16 * MT * WM;
        int twl = ilog2_ceil(p.Wout);
        if (twl > 5) twl = 5;
        if (twl < 2) twl = 2;
        while ((1 << twl) > MBLK) --twl;
        while (twl < 5 && (MBLK >> twl) > p.Hout && (1 << twl) < p.Wout) ++twl;  // no taller than the image needs
        const int TW = 1 << twl, TH = MBLK >> twl;
        if (TH < 1) continue;
        const int PHPW = (TW + 2 * halo) * (TH + 2 * halo);
        // channel chunks: as few as possible, evenly sized, each fitting the register window (7 float4 x 512 threads,
        // float4 slots per pixel padded to a power of two) and two LDS buffers
        int nchunks = (p.Cin_pad + 63) / 64, kch = 0;
        if (fp[5] >= 16 && p.ksize == 3) nchunks = (p.Cin_pad + fp[5] - 1) / fp[5];
        for (; nchunks <= p.Cin_pad / 16; ++nchunks) {
            kch = (((p.Cin_pad / 16) + nchunks - 1) / nchunks) * 16;
            const int k4p = kch <= 16 ? 4 : (kch <= 32 ? 8 : 16);
            if (PHPW * k4p <= 7 * 512 && 2 * (size_t)PHPW * (kch + 8) * 4 <= 160 * 1024) break;
            kch = 0;
        }
        if (!kch) continue;
        p.TW_log2 = twl; p.TH = TH;
        p.tiles_x = (p.Wout + TW - 1) / TW;
        p.tiles_y = (p.Hout + TH - 1) / TH;
        p.ntiles = p.B * p.tiles_x * p.tiles_y;
        p.KCH = kch;
        p.nchunks = (p.Cin_pad + kch - 1) / kch;
        int G = tmg_num_cus() / gy;
        if (fp[4] > 0 && p.ksize == 3) G *= fp[4];
        if (G < 1) G = 1;
        if (G > p.ntiles) G = p.ntiles;
        const size_t lds_bytes = 2 * (size_t)PHPW * (kch + 8) * 4;
#define TMG_FWD_CASE(NTW_, WM_, WN_)                                                                   \
        if (NTW == NTW_ && WM == WM_ && WN == WN_) {                                                   \
            if (MT == 4) return launch_fwd<4, NTW_, WM_, WN_>(p, G, gy, lds_bytes, st);                \
            if (MT == 2) return launch_fwd<2, NTW_, WM_, WN_>(p, G, gy, lds_bytes, st);                \
            return launch_fwd<1, NTW_, WM_, WN_>(p, G, gy, lds_bytes, st);                             \
        }
        TMG_FWD_CASE(1, 8, 1) TMG_FWD_CASE(2, 8, 1) TMG_FWD_CASE(3, 8, 1) TMG_FWD_CASE(4, 8, 1)
        TMG_FWD_CASE(3, 4, 2) TMG_FWD_CASE(4, 4, 2) TMG_FWD_CASE(3, 2, 4) TMG_FWD_CASE(4, 2, 4)
        TMG_FWD_CASE(2, 4, 2) TMG_FWD_CASE(1, 4, 2) TMG_FWD_CASE(2, 2, 4) TMG_FWD_CASE(1, 2, 4)
#undef TMG_FWD_CASE
        return -100;
    }
    return -100;
}

static void fill_segs(TmgSeg* dst, const void* const* ptrs, const int64_t* desc, int n, int* vec4) {
    for (int i = 0; i < TMG_MAX_IN_SEG; ++i) dst[i] = TmgSeg{nullptr, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        dst[i].p = (const float*)ptrs[i];
        dst[i].stride = (int)desc[3 * i + 0];
        dst[i].off = (int)desc[3 * i + 1];
        dst[i].n = (int)desc[3 * i + 2];
        if ((dst[i].stride | dst[i].off | dst[i].n) & 3) *vec4 = 0;
        if (((uintptr_t)ptrs[i]) & 15) *vec4 = 0;
    }
}

extern "C" int tmg_conv_pack_map(const void* w, void* wpk, int64_t Cout, int64_t Cin, int64_t cin_eff, int64_t ksize, int64_t mode,
                                 const int64_t* map, hipStream_t st);

// cin_eff >= Cin: the operand is built for cin_eff input channels, the extra ones zero (lets a conv read a wider,
// 16-byte aligned segment list than the weight tensor has channels for).
extern "C" int tmg_conv_pack(const void* w, void* wpk, int64_t Cout, int64_t Cin, int64_t cin_eff, int64_t ksize, int64_t mode,
                             hipStream_t st) {
    const int64_t map[3] = {Cin, 0x7fffffff, 0};
    return tmg_conv_pack_map(w, wpk, Cout, Cin, cin_eff < Cin ? Cin : cin_eff, ksize, mode, map, st);
}

// As tmg_conv_pack with an input-channel map = {cvalid, csplit, cgap}: operand channel c < cvalid reads source channel
// c (+ cgap if c >= csplit); operand channels >= cvalid (up to cin_eff) are zero.  cin_eff may be smaller than Cin.
extern "C" int tmg_conv_pack_map(const void* w, void* wpk, int64_t Cout, int64_t Cin, int64_t cin_eff, int64_t ksize, int64_t mode,
                                 const int64_t* map, hipStream_t st) {
    const int ntaps = (int)(ksize * ksize);
    const int K = mode == 0 ? (int)cin_eff : (int)Cout, N = mode == 0 ? (int)Cout : (int)cin_eff;
    const int Kpad = (K + 15) & ~15, Npad = (N + 15) & ~15;
    const size_t total = (size_t)ntaps * Kpad * Npad;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(conv_pack_kernel, dim3(blocks), dim3(256), 0, st, (const float*)w, (float*)wpk, (int)Cout, (int)Cin, ntaps, Kpad, Npad,
                       (int)mode, (int)map[0], (int)map[1], (int)map[2], (size_t)0, (size_t)0);
    TMG_CHECK_LAUNCH();
    return 0;
}

// `nbatch` equally shaped weight tensors (w + b * Cout*Cin*k*k) packed in one launch into wpk + b * taps*Kpad*Npad.
// map = {cvalid, csplit, cgap} as tmg_conv_pack_map ({Cin, INT_MAX, 0}: identity).
extern "C" int tmg_conv_pack_batched(const void* w, void* wpk, int64_t nbatch, int64_t Cout, int64_t Cin, int64_t cin_eff, int64_t ksize,
                                      int64_t mode, const int64_t* map, hipStream_t st) {
    const int ntaps = (int)(ksize * ksize);
    const int K = mode == 0 ? (int)cin_eff : (int)Cout, N = mode == 0 ? (int)Cout : (int)cin_eff;
    const int Kpad = (K + 15) & ~15, Npad = (N + 15) & ~15;
    const size_t total = (size_t)ntaps * Kpad * Npad;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(conv_pack_kernel, dim3(blocks, (unsigned)nbatch), dim3(256), 0, st, (const float*)w, (float*)wpk, (int)Cout, (int)Cin, ntaps,
                       Kpad, Npad, (int)mode, (int)map[0], (int)map[1], (int)map[2], (size_t)(Cout * Cin * ntaps), total);
    TMG_CHECK_LAUNCH();
    return 0;
}

// njobs <= 16 packing jobs in one launch.  w / wpk: the source / destination pointer of every job; jobs: njobs x
// {Cout, Cin, cin_eff, ksize, mode, cvalid, csplit, cgap} with the meaning of tmg_conv_pack_map (cvalid = Cin, csplit = INT_MAX, cgap = 0:
// the identity map of tmg_conv_pack).
extern "C" int tmg_conv_pack_many(const void* const* w, void* const* wpk, const int64_t* jobs, int64_t njobs, hipStream_t st) {
    if (njobs < 1 || njobs > TMG_PACK_JOBS) return -3;
    PackJobs J;
    size_t maxtot = 0;
    for (int i = 0; i < TMG_PACK_JOBS; ++i) {
        const int s_ = i < (int)njobs ? i : 0;     // unused slots repeat job 0 (never selected: gridDim.y = njobs)
        const int64_t* d = jobs + 8 * s_;
        const int Cout = (int)d[0], Cin = (int)d[1], ce = (int)d[2], ks = (int)d[3], mode = (int)d[4];
        const int K = mode == 0 ? ce : Cout, N = mode == 0 ? Cout : ce;
        PackJob& jb = J.j[i];
        jb.w = (const float*)w[s_]; jb.wpk = (float*)wpk[s_];
        jb.Cout = Cout; jb.Cin = Cin; jb.ntaps = ks * ks; jb.Kpad = (K + 15) & ~15; jb.Npad = (N + 15) & ~15; jb.mode = mode;
        jb.cvalid = (int)d[5]; jb.csplit = (int)d[6]; jb.cgap = (int)d[7];
        const size_t tot = (size_t)jb.ntaps * jb.Kpad * jb.Npad;
        if (i < (int)njobs && tot > maxtot) maxtot = tot;
    }
    const int blocks = (int)((maxtot + 255) / 256 < 512 ? (maxtot + 255) / 256 : 512);
    hipLaunchKernelGGL(conv_pack_many_kernel, dim3(blocks < 1 ? 1 : blocks, (unsigned)njobs), dim3(256), 0, st, J);
    TMG_CHECK_LAUNCH();
    return 0;
}

extern "C" int tmg_conv_fwd_add(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* wpk, const void* bias,
                                const void* kappa, const void* in_scale, const void* in_shift, const void* add,
                                const int64_t* add_desc, void* const* out_ptrs, const int64_t* out_desc, int64_t nout,
                                const int64_t* dims, hipStream_t st);

// dims: [B,Hin,Win,Hout,Wout,ksize,stride,Cin,Cout,relu_in,pad_rep,relu_out,accumulate]
extern "C" int tmg_conv_fwd(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* wpk, const void* bias,
                            const void* kappa, const void* in_scale, const void* in_shift, void* const* out_ptrs,
                            const int64_t* out_desc, int64_t nout, const int64_t* dims, hipStream_t st) {
    return tmg_conv_fwd_add(in_ptrs, in_desc, nseg, wpk, bias, kappa, in_scale, in_shift, nullptr, nullptr, out_ptrs, out_desc, nout,
                            dims, st);
}

// As tmg_conv_fwd with an extra tensor `add` ({stride, off}, Cout channels) summed into the accumulator before the bias
// and the exp(kappa) scale:  out = [relu]((conv + add + bias) * scale).
extern "C" int tmg_conv_fwd_add(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* wpk, const void* bias,
                                const void* kappa, const void* in_scale, const void* in_shift, const void* add,
                                const int64_t* add_desc, void* const* out_ptrs, const int64_t* out_desc, int64_t nout,
                                const int64_t* dims, hipStream_t st) {
    ConvP p;
    p.add = TmgSeg{(const float*)add, add ? (int)add_desc[0] : 0, add ? (int)add_desc[1] : 0, 0};
    p.nseg = (int)nseg;
    p.vec4 = 1;
    fill_segs(p.in, in_ptrs, in_desc, (int)nseg, &p.vec4);
    p.B = (int)dims[0]; p.Hin = (int)dims[1]; p.Win = (int)dims[2]; p.Hout = (int)dims[3]; p.Wout = (int)dims[4];
    p.ksize = (int)dims[5]; p.stride = (int)dims[6]; p.Cin = (int)dims[7]; p.Cout = (int)dims[8];
    p.relu_in = (int)dims[9]; p.pad_rep = (int)dims[10]; p.relu_out = (int)dims[11]; p.accumulate = (int)dims[12];
    if (p.ksize != 1 && p.ksize != 3) return -2;
    int csum = 0;
    for (int i = 0; i < p.nseg; ++i) csum += p.in[i].n;
    if (csum != p.Cin || p.nseg < 1 || p.nseg > TMG_MAX_IN_SEG || nout < 1 || nout > TMG_MAX_OUT_SEG) return -3;
    p.Cin_pad = (p.Cin + 15) & ~15;
    p.Cout_pad = (p.Cout + 15) & ~15;
    p.wpk = (const float*)wpk; p.bias = (const float*)bias; p.kappa = (const float*)kappa;
    p.in_scale = (const float*)in_scale; p.in_shift = (const float*)in_shift;
    p.nout = (int)nout;
    int osum = 0;
    for (int i = 0; i < TMG_MAX_OUT_SEG; ++i) p.out[i] = TmgOSeg{nullptr, 0, 0, 0};
    for (int i = 0; i < p.nout; ++i) {
        p.out[i] = TmgOSeg{(float*)out_ptrs[i], (int)out_desc[3 * i], (int)out_desc[3 * i + 1], (int)out_desc[3 * i + 2]};
        osum += p.out[i].n;
    }
    if (osum != p.Cout) return -4;

    {
        const int rc = conv_fwd_lean(p, st);
        if (rc != -100) return rc;
    }

    // tile configuration
    const int ntt = p.Cout_pad >> 4;
    int WM, WN, NTW;
    if (ntt <= 4) { WM = 4; WN = 1; NTW = ntt; }
    else if (ntt <= 6) { WM = 2; WN = 2; NTW = 3; }
    else if (ntt <= 8) { WM = 2; WN = 2; NTW = 4; }
    else if (ntt <= 12) { WM = 1; WN = 4; NTW = 3; }
    else { WM = 1; WN = 4; NTW = 4; }
    const int gy = (ntt + WN * NTW - 1) / (WN * NTW);
    // m-tiles per wave: 4 when the image is large; fewer when that would leave CUs idle (small levels)
    int MT = 4;
    while (MT > 1) {
        const long npix = (long)p.B * p.Hout * p.Wout;
        if ((npix / (16 * MT * WM)) * gy >= 512) break;
        MT >>= 1;
    }
    const int MBLK = 16 * MT * WM;
    int twl = ilog2_ceil(p.Wout);
    if (twl > 5) twl = 5;
    if (twl < 2) twl = 2;
    while ((1 << twl) > MBLK) --twl;
    // keep the tile no taller than needed when the image is small
    while (twl < 5 && (MBLK >> twl) > p.Hout && (1 << twl) < p.Wout) ++twl;
    p.TW_log2 = twl;
    const int TW = 1 << twl;
    p.TH = MBLK >> twl;
    if (p.TH < 1) return -5;
    p.tiles_x = (p.Wout + TW - 1) / TW;
    p.tiles_y = (p.Hout + p.TH - 1) / p.TH;
    const int halo = p.ksize >> 1;
    const int PW = p.stride * (TW - 1) + 1 + 2 * halo, PH = p.stride * (p.TH - 1) + 1 + 2 * halo;
    // LDS per block: ~40 KB keeps 3-4 blocks per CU for the narrow-N kernels (their staging overlaps other blocks' MFMA
    // work); the wide-N kernels are register-limited to 2 blocks per CU and take the whole input patch in one chunk
    const int lds_budget = (WM == 1) ? 65536 : 40000;
    int kch = (int)(lds_budget / (4 * (size_t)PH * PW)) - 8;
    kch &= ~15;
    if (kch < 16) kch = 16;
    if (kch > p.Cin_pad) kch = p.Cin_pad;
    p.KCH = kch;
    const size_t lds_bytes = (size_t)PH * PW * (kch + 8) * 4;
    if (lds_bytes > 160 * 1024) return -6;

#define TMG_CONV_CASE(NTW_, WM_, WN_)                                                                  \
    if (NTW == NTW_ && WM == WM_ && WN == WN_) {                                                       \
        if (MT == 4) return launch_conv<4, NTW_, WM_, WN_>(p, gy, lds_bytes, st);                      \
        if (MT == 2) return launch_conv<2, NTW_, WM_, WN_>(p, gy, lds_bytes, st);                      \
        return launch_conv<1, NTW_, WM_, WN_>(p, gy, lds_bytes, st);                                   \
    }
    TMG_CONV_CASE(1, 4, 1)
    TMG_CONV_CASE(2, 4, 1)
    TMG_CONV_CASE(3, 4, 1)
    TMG_CONV_CASE(4, 4, 1)
    TMG_CONV_CASE(3, 2, 2)
    TMG_CONV_CASE(4, 2, 2)
    TMG_CONV_CASE(3, 1, 4)
    TMG_CONV_CASE(4, 1, 4)
#undef TMG_CONV_CASE
    return -7;
}

template <int NP, int NCO, bool LEAN>
static int launch_wgrad(const WgradP& p, dim3 grid, size_t lds_bytes, hipStream_t st) {
    TMG_LDS_OPTIN((&conv_wgrad_kernel<NP, NCO, LEAN>));
    const int nco_i = NCO == 1 ? 0 : (NCO == 2 ? 1 : 2);
    const int kid = NP == 3 ? 8 + nco_i : (NP == 8 ? 28 + nco_i : 19 + ((NP - 5) / 2) * 3 + nco_i);
    // (a grouped launch does the work of grid.y / bpg identically shaped contractions: rounds 2-5 booked ONE group's flops - the
    // '6.4 TFLOP/s' of conv_wgrad_kernel<9,1> in their bench lines was 1/15 of the first level's 0.87-ms launch)
    const double ngr = p.gtab ? (double)(grid.y / (p.bpg > 0 ? p.bpg : 1)) : 1.0;
    ProfScope prof(kid, ngr * 2.0 * p.B * p.Hout * p.Wout * (double)p.Cout * p.Cin * p.ksize * p.ksize, st);
    hipLaunchKernelGGL((conv_wgrad_kernel<NP, NCO, LEAN>), grid, dim3(512), lds_bytes, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

// TMG_WG_PLAN=MPIXMAX,GXMUL (0 = planner's choice): plan override of conv_wgrad_kernel for measurements (tools/bench_wgrad_groups.py)
static int tmg_wg_plan(int i) {
    static int v[2] = {-1, 0};
    if (v[0] < 0) {
        v[0] = 0;
        if (const char* e = getenv("TMG_WG_PLAN")) sscanf(e, "%d,%d", &v[0], &v[1]);
    }
    return v[i];
}

struct WgradPlan {
    int twl, TH, MPIX, tiles_x, tiles_y, ntiles, CITG, PPG, NCO, NP, ksplit, gx, gy, gz;
    size_t lds_bytes, ws_floats;
};

static int plan_wgrad_impl(int B, int Hout, int Wout, int ksize, int stride, int Cin, int Cout, WgradPlan* pl, bool allow4) {
    int twl = ilog2_ceil(Wout);
    if (twl > 5) twl = 5;
    if (twl < 1) twl = 1;
    pl->twl = twl;
    const int TW = 1 << twl;
    // 128-pixel tiles on large images (fewer barriers and less halo per staged byte), 64 otherwise
    pl->MPIX = (stride == 1 && (long)B * Hout * Wout >= 262144 && TW >= 16) ? 128 : 64;
    pl->TH = pl->MPIX >> twl;
    pl->tiles_x = (Wout + TW - 1) / TW;
    pl->tiles_y = (Hout + pl->TH - 1) / pl->TH;
    pl->ntiles = B * pl->tiles_x * pl->tiles_y;
    const int cit = ((Cin + 15) & ~15) >> 4, ntaps = ksize * ksize;
    const int cot = (Cout + 15) >> 4;
    // Decomposition: a block owns (CITG input-channel tiles x all taps) x (NCO output-channel tiles) of dW and a
    // strided share of the pixel tiles.  Prefer large register tiles (operand reuse); when the image is small,
    // fall back to smaller ones so that output groups x pixel shares still fill the chip with >= 4 tiles per block.
    static const int pref[][2] = {{4, 2}, {3, 2}, {2, 4}, {2, 2}, {4, 1}, {3, 1}, {1, 4}, {1, 2}, {2, 1}, {1, 1}};
    const int gmin = (1024 + pl->ntiles - 1) / pl->ntiles;
    const int halo = ksize >> 1;
    const int PW = stride * (TW - 1) + 1 + 2 * halo, PH = stride * (pl->TH - 1) + 1 + 2 * halo;
    // kernel limits: the patch of one tile fits the register window (7 float4 x 512 threads) and two LDS buffers fit
    auto fits = [&](int cg, int nc) {
        if (cg > 8) return false;                          // the lean staging path addresses at most 32 float4 slots per pixel
        const int k4p = cg <= 1 ? 4 : (cg <= 2 ? 8 : (cg <= 4 ? 16 : 32));  // float4 slots per pixel as the lean staging path pads them
        return PH * PW * k4p <= 7 * 512 && 2 * ((size_t)(PH * PW * 16 + 16) * cg + (size_t)(pl->MPIX * 16 + 16) * nc) * 4 <= 160 * 1024;
    };
    int CITG = 1, NCO = 1, bestg = -1;
    for (auto& c : pref) {
        int cg = c[0], nc = c[1];
        if (ntaps == 1) cg *= 4;  // 1x1: one pair per channel tile
        if (cg > cit) {
            // few input-channel tiles, many output tiles (the level-wide conditioning contractions: 2 x 15..): spend the register
            // tile on output channels - (2, 4) stages the same patch for twice the MFMA work of (2, 2)
            if (cit <= 2 && cot >= 8 && nc < 4 && ntaps == 9) continue;
            cg = cit;
        }
        if (nc > cot || (nc == 4 && !allow4)) continue;
        while (cg > 1 && !fits(cg, nc)) --cg;
        if (!fits(cg, nc)) continue;
        const int g = ((cit + cg - 1) / cg) * ((cot + nc - 1) / nc);
        if (g >= gmin) { CITG = cg; NCO = nc; bestg = g; break; }
        if (g > bestg) { CITG = cg; NCO = nc; bestg = g; }
    }
    if (bestg < 0) return -6;
    const int ngroups = (cit + CITG - 1) / CITG;
    pl->CITG = (cit + ngroups - 1) / ngroups;
    pl->NCO = NCO;
    // <= 9 pairs: every wave owns all of them and an eighth of each tile's pixels; otherwise the pairs are dealt to 4 waves
    pl->ksplit = ntaps * pl->CITG <= (NCO == 4 ? 5 : 9);  // (NP, NCO) = (7|9, 4) would not fit the register file
    pl->PPG = ntaps * pl->CITG;   // whole channel tiles per group ...
    if (!pl->ksplit && ngroups > 1) {
        // ... or equal ranges of the (channel tile, tap) pair list, rounded up to the 4 waves that share them, when that loads the
        // groups more evenly and a range never spans more channel tiles than fit (registers / LDS)
        const int tp = ntaps * cit, ppg = (((tp + ngroups - 1) / ngroups) + 3) & ~3;
        int span = 0;
        for (int g = 0; g * ppg < tp; ++g) {
            const int a = g * ppg, b = (a + ppg < tp ? a + ppg : tp) - 1;
            if (b / ntaps - a / ntaps + 1 > span) span = b / ntaps - a / ntaps + 1;
        }
        if (ppg < pl->PPG && (ngroups - 1) * ppg < tp && span <= 4 && fits(span, NCO)) { pl->PPG = ppg; pl->CITG = span; }
    }
    const int np = pl->ksplit ? pl->PPG : (pl->PPG + 3) / 4;
    pl->NP = np <= 3 ? 3 : (np <= 5 ? 5 : (np <= 7 ? 7 : (np <= 8 ? 8 : 9)));
    if (pl->NP == 8 && NCO == 4) pl->NP = 9;   // no (8, 4) instance
    if (np > 9) return -7;
    pl->gy = (cot + NCO - 1) / NCO;
    pl->gz = ngroups;
    int PHe = PH, PWe = PW;
    if (pl->ksplit && NCO == 1 && stride == 1 && TW >= 16 && pl->MPIX == 128) {
        // Narrow layers do little MFMA work per pixel: with 128-pixel tiles every wave walks ONE 16-pixel unit per tile and
        // the read/MFMA pipeline never fills.  Larger tiles (as far as registers, LDS and the tile count allow) give each
        // wave 2-4 units per tile and less halo per staged pixel.
        const int k4p = pl->CITG <= 1 ? 4 : (pl->CITG <= 2 ? 8 : (pl->CITG <= 4 ? 16 : 32));
        for (int mp = (tmg_wg_plan(0) >= 128 ? tmg_wg_plan(0) : 512); mp > 128; mp >>= 1) {
            const int th = mp >> twl, ph = th + 2 * halo;
            const int tiles = B * ((Wout + TW - 1) / TW) * ((Hout + th - 1) / th);
            if (ph * PW * k4p <= 7 * 512 && 2 * ((size_t)(ph * PW * 16 + 16) * pl->CITG + (size_t)(mp * 16 + 16) * NCO) * 4 <= 160 * 1024 &&
                tiles >= 4 * tmg_num_cus() / (pl->gy * pl->gz)) {
                pl->MPIX = mp; pl->TH = th;
                pl->tiles_y = (Hout + th - 1) / th;
                pl->ntiles = tiles;
                PHe = ph;
                break;
            }
        }
    }
    pl->lds_bytes = 2 * ((size_t)(PHe * PWe * 16 + 16) * pl->CITG + (size_t)(pl->MPIX * 16 + 16) * NCO) * 4;
    if (pl->lds_bytes < 8192) pl->lds_bytes = 8192;  // the dbias fold reuses the first 8 KB
    if (pl->lds_bytes < (size_t)4 * pl->NP * NCO * 1024) pl->lds_bytes = (size_t)4 * pl->NP * NCO * 1024;  // cross-wave fold of the partial sums
    // pixel shares: one 512-thread block per CU, but >= 4 tiles per block (two rounds fill the pipeline)
    int gx = tmg_num_cus() / (pl->gy * ngroups);
    if (gx > pl->ntiles / 4) gx = pl->ntiles / 4;
    if (gx < 1) gx = 1;
    pl->gx = gx;
    pl->ws_floats = (size_t)gx * pl->gy * pl->gz * (pl->ksplit ? 1 : 4) * pl->NP * NCO * 256 + (size_t)gx * pl->gy * 64;
    return 0;
}

// (NP, NCO) = (5, 4) needs 21 registers more than the file has (84 B of scratch per lane, rounds 3-5; outside config M's step): such a
// plan is made again without the four-output-tile register tiles.
static int plan_wgrad(int B, int Hout, int Wout, int ksize, int stride, int Cin, int Cout, WgradPlan* pl) {
    const int rc = plan_wgrad_impl(B, Hout, Wout, ksize, stride, Cin, Cout, pl, true);
    if (rc == 0 && pl->NP == 5 && pl->NCO == 4) return plan_wgrad_impl(B, Hout, Wout, ksize, stride, Cin, Cout, pl, false);
    return rc;
}

// Number of floats of scratch the slab path of tmg_conv_wgrad wants for these dims (same layout as tmg_conv_wgrad's dims).
extern "C" int64_t tmg_conv_wgrad_ws_floats(const int64_t* dims) {
    WgradPlan pl;
    if (plan_wgrad((int)dims[0], (int)dims[3], (int)dims[4], (int)dims[5], (int)dims[6], (int)dims[7], (int)dims[8], &pl) != 0) return 0;
    return (int64_t)pl.ws_floats;
}

// dims: [B,Hin,Win,Hout,Wout,ksize,stride,Cin,Cout,relu_in,pad_rep,cin_dst,cin_valid,ci_split,ci_off0,ci_off1]; dy_desc: [stride, off]
// dW is [Cout][cin_dst][k*k] (cin_dst = 0 -> Cin).  dW (and dbias) are ACCUMULATED onto (caller zero-fills).  ws: optional scratch of >= tmg_conv_wgrad_ws_floats(dims)
// floats; when given, per-block partial sums go through it and a small reduce kernel (few, low-contention atomics),
// otherwise every block adds its partial sums to dW with float atomics.
// Shared body of tmg_conv_wgrad (ngroups == 1, gtab == null) and tmg_conv_wgrad_grouped.
static int wgrad_impl(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* in_scale,
                      const void* in_shift, const void* dy, const int64_t* dy_desc, void* dW, void* dbias,
                      const void* kappa, void* ws, int64_t ws_floats, const int64_t* dims, hipStream_t st,
                      const long long* gtab, int ngroups, int dy_goff, long long dw_gstride, int db_gstride) {
    WgradP p;
    p.nseg = (int)nseg;
    p.vec4 = 1;
    fill_segs(p.in, in_ptrs, in_desc, (int)nseg, &p.vec4);
    p.B = (int)dims[0]; p.Hin = (int)dims[1]; p.Win = (int)dims[2]; p.Hout = (int)dims[3]; p.Wout = (int)dims[4];
    p.ksize = (int)dims[5]; p.stride = (int)dims[6]; p.Cin = (int)dims[7]; p.Cout = (int)dims[8];
    p.relu_in = (int)dims[9]; p.pad_rep = (int)dims[10];
    // dims[11..15] = {cin_dst, cin_valid, ci_split, ci_off0, ci_off1}; all zero -> dense [Cout][Cin][k*k]
    p.cin_dst = dims[11] > 0 ? (int)dims[11] : p.Cin;
    p.cin_valid = dims[12] > 0 ? (int)dims[12] : (p.cin_dst < p.Cin ? p.cin_dst : p.Cin);
    p.ci_split = dims[13] > 0 ? (int)dims[13] : 0x7fffffff;
    p.ci_off0 = (int)dims[14]; p.ci_off1 = (int)dims[15];
    if (p.ksize != 1 && p.ksize != 3) return -2;
    p.Cin_pad = (p.Cin + 15) & ~15;
    p.in_scale = (const float*)in_scale; p.in_shift = (const float*)in_shift;
    p.dy = TmgSeg{(const float*)dy, (int)dy_desc[0], (int)dy_desc[1], p.Cout};
    p.dy_vec4 = (((p.dy.stride | p.dy.off) & 3) == 0) && ((((uintptr_t)dy) & 15) == 0);
    p.dW = (float*)dW; p.dbias = (float*)dbias; p.kappa = (const float*)kappa;
    WgradPlan pl;
    const int rc = plan_wgrad(p.B, p.Hout, p.Wout, p.ksize, p.stride, p.Cin, p.Cout, &pl);
    if (rc != 0) return rc;
    p.TW_log2 = pl.twl; p.TH = pl.TH; p.MPIX = pl.MPIX; p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles; p.CITG = pl.CITG;
    p.PPG = pl.PPG;
    const int bpg = pl.gy;
    if (ngroups > 1) {
        // one launch for `ngroups` independent, identically shaped contractions: the groups share the pixel tiles' geometry,
        // blockIdx.y walks (group, output-channel block)
        // (blockIdx.z still walks the input-channel blocks of a group wider than one block)
        pl.gy = bpg * ngroups;
        int gx = tmg_num_cus() / (pl.gy * pl.gz);
        if (tmg_wg_plan(1) > 1) gx = tmg_wg_plan(1) * tmg_num_cus() / (pl.gy * pl.gz);
        if (gx > pl.ntiles / 4) gx = pl.ntiles / 4;
        if (gx < 1) gx = 1;
        pl.gx = gx;
        pl.ws_floats = (size_t)gx * pl.gy * pl.gz * (pl.ksplit ? 1 : 4) * pl.NP * pl.NCO * 256 + (size_t)gx * pl.gy * 64;
    }
    p.ws = (ws && (size_t)ws_floats >= pl.ws_floats && (((uintptr_t)ws) & 15) == 0) ? (float*)ws : nullptr;
    dim3 grid(pl.gx, pl.gy, pl.gz);
    p.ksplit = pl.ksplit;
    p.gtab = gtab; p.bpg = bpg; p.dy_goff = dy_goff; p.dw_gstride = dw_gstride; p.db_gstride = db_gstride;
    // lean staging needs float4-addressable operands and element offsets that fit the 24-bit multiplies / 32-bit adds
    p.fstage = p.vec4 && p.dy_vec4 && (p.Cout % 4 == 0) && (p.Cin % 4 == 0) && (long)p.Hin * p.Win < (1 << 24) && (long)p.Hout * p.Wout < (1 << 24);
    for (int i = 0; i < p.nseg; ++i)
        if (p.in[i].stride >= (1 << 24) || (long)p.B * p.Hin * p.Win * p.in[i].stride >= (1L << 31)) p.fstage = 0;
    if (p.dy.stride >= (1 << 24) || (long)p.B * p.Hout * p.Wout * p.dy.stride >= (1L << 31)) p.fstage = 0;
    p.dbg = 0;
    if (ngroups > 1 && (!p.fstage || !p.ws)) return -100;  // grouped launches exist only on the lean, slab-reduced path
    int lrc = -7;
#define TMG_WG_CASE(NP_, NCO_)                                                                    \
    if (pl.NP == NP_ && pl.NCO == NCO_)                                                           \
        lrc = p.fstage ? launch_wgrad<NP_, NCO_, true>(p, grid, pl.lds_bytes, st) : launch_wgrad<NP_, NCO_, false>(p, grid, pl.lds_bytes, st);
    TMG_WG_CASE(3, 1) TMG_WG_CASE(3, 2) TMG_WG_CASE(3, 4)
    TMG_WG_CASE(5, 1) TMG_WG_CASE(5, 2)
    TMG_WG_CASE(7, 1) TMG_WG_CASE(7, 2)
    TMG_WG_CASE(8, 1) TMG_WG_CASE(8, 2)
    TMG_WG_CASE(9, 1) TMG_WG_CASE(9, 2)
#undef TMG_WG_CASE
    if (lrc != 0) return lrc;
    if (p.ws) {
        const int items = pl.gy * pl.gz * (pl.ksplit ? 1 : 4) * pl.NP * pl.NCO * 64;
        int xchunk = 32;
        const int xc = (pl.gx + xchunk - 1) / xchunk;
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((items + 255) / 256, xc), dim3(256), 0, st, (const float*)p.ws, p.dW, p.dbias,
                           p.kappa, pl.gx, pl.gy, pl.gz, pl.NP, pl.NCO, pl.PPG, (p.Cin_pad >> 4) * p.ksize * p.ksize, p.cin_dst, p.Cout, p.ksize * p.ksize, xchunk,
                           p.cin_valid, p.ci_split, p.ci_off0, p.ci_off1, pl.ksplit, bpg, dw_gstride, db_gstride);
        TMG_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int tmg_conv_wgrad(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* in_scale,
                              const void* in_shift, const void* dy, const int64_t* dy_desc, void* dW, void* dbias,
                              const void* kappa, void* ws, int64_t ws_floats, const int64_t* dims, hipStream_t st) {
    return wgrad_impl(in_ptrs, in_desc, nseg, in_scale, in_shift, dy, dy_desc, dW, dbias, kappa, ws, ws_floats, dims, st, nullptr, 1, 0, 0, 0);
}

// `ngroups` identically shaped weight-gradient contractions in ONE launch (the per-layer coupling convolutions of a flow
// level: 15 launches of a few microseconds of MFMA work each otherwise).  in_ptrs / in_desc describe group 0 (geometry,
// alignment); gtab is a DEVICE table [ngroups][4][4] of int64: rows 0-2 {pointer, pixel stride, channel offset, channels} of the
// group's input segments, row 3 {dy pointer, dy pixel stride, 0, 0} or zeros.  Group g reads its own dy (row 3) or dy channels
// [g*gdims[0], +Cout) of the shared tensor, accumulates into dW + g*gdims[1] floats and dbias + g*gdims[2] floats.  Returns -100 when the shape cannot be grouped (caller issues per-group launches).
// ws must hold tmg_conv_wgrad_grouped_ws_floats(dims, ngroups) floats.
extern "C" int tmg_conv_wgrad_grouped(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* gtab,
                                      int64_t ngroups, const int64_t* gdims, const void* dy, const int64_t* dy_desc, void* dW,
                                      void* dbias, void* ws, int64_t ws_floats, const int64_t* dims, hipStream_t st) {
    if (ngroups < 1 || !gtab) return -3;
    return wgrad_impl(in_ptrs, in_desc, nseg, nullptr, nullptr, dy, dy_desc, dW, dbias, nullptr, ws, ws_floats, dims, st,
                      (const long long*)gtab, (int)ngroups, (int)gdims[0], (long long)gdims[1], (int)gdims[2]);
}

extern "C" int64_t tmg_conv_wgrad_grouped_ws_floats(const int64_t* dims, int64_t ngroups) {
    WgradPlan pl;
    if (plan_wgrad((int)dims[0], (int)dims[3], (int)dims[4], (int)dims[5], (int)dims[6], (int)dims[7], (int)dims[8], &pl) != 0) return 0;
    const int gy = pl.gy * (int)ngroups;
    int gx = tmg_num_cus() / (gy * pl.gz);
    if (tmg_wg_plan(1) > 1) gx = tmg_wg_plan(1) * tmg_num_cus() / (gy * pl.gz);
    if (gx > pl.ntiles / 4) gx = pl.ntiles / 4;
    if (gx < 1) gx = 1;
    return (int64_t)((size_t)gx * gy * pl.gz * (pl.ksplit ? 1 : 4) * pl.NP * pl.NCO * 256 + (size_t)gx * gy * 64);
}

// Replicate-padding fold for the 3x3 input gradient (see conv_rep_border_fix_kernel).
// dims: [B,H,W,Cdy,Cx]; dy_desc: [stride, off]; w: the mode-1 packed weights of the SAME conv (tmg_conv_pack)
extern "C" int tmg_conv_rep_border_fix(const void* dy, const int64_t* dy_desc, const void* w, const void* kappa,
                                       void* const* out_ptrs, const int64_t* out_desc, int64_t nout, const int64_t* dims,
                                       hipStream_t st) {
    BorderP p;
    p.dy = (const float*)dy; p.dy_stride = (int)dy_desc[0]; p.dy_off = (int)dy_desc[1];
    p.B = (int)dims[0]; p.H = (int)dims[1]; p.W = (int)dims[2]; p.Cdy = (int)dims[3]; p.Cx = (int)dims[4];
    p.wpk = (const float*)w; p.kappa = (const float*)kappa;
    p.Npad = (p.Cx + 15) & ~15;
    p.KB = (p.Cdy + 15) >> 4;
    p.dy_vec = ((p.Cdy & 15) == 0) && (((p.dy_stride | p.dy_off) & 3) == 0) && ((((uintptr_t)dy) & 15) == 0);
    for (int i = 0; i < TMG_MAX_OUT_SEG; ++i) p.out[i] = TmgOSeg{nullptr, 0, 0, 0};
    for (int i = 0; i < (int)nout; ++i)
        p.out[i] = TmgOSeg{(float*)out_ptrs[i], (int)out_desc[3 * i], (int)out_desc[3 * i + 1], (int)out_desc[3 * i + 2]};
    if (p.H == 1 && p.W == 1) p.nborder = 1;
    else if (p.H == 1) p.nborder = p.W;
    else p.nborder = 2 * p.W + (p.W > 1 ? 2 : 1) * (p.H - 2);
    // matrix-core path: images with an interior (H, W >= 2), float4-addressable dy and dx segments, <= 8 channel tiles
    bool mf = p.H >= 2 && p.W >= 2 && (p.Cdy & 3) == 0 && (((p.dy_stride | p.dy_off) & 3) == 0) && ((((uintptr_t)dy) & 15) == 0) &&
              (p.Cx & 3) == 0 && p.Npad <= 128;
    for (int i = 0; i < (int)nout; ++i)
        if (((p.out[i].stride | p.out[i].off | p.out[i].n) & 3) || (((uintptr_t)p.out[i].p) & 15)) mf = false;
    if (mf) {
        BorderMP m;
        m.b = p;
        const int cnt[8] = {p.H - 2, p.H - 2, p.W - 2, p.W - 2, 1, 1, 1, 1};
        int t0 = 0;
        for (int v = 0; v < 16; ++v) {
            const int c = v < 4 ? v : 4 + (v - 4) / 3;
            m.cnt[c] = cnt[c] > 0 ? cnt[c] : 1;
            m.tile0[v] = t0;
            t0 += cnt[c] > 0 ? (p.B * cnt[c] + 15) / 16 : 0;
        }
        m.tile0[16] = t0;
        // enough waves to fill the chip: split the dy channels of a tile over several waves when there are few tiles
        int S = (4096 + t0 - 1) / t0;
        if (S > p.KB / 4) S = p.KB / 4;
        if (S < 1) S = 1;
        m.ksplit = S;
        const long items = (long)t0 * S;
        const int blocks = (int)((items + 3) / 4 < 4096 ? (items + 3) / 4 : 4096);
        switch (p.Npad >> 4) {
            case 1: hipLaunchKernelGGL(conv_rep_border_mfma_kernel<1>, dim3(blocks), dim3(256), 0, st, m); break;
            case 2: hipLaunchKernelGGL(conv_rep_border_mfma_kernel<2>, dim3(blocks), dim3(256), 0, st, m); break;
            case 3: hipLaunchKernelGGL(conv_rep_border_mfma_kernel<3>, dim3(blocks), dim3(256), 0, st, m); break;
            case 4: hipLaunchKernelGGL(conv_rep_border_mfma_kernel<4>, dim3(blocks), dim3(256), 0, st, m); break;
            case 5: hipLaunchKernelGGL(conv_rep_border_mfma_kernel<5>, dim3(blocks), dim3(256), 0, st, m); break;
            case 6: hipLaunchKernelGGL(conv_rep_border_mfma_kernel<6>, dim3(blocks), dim3(256), 0, st, m); break;
            case 7: hipLaunchKernelGGL(conv_rep_border_mfma_kernel<7>, dim3(blocks), dim3(256), 0, st, m); break;
            default: hipLaunchKernelGGL(conv_rep_border_mfma_kernel<8>, dim3(blocks), dim3(256), 0, st, m); break;
        }
        TMG_CHECK_LAUNCH();
        return 0;
    }
    const size_t total = (size_t)p.B * p.nborder * p.Cx;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(conv_rep_border_fix_kernel, dim3(blocks), dim3(256), 0, st, p);
    TMG_CHECK_LAUNCH();
    return 0;
}

// dims: [B,Hin,Win,Hout,Wout,Cin,Cout,ksize,stride,accumulate]; dy_desc/dx_desc: [stride, off]
extern "C" int tmg_conv_dgrad_direct(const void* dy, const int64_t* dy_desc, const void* w, void* dx, const int64_t* dx_desc,
                                     const int64_t* dims, hipStream_t st) {
    const size_t total = (size_t)dims[0] * dims[1] * dims[2] * dims[5];
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(conv_dgrad_direct_kernel, dim3(blocks), dim3(256), 0, st, (const float*)dy, (int)dy_desc[0], (int)dy_desc[1],
                       (const float*)w, (float*)dx, (int)dx_desc[0], (int)dx_desc[1], (int)dims[0], (int)dims[1], (int)dims[2],
                       (int)dims[3], (int)dims[4], (int)dims[5], (int)dims[6], (int)dims[7], (int)dims[8], (int)dims[9]);
    TMG_CHECK_LAUNCH();
    return 0;
}
