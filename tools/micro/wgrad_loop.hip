// Inner-loop laboratory for conv_wgrad_kernel<9,2>: LDS-resident operands, no staging.  Variants by -DVAR=n.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef VAR
#define VAR 0
#endif
constexpr int NP = 9, NCO = 2, TWl = 5, TW = 32, MPIX = 128, PW = 34, PH = 6, CS = 64, DS = 32;

__global__ __launch_bounds__(256) void loop_kernel(float* out, int ntiles, int rTWl, int rPW, int rPH, int rMPIX, int rcitn) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    float* ldy = lds + PH * PW * CS;
    for (int i = tid; i < PH * PW * CS + MPIX * DS; i += 256) lds[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    f32x4 acc[NP][NCO];
    for (int j = 0; j < NP; ++j)
        for (int n = 0; n < NCO; ++n) acc[j][n] = (f32x4){0, 0, 0, 0};
    int aoff[NP];
    for (int j = 0; j < NP; ++j) {
        const int pid = wave + 4 * j;
        const int tap = pid / 4, cit = pid - tap * 4;
        const int tyy = tap / 3, txx = tap - tyy * 3;
        aoff[j] = (tyy * PW + txx) * CS + ((cit * 16) ^ (((q + txx) & 1) ? 16 : 0));
    }
    const int nks = MPIX / 4;
    for (int tile = 0; tile < ntiles; ++tile) {
        __syncthreads();
        float av[NP], bfr[NCO], avn[NP], bfn[NCO];
#define LOAD(AV, BF, KS)                                                                     \
        {                                                                                    \
            const int m_ = (KS) * 4 + q;                                                     \
            const float* ab_ = lds + ((m_ >> TWl) * PW + (m_ & (TW - 1))) * CS + li;         \
            _Pragma("unroll") for (int n = 0; n < NCO; ++n) BF[n] = ldy[m_ * DS + ((n * 16) ^ ((q & 1) ? 16 : 0)) + li]; \
            _Pragma("unroll") for (int j = 0; j < NP; ++j) AV[j] = ab_[aoff[j]];             \
        }
#define MFMA(AV, BF)                                                                         \
        _Pragma("unroll") for (int j = 0; j < NP; ++j) {                                     \
            _Pragma("unroll") for (int n = 0; n < NCO; ++n)                                  \
                acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[j], BF[n], acc[j][n], 0, 0, 0); \
        }
#if VAR == 0
        LOAD(av, bfr, 0)
        for (int ks = 0; ks < nks; ks += 2) {
            LOAD(avn, bfn, ks + 1)
            MFMA(av, bfr)
            LOAD(av, bfr, min(ks + 2, nks - 1))
            MFMA(avn, bfn)
        }
#elif VAR == 1
        // interleave hints: 1 LDS read per 2 MFMAs
        LOAD(av, bfr, 0)
        for (int ks = 0; ks < nks; ks += 2) {
            LOAD(avn, bfn, ks + 1)
            MFMA(av, bfr)
#pragma unroll
            for (int g = 0; g < 11; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
            LOAD(av, bfr, min(ks + 2, nks - 1))
            MFMA(avn, bfn)
#pragma unroll
            for (int g = 0; g < 11; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
        }
#elif VAR == 2
        // no LDS reads in the loop at all: pure MFMA + loop overhead (upper bound)
        LOAD(av, bfr, 0)
        for (int ks = 0; ks < nks; ks += 2) {
            MFMA(av, bfr)
            MFMA(av, bfr)
        }
#elif VAR == 3
        // precomputed pixel offsets in a row: row-major walk so addresses advance by constants
        // k-step ks covers pixels 4ks..4ks+3 of the 4x32 tile: row = ks>>3, col = (ks&7)*4+q
        LOAD(av, bfr, 0)
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                const int ks = r * 8 + c;
                LOAD(avn, bfn, ks + 1)
                MFMA(av, bfr)
                LOAD(av, bfr, min(ks + 2, nks - 1))
                MFMA(avn, bfn)
            }
        }
#elif VAR == 4
        // channel-tile-major layout [cit][py][px][16] with runtime geometry: pixel stride is 16 words for every
        // configuration, so the four k-steps of a 16-pixel unit are immediates; only the unit base moves.
        {
            const int TWr = 1 << rTWl, plane = rPH * rPW * 16;
            int ao[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int pid = wave + 4 * j;
                const int tap = pid / rcitn, cit = pid - tap * rcitn;
                const int tyy = tap / 3, txx = tap - tyy * 3;
                ao[j] = cit * plane + (tyy * rPW + txx) * 16 + q * 16 + li;
            }
            const float* ldyr = lds + rcitn * plane;
            const int bo = q * 16 + li;
            const int nun = rMPIX >> 4;
#define LOADU(AV, BF, U, K)                                                                                   \
            {                                                                                                 \
                const int p0_ = (U) * 16;                                                                     \
                const float* ab_ = lds + (((p0_ >> rTWl) * rPW + (p0_ & (TWr - 1))) * 16) + (K) * 64;         \
                const float* bb_ = ldyr + p0_ * 16 + (K) * 64;                                                \
                _Pragma("unroll") for (int n = 0; n < NCO; ++n) BF[n] = bb_[bo + n * rMPIX * 16];             \
                _Pragma("unroll") for (int j = 0; j < NP; ++j) AV[j] = ab_[ao[j]];                            \
            }
            LOADU(av, bfr, 0, 0)
            for (int u = 0; u < nun; ++u) {
                LOADU(avn, bfn, u, 1)
                MFMA(av, bfr)
                LOADU(av, bfr, u, 2)
                MFMA(avn, bfn)
                LOADU(avn, bfn, u, 3)
                MFMA(av, bfr)
                LOADU(av, bfr, min(u + 1, nun - 1), 0)
                MFMA(avn, bfn)
            }
        }
#elif VAR == 5
        // VAR 0 with runtime geometry (what the production kernel does today)
        {
            const int TWr = 1 << rTWl, CSr = rcitn * 16;
            int ao[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int pid = wave + 4 * j;
                const int tap = pid / rcitn, cit = pid - tap * rcitn;
                const int tyy = tap / 3, txx = tap - tyy * 3;
                ao[j] = (tyy * rPW + txx) * CSr + ((cit * 16) ^ (((q + txx) & 1) ? 16 : 0));
            }
            const float* ldyr = lds + rPH * rPW * CSr;
            const int nksr = rMPIX / 4;
#define LOADR(AV, BF, KS)                                                                    \
            {                                                                                \
                const int m_ = (KS) * 4 + q;                                                 \
                const float* ab_ = lds + ((m_ >> rTWl) * rPW + (m_ & (TWr - 1))) * CSr + li; \
                _Pragma("unroll") for (int n = 0; n < NCO; ++n) BF[n] = ldyr[m_ * DS + ((n * 16) ^ ((q & 1) ? 16 : 0)) + li]; \
                _Pragma("unroll") for (int j = 0; j < NP; ++j) AV[j] = ab_[ao[j]];           \
            }
            LOADR(av, bfr, 0)
            for (int ks = 0; ks < nksr; ks += 2) {
                LOADR(avn, bfn, ks + 1)
                MFMA(av, bfr)
                LOADR(av, bfr, min(ks + 2, nksr - 1))
                MFMA(avn, bfn)
            }
        }
#elif VAR == 6
        // channel-tile-major layout [cit][py][px][16] with runtime geometry: pixel stride is 16 words for every
        // configuration, so the four k-steps of a 16-pixel unit are immediates; only the unit base moves.
        {
            const int TWr = 1 << rTWl, plane = rPH * rPW * 16;
            int ao[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int pid = wave + 4 * j;
                const int tap = pid / rcitn, cit = pid - tap * rcitn;
                const int tyy = tap / 3, txx = tap - tyy * 3;
                ao[j] = cit * plane + (tyy * rPW + txx) * 16 + q * 16 + li;
            }
            const float* ldyr = lds + rcitn * plane;
            const int bo = q * 16 + li;
            const int nun = rMPIX >> 4;
#undef LOADU
#define LOADU(AV, BF, U, K)                                                                                   \
            {                                                                                                 \
                const int p0_ = (U) * 16;                                                                     \
                const float* ab_ = lds + (((p0_ >> rTWl) * rPW + (p0_ & (TWr - 1))) * 16) + (K) * 64;         \
                const float* bb_ = ldyr + p0_ * 16 + (K) * 64;                                                \
                _Pragma("unroll") for (int n = 0; n < NCO; ++n) BF[n] = bb_[bo + n * rMPIX * 16];             \
                _Pragma("unroll") for (int j = 0; j < NP; ++j) AV[j] = ab_[ao[j]];                            \
            }
            LOADU(av, bfr, 0, 0)
            for (int u = 0; u < nun; ++u) {
#define SB __builtin_amdgcn_sched_barrier(0);
                LOADU(avn, bfn, u, 1) SB
                MFMA(av, bfr) SB
                LOADU(av, bfr, u, 2) SB
                MFMA(avn, bfn) SB
                LOADU(avn, bfn, u, 3) SB
                MFMA(av, bfr) SB
                LOADU(av, bfr, min(u + 1, nun - 1), 0) SB
                MFMA(avn, bfn) SB
            }
        }
#elif VAR == 7
        // VAR 0 with runtime geometry (what the production kernel does today)
        {
            const int TWr = 1 << rTWl, CSr = rcitn * 16;
            int ao[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int pid = wave + 4 * j;
                const int tap = pid / rcitn, cit = pid - tap * rcitn;
                const int tyy = tap / 3, txx = tap - tyy * 3;
                ao[j] = (tyy * rPW + txx) * CSr + ((cit * 16) ^ (((q + txx) & 1) ? 16 : 0));
            }
            const float* ldyr = lds + rPH * rPW * CSr;
            const int nksr = rMPIX / 4;
#undef LOADR
#define LOADR(AV, BF, KS)                                                                    \
            {                                                                                \
                const int m_ = (KS) * 4 + q;                                                 \
                const float* ab_ = lds + ((m_ >> rTWl) * rPW + (m_ & (TWr - 1))) * CSr + li; \
                _Pragma("unroll") for (int n = 0; n < NCO; ++n) BF[n] = ldyr[m_ * DS + ((n * 16) ^ ((q & 1) ? 16 : 0)) + li]; \
                _Pragma("unroll") for (int j = 0; j < NP; ++j) AV[j] = ab_[ao[j]];           \
            }
            LOADR(av, bfr, 0)
            for (int ks = 0; ks < nksr; ks += 2) {
                LOADR(avn, bfn, ks + 1) __builtin_amdgcn_sched_barrier(0);
                MFMA(av, bfr) __builtin_amdgcn_sched_barrier(0);
                LOADR(av, bfr, min(ks + 2, nksr - 1)) __builtin_amdgcn_sched_barrier(0);
                MFMA(avn, bfn) __builtin_amdgcn_sched_barrier(0);
            }
        }
#elif VAR == 8
        // VAR 6 with explicit LDS byte addresses bumped in place once per 16-pixel unit; every read uses an immediate
        {
            typedef const __attribute__((address_space(3))) float* lptr;
            const int TWr = 1 << rTWl, plane = rPH * rPW * 16;
            unsigned aa[NP], ba[NCO];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int pid = wave + 4 * j;
                const int tap = pid / rcitn, cit = pid - tap * rcitn;
                const int tyy = tap / 3, txx = tap - tyy * 3;
                aa[j] = 4u * (cit * plane + (tyy * rPW + txx) * 16 + q * 16 + li);
            }
#pragma unroll
            for (int n = 0; n < NCO; ++n) ba[n] = 4u * (rcitn * plane + n * rMPIX * 16 + q * 16 + li);
            const int nun = rMPIX >> 4;
#define LD(AV, BF, K)                                                                                           \
            {                                                                                                   \
                _Pragma("unroll") for (int n = 0; n < NCO; ++n) BF[n] = *(lptr)(uintptr_t)(ba[n] + (K) * 256);  \
                _Pragma("unroll") for (int j = 0; j < NP; ++j) AV[j] = *(lptr)(uintptr_t)(aa[j] + (K) * 256);   \
            }
#define SB __builtin_amdgcn_sched_barrier(0);
            LD(av, bfr, 0)
            for (int u = 0; u < nun; ++u) {
                LD(avn, bfn, 1) SB
                MFMA(av, bfr) SB
                LD(av, bfr, 2) SB
                MFMA(avn, bfn) SB
                LD(avn, bfn, 3) SB
                MFMA(av, bfr) SB
                {
                    const int un = min(u + 1, nun - 1);
                    const int p0 = u * 16, p1 = un * 16;
                    const int d = (((p1 >> rTWl) * rPW + (p1 & (TWr - 1))) - ((p0 >> rTWl) * rPW + (p0 & (TWr - 1)))) * 64;
                    const int db = (p1 - p0) * 64;
#pragma unroll
                    for (int j = 0; j < NP; ++j) aa[j] += d;
#pragma unroll
                    for (int n = 0; n < NCO; ++n) ba[n] += db;
                }
                LD(av, bfr, 0) SB
                MFMA(avn, bfn) SB
            }
        }
#endif
    }
    float s = 0;
    for (int j = 0; j < NP; ++j)
        for (int n = 0; n < NCO; ++n) s += acc[j][n][0] + acc[j][n][1] + acc[j][n][2] + acc[j][n][3];
    out[blockIdx.x * 256 + tid] = s;
}
int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * 4);
    const size_t ldsb = (PH * PW * CS + MPIX * DS) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512}) {
        const int ntiles = 256;
        hipLaunchKernelGGL(loop_kernel, dim3(blocks), dim3(256), ldsb, 0, out, ntiles, 5, 34, 6, 128, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop_kernel, dim3(blocks), dim3(256), ldsb, 0, out, ntiles, 5, 34, 6, 128, 4);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * ntiles * 4 * NP * NCO * (MPIX / 4) * 2048.0;
        printf("VAR %d blocks %d: %.3f ms  %.1f TFLOP/s (lds %zu B)\n", VAR, blocks, ms, fl / ms / 1e9, ldsb);
    }
    return 0;
}
