// Does the sustained fp32 MFMA rate depend on the operand data (power management)?  Same instruction stream,
// operands either constant-ish or pseudo-random full-mantissa values that change every instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
    f32x4 acc[18];
    for (int i = 0; i < 18; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a[9], b[2], sh[9];
    for (int i = 0; i < 9; ++i) a[i] = MODE ? in[(threadIdx.x * 9 + i) & 4095] : 1e-3f * threadIdx.x;
    for (int i = 0; i < 9; ++i) sh[i] = a[i];
    for (int i = 0; i < 2; ++i) b[i] = MODE ? in[(threadIdx.x * 2 + i + 2048) & 4095] : 2e-3f * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[j * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[n], acc[j * 2 + n], 0, 0, 0);
        if (MODE == 3) {  // same VALU work as mode 2 on shadow registers: the MFMA operands stay fixed
#pragma unroll
            for (int j = 0; j < 9; ++j) sh[j] = __int_as_float((__float_as_int(sh[j]) * 1664525 + 1013904223) & 0x3FFFFFFF | 0x30000000);
        }
        if (MODE == 2) {  // new operand values every iteration (cheap integer scramble, stays finite)
#pragma unroll
            for (int j = 0; j < 9; ++j) a[j] = __int_as_float((__float_as_int(a[j]) * 1664525 + 1013904223) & 0x3FFFFFFF | 0x30000000);
        }
    }
    float s = 0;
    for (int i = 0; i < 9; ++i) s += sh[i];
    for (int i = 0; i < 18; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *out, *in;
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&in, 4096 * 4);
    float h[4096];
    unsigned x = 12345;
    for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 512;
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d (%s): %.3f ms  %.1f TFLOP/s\n", mode, mode == 0 ? "smooth operands" : mode == 1 ? "random operands, fixed" : mode == 2 ? "random operands, changing" : "random fixed operands + shadow VALU",
                   ms / 5, 4.0 * 18 * 2048 * (double)blocks * iters / (ms / 5) / 1e9);
        }
    return 0;
}
