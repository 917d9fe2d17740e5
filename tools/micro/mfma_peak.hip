// Calibration: sustained fp32 MFMA rate of the chip with operands in registers (no memory traffic).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static void run(const char* name, F launch, double flop_per_block_iter, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(blocks, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(blocks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks %5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flop_per_block_iter * blocks * iters / ms / 1e9);
}
int main() {
    float* out;
    hipMalloc(&out, 8192 * 256 * 4);
    const int iters = 20000;
    for (int blocks : {256, 512, 1024, 2048}) {
        run("16x16x4 f32, 8 acc", [&](int b, int it) { hipLaunchKernelGGL(k16<8>, dim3(b), dim3(256), 0, 0, out, it); }, 4.0 * 8 * 2048, blocks, iters);
        run("16x16x4 f32, 18 acc", [&](int b, int it) { hipLaunchKernelGGL(k16<18>, dim3(b), dim3(256), 0, 0, out, it / 2); }, 4.0 * 18 * 2048 / 2, blocks, iters);
        run("32x32x2 f32, 4 acc", [&](int b, int it) { hipLaunchKernelGGL(k32<4>, dim3(b), dim3(256), 0, 0, out, it); }, 4.0 * 4 * 4096, blocks, iters);
    }
    return 0;
}
