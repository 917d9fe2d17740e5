// Lane layout of v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products per instruction) and its issue rate.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma4x4.hip -o /tmp/mfma4x4 && /tmp/mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

__global__ void rate(float* out, int iters) {
    f32x4 acc[8];
    for (int s = 0; s < 8; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[s] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[s], 0, 0, 0);
    }
    float v = 0.f;
    for (int s = 0; s < 8; ++s) v += acc[s][0] + acc[s][1] + acc[s][2] + acc[s][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

int main() {
    std::vector<float> a(64), b(64), d(256);
    for (int l = 0; l < 64; ++l) { a[l] = 1.f + l; b[l] = 100.f + l; }
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    // hypothesis: D[lane l][reg r] = A[lane 4 (l / 4) + r] * B[lane l]   (block = l / 4, row i = r, column j = l % 4)
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const float e = a[4 * (l / 4) + r] * b[l];
            if (d[l * 4 + r] != e) ok = 0;
        }
    printf("hypothesis D[l][r] = A[4(l/4)+r] * B[l]: %s\n", ok ? "TRUE" : "false");
    if (!ok) {
        for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, d[l * 4], d[l * 4 + 1], d[l * 4 + 2], d[l * 4 + 3]);
        for (int l = 60; l < 64; ++l) printf("lane %d: %g %g %g %g\n", l, d[l * 4], d[l * 4 + 1], d[l * 4 + 2], d[l * 4 + 3]);
    }
    float* out; hipMalloc(&out, 1024 * 256 * 4 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate, dim3(1024), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate, dim3(1024), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = 1024.0 * 4 * iters * 8;     // MFMA instructions (per wave) in total
    printf("4x4x1: %.3f ms, %.2f T instr/s, %.1f TFLOP/s (512 flop / instr), cycles per instr per SIMD at 2.4 GHz: %.2f\n", ms, n / ms / 1e9,
           n * 512 / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / n);
    return 0;
}
