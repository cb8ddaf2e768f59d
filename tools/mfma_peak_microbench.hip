// Sustained fp32 MFMA rate of the chip with NO memory traffic: every wave issues independent v_mfma_f32_32x32x2_f32
// back to back (4 accumulator tiles).  What the GEMM kernels can be priced against under the clocks a long MFMA
// burst actually gets.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/mfma_peak.bin tools/mfma_peak_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 4096 * sizeof(float));
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu)
        for (int iters : {2000, 20000}) {
            const int grid = 256 * wgs_per_cu;
            hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, 100, 1.0f, 0.5f);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, iters, 1.234567f, 0.7654321f);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, a, b);
            const double flop = 2.0 * 32 * 32 * 2 * 16.0 * iters * 4 * grid;   // 16 MFMAs per iteration per wave, 4 waves
            printf("%d workgroup(s) per CU, %5d iterations: %.3f ms, %.1f TFLOP/s\n", wgs_per_cu, iters, ms, flop / (ms * 1e-3) / 1e12);
        }
    return 0;
}
