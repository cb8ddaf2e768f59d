// VALU issue-rate micro benchmark: independent v_fma_f32 / v_add_f32 chains, 1..4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/valu_mb.bin tools/valu_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(1024) void valu_kernel(float* out, int iters, float a, float b) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) x[i] = fmaf(x[i], a, b);
                else if (KIND == 1) x[i] = x[i] + a;
                else x[i] = x[i] * a;
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(float* out, int waves_per_simd, int iters) {
    const int threads = 64 * 4 * waves_per_simd;   // one workgroup per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(valu_kernel<KIND>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(valu_kernel<KIND>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 128;
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("kind %d, %d waves/SIMD: %.1f us; %.2f cycles per VALU instr per SIMD (at 2.4 GHz), %.2f per wave\n", KIND,
           waves_per_simd, ms * 1e3, cyc / (instr_per_wave * waves_per_simd), cyc / instr_per_wave);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    for (int w = 1; w <= 4; ++w) run<0>(out, w, 4000);
    for (int w = 1; w <= 4; ++w) run<1>(out, w, 4000);
    for (int w = 1; w <= 4; ++w) run<2>(out, w, 4000);
    return 0;
}
