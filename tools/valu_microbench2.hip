// VALU issue rate: scalar f32 ops (v_add/v_mul/v_fma_f32) against packed ones (v_pk_add/mul/fma_f32 on float2),
// 1..4 waves per SIMD, independent chains.  Compile the scalar variants with -fno-slp-vectorize:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/bin/valu_mb2.bin tools/valu_microbench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(1024) void k_scalar(float* out, int iters, float a, float b) {
    float x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (KIND == 0) x[i] = fmaf(x[i], a, b);
                else if (KIND == 1) x[i] = x[i] + a;
                else x[i] = x[i] * a;
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
__global__ __launch_bounds__(1024) void k_packed(float* out, int iters, float a, float b) {
    f2 x[16];
    const f2 av = {a, a * 1.0001f}, bv = {b, b * 0.5f};
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (f2){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) x[i] = __builtin_elementwise_fma(x[i], av, bv);
                else if (KIND == 1) x[i] = x[i] + av;
                else x[i] = x[i] * av;
            }
    }
    f2 s = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <typename K>
void run(const char* name, K kern, float* out, int w, int iters, int instr_per_iter) {
    const int threads = 64 * 4 * w;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.0001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)iters * instr_per_iter;
    printf("%-14s %d waves/SIMD: %8.1f us  %.2f ns per instr per SIMD (= %.2f cycles at 2.4 GHz)\n", name, w, ms * 1e3,
           ms * 1e6 / (instr * w), ms * 1e-3 * 2.4e9 / (instr * w));
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    for (int w = 1; w <= 4; ++w) run("scalar fma", k_scalar<0>, out, w, 4000, 128);
    for (int w = 1; w <= 4; ++w) run("scalar add", k_scalar<1>, out, w, 4000, 128);
    for (int w = 1; w <= 4; ++w) run("scalar mul", k_scalar<2>, out, w, 4000, 128);
    for (int w = 1; w <= 4; ++w) run("packed fma", k_packed<0>, out, w, 4000, 64);
    for (int w = 1; w <= 4; ++w) run("packed add", k_packed<1>, out, w, 4000, 64);
    for (int w = 1; w <= 4; ++w) run("packed mul", k_packed<2>, out, w, 4000, 64);
    return 0;
}
