// What HBM gives for Griffin-Lim's byte mix (12 B read + 8 B written per element) with the simplest possible access
// pattern: a large grid, every thread streams 16-byte vectors, nothing else.  The yardstick for
// tools/stream_microbench.hip (one spectrum row per wave, persistent workgroups) and for gl_iter_kernel itself.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/hbm_mix.bin tools/hbm_mix_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(256) void mix_kernel(const f4* __restrict__ x, const f4* __restrict__ m, f4* __restrict__ y, size_t n4, int persistent) {
    // element i4: 16 B of |S| (4 bins) pair with 32 B of X in and 32 B of X out
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f4 a = x[2 * i], b = x[2 * i + 1];
        const f4 g = NT ? __builtin_nontemporal_load(m + i) : m[i];
        f4 o0 = a * g.x + b, o1 = b * g.y + a;
        if (NT) { __builtin_nontemporal_store(o0, y + 2 * i); __builtin_nontemporal_store(o1, y + 2 * i + 1); }
        else { y[2 * i] = o0; y[2 * i + 1] = o1; }
        if (!persistent) break;
    }
}
int main() {
    const size_t bins = (size_t)64 * 1025 * 1000, n4 = bins / 4;
    f4 *x, *m, *y;
    (void)hipMalloc(&x, bins * 8); (void)hipMalloc(&m, bins * 4); (void)hipMalloc(&y, bins * 8);
    (void)hipMemset(x, 0x11, bins * 8); (void)hipMemset(m, 0x22, bins * 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto run = [&](int nt, int grid, int persistent, const char* what) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            (void)hipEventRecord(a);
            if (nt) hipLaunchKernelGGL(mix_kernel<1>, dim3(grid), dim3(256), 0, 0, x, m, y, n4, persistent);
            else hipLaunchKernelGGL(mix_kernel<0>, dim3(grid), dim3(256), 0, 0, x, m, y, n4, persistent);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, a, b);
            if (rep) best = ms < best ? ms : best;
        }
        printf("%-44s nt=%d: %7.1f us  %.2f TB/s (20 B per bin)\n", what, nt, best * 1e3, 20.0 * bins / (best * 1e-3) / 1e12);
    };
    for (int nt = 0; nt < 2; ++nt) {
        run(nt, (int)((n4 + 255) / 256), 0, "one 16-byte group per thread, 64 K workgroups");
        run(nt, 256 * 8, 1, "grid-stride, 2048 workgroups");
        run(nt, 256 * 2, 1, "grid-stride, 512 workgroups");
        run(nt, 224 * 2, 1, "grid-stride, 448 workgroups");
    }
    return 0;
}
