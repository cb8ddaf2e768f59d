// The hop of tools/cluster_sync_microbench.hip (clusters of workgroups handing a slab to each other through sc1
// stores / an agent-scope counter / sc1 loads) priced UNDER MEMORY LOAD: a streaming kernel shaped like
// gl_iter_kernel's traffic (224 persistent 512-thread workgroups, 20 B per element, plain loads + non-temporal
// stores) runs on a second stream for the whole measurement.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/cluster_sync_loaded.bin tools/cluster_sync_loaded_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void stream_load_kernel(const f2* __restrict__ in, const float* __restrict__ mag, f2* __restrict__ out,
                                                          size_t n, int passes) {
    const size_t stride = (size_t)gridDim.x * 512;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n; i += stride) {
            f2 v = in[i];
            const float m = __builtin_nontemporal_load(mag + i);
            v.x *= m; v.y *= m;
            __builtin_nontemporal_store(v, out + i);
        }
}

template <int SLAB_FLOATS, int THREADS>
__global__ __launch_bounds__(THREADS) void hop_kernel_wide(float* slabs, unsigned* counters, unsigned* timeout, int W, int iters, float* sink) {
    const int wg = blockIdx.x, c = wg / W, tid = threadIdx.x;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    unsigned* cnt = counters + 64 * c;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, (int)((size_t)2 * gridDim.x * SLAB_FLOATS * 4), 0x00020000);
    for (int it = 0; it < iters; ++it) {
        const int mine = ((it & 1) * gridDim.x + wg) * SLAB_FLOATS * 4;
        for (int i = tid * 4; i < SLAB_FLOATS; i += THREADS * 4)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, acc), rs, mine + i * 4, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)W * (unsigned)(it + 1);
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000u) { *timeout = 1; break; }
            }
        }
        __syncthreads();
        const int base = ((it & 1) * gridDim.x + c * W) * SLAB_FLOATS * 4;
        for (int i = tid * 4; i < W * SLAB_FLOATS; i += THREADS * 4) {
            auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, base + i * 4, 0, 16);
            acc += __builtin_bit_cast(f32x4_t, v) * 1e-9f;
        }
    }
    sink[wg * THREADS + tid] = acc.x + acc.y + acc.z + acc.w;
}

template <int SLAB, int THREADS>
void run(int G, int W, int iters, bool loaded, hipStream_t s_hop, hipStream_t s_bg, const f2* in, const float* mag, f2* out, size_t n) {
    float *slabs, *sink;
    unsigned *counters, *timeout;
    (void)hipMalloc(&slabs, (size_t)2 * G * SLAB * sizeof(float));
    (void)hipMalloc(&sink, (size_t)G * THREADS * sizeof(float));
    (void)hipMalloc(&counters, 64 * 64 * sizeof(unsigned));
    (void)hipMalloc(&timeout, sizeof(unsigned));
    hipEvent_t a, b, c, d;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventCreate(&c); (void)hipEventCreate(&d);
    float best = 1e30f, bg_ms = 0;
    unsigned tmo = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(counters, 0, 64 * 64 * sizeof(unsigned));
        (void)hipMemset(timeout, 0, sizeof(unsigned));
        (void)hipDeviceSynchronize();
        if (loaded) {   // hop workgroups first (they need their compute units), then the stream load fills the rest
            (void)hipEventRecord(c, s_bg);
        }
        (void)hipEventRecord(a, s_hop);
        hipLaunchKernelGGL((hop_kernel_wide<SLAB, THREADS>), dim3(G), dim3(THREADS), 0, s_hop, slabs, counters, timeout, W, iters, sink);
        (void)hipEventRecord(b, s_hop);
        if (loaded) {
            hipLaunchKernelGGL(stream_load_kernel, dim3(224), dim3(512), 0, s_bg, in, mag, out, n, 12);
            (void)hipEventRecord(d, s_bg);
        }
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
        if (loaded) (void)hipEventElapsedTime(&bg_ms, c, d);
        (void)hipMemcpy(&tmo, timeout, sizeof(unsigned), hipMemcpyDeviceToHost);
    }
    printf("%s G=%3d x %d threads, clusters of %2d, slab %5d B: %.2f us per hop (%s%s)", loaded ? "LOADED" : "alone ", G, THREADS, W, SLAB * 4,
           best * 1e3 / iters, hipGetErrorString(hipGetLastError()), tmo ? ", TIMEOUT" : "");
    if (loaded) printf("   [stream load: %.2f ms for %.1f GB = %.2f TB/s]", bg_ms, 12 * 20.0 * n / 1e9, 12 * 20.0 * n / bg_ms / 1e9);
    printf("\n");
    (void)hipFree(slabs); (void)hipFree(sink); (void)hipFree(counters); (void)hipFree(timeout);
}

int main() {
    const size_t n = (size_t)64 * 1025 * 1000;   // one Griffin-Lim iteration's worth of bins
    f2 *in, *out;
    float* mag;
    (void)hipMalloc(&in, n * sizeof(f2));
    (void)hipMalloc(&out, n * sizeof(f2));
    (void)hipMalloc(&mag, n * sizeof(float));
    (void)hipMemset(in, 0x11, n * sizeof(f2));
    (void)hipMemset(mag, 0x22, n * sizeof(float));
    hipStream_t s_hop, s_bg;
    int lo, hi;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    (void)hipStreamCreateWithPriority(&s_hop, hipStreamNonBlocking, hi);
    (void)hipStreamCreateWithPriority(&s_bg, hipStreamNonBlocking, lo);
    const int iters = 2000;
    for (int loaded = 0; loaded < 2; ++loaded) {
        run<256, 256>(64, 16, iters, loaded, s_hop, s_bg, in, mag, out, n);
        run<256, 256>(128, 32, iters, loaded, s_hop, s_bg, in, mag, out, n);
        run<256, 512>(128, 32, iters, loaded, s_hop, s_bg, in, mag, out, n);
        run<256, 256>(32, 32, iters, loaded, s_hop, s_bg, in, mag, out, n);
        run<256, 256>(32, 8, iters, loaded, s_hop, s_bg, in, mag, out, n);
        run<1024, 256>(32, 8, iters, loaded, s_hop, s_bg, in, mag, out, n);
        run<64, 256>(128, 32, iters, loaded, s_hop, s_bg, in, mag, out, n);
    }
    return 0;
}
