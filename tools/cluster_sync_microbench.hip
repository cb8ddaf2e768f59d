// Price of one dependent hop inside a cluster of workgroups (the inter-layer hand-off of a persistent decoder):
// every workgroup of a cluster publishes a slab (write-through sc1 stores), arrives on the cluster's counter, waits
// for its peers and reads all slabs back with sc1 loads (MI355X_MICROARCH.md, valid hand-off forms).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/cluster_sync.bin tools/cluster_sync_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) unsigned gu32;

// 16-byte write-through store / L1-bypassing load (sc1) through a buffer resource: aux bit 4 = sc1
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
template <int SLAB_FLOATS>   // wide variant: one 16-byte sc1 store per thread and hop, 16-byte sc1 loads
__global__ __launch_bounds__(256) void hop_kernel_wide(float* slabs, unsigned* counters, unsigned* timeout, int W, int iters, float* sink) {
    const int wg = blockIdx.x, c = wg / W, tid = threadIdx.x;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    unsigned* cnt = counters + 64 * c;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, (int)((size_t)2 * gridDim.x * SLAB_FLOATS * 4), 0x00020000);
    for (int it = 0; it < iters; ++it) {
        const int mine = ((it & 1) * gridDim.x + wg) * SLAB_FLOATS * 4;
        for (int i = tid * 4; i < SLAB_FLOATS; i += 1024)
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
        for (int i = tid * 4; i < W * SLAB_FLOATS; i += 1024) {
            auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, base + i * 4, 0, 16);
            acc += __builtin_bit_cast(f32x4_t, v) * 1e-9f;
        }
    }
    sink[wg * 256 + tid] = acc.x + acc.y + acc.z + acc.w;
}

template <int SLAB_FLOATS>   // floats published per workgroup and hop (256 threads)
__global__ __launch_bounds__(256) void hop_kernel(float* slabs, unsigned* counters, unsigned* timeout, int W, int iters, float* sink) {
    const int wg = blockIdx.x, c = wg / W, tid = threadIdx.x;
    float acc = 0.f;
    unsigned* cnt = counters + 64 * c;   // one counter per cluster, on its own cache lines
    for (int it = 0; it < iters; ++it) {
        float* mine = slabs + ((size_t)(it & 1) * gridDim.x + wg) * SLAB_FLOATS;
        for (int i = tid; i < SLAB_FLOATS; i += 256)
            __hip_atomic_store(mine + i, acc + (float)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 store
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
        const float* base = slabs + ((size_t)(it & 1) * gridDim.x + (size_t)c * W) * SLAB_FLOATS;
        for (int i = tid; i < W * SLAB_FLOATS; i += 256)
            acc += __hip_atomic_load(base + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 1e-9f;   // sc1 load
    }
    sink[wg * 256 + tid] = acc;
}

template <int SLAB, bool WIDE = false>
void run(int G, int W, int iters) {
    float *slabs, *sink;
    unsigned *counters, *timeout;
    (void)hipMalloc(&slabs, (size_t)2 * G * SLAB * sizeof(float));
    (void)hipMalloc(&sink, (size_t)G * 256 * sizeof(float));
    (void)hipMalloc(&counters, 64 * 64 * sizeof(unsigned));
    (void)hipMalloc(&timeout, sizeof(unsigned));
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e30f;
    unsigned tmo = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(counters, 0, 64 * 64 * sizeof(unsigned));
        (void)hipMemset(timeout, 0, sizeof(unsigned));
        (void)hipEventRecord(a);
        if (WIDE) hipLaunchKernelGGL(hop_kernel_wide<SLAB>, dim3(G), dim3(256), 0, 0, slabs, counters, timeout, W, iters, sink);
        else hipLaunchKernelGGL(hop_kernel<SLAB>, dim3(G), dim3(256), 0, 0, slabs, counters, timeout, W, iters, sink);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
        (void)hipMemcpy(&tmo, timeout, sizeof(unsigned), hipMemcpyDeviceToHost);
    }
    printf("G=%3d workgroups, clusters of %2d, slab %5d B, %s accesses: %.2f us per hop (%s%s)\n", G, W, SLAB * 4, WIDE ? "16-byte" : " 4-byte", best * 1e3 / iters,
           hipGetErrorString(hipGetLastError()), tmo ? ", TIMEOUT" : "");
    (void)hipFree(slabs); (void)hipFree(sink); (void)hipFree(counters); (void)hipFree(timeout);
}

int main() {
    const int iters = 2000;
    run<256>(64, 16, iters);
    run<1024>(64, 16, iters);
    run<256>(64, 8, iters);
    run<256>(32, 8, iters);
    run<256>(32, 32, iters);
    run<256>(64, 64, iters);
    run<1024>(64, 64, iters);
    run<256>(128, 32, iters);
    run<64>(64, 16, iters);
    run<256, true>(64, 16, iters);
    run<1024, true>(64, 16, iters);
    run<1024, true>(64, 8, iters);
    run<1024, true>(32, 8, iters);
    run<256, true>(32, 32, iters);
    run<1024, true>(64, 64, iters);
    return 0;
}
