// Is a CU-masked stream honoured, and do two 512-thread workgroups share a compute unit of the mask while 240
// whole-CU workgroups (152 KB of LDS, the Griffin-Lim shape) hold the rest of the chip?
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/cu_mask_census.bin tools/cu_mask_census.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void census_kernel(unsigned* out, unsigned* arrived, const int* flag, unsigned long long timeout_ticks) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        lds[0] = 0;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
        __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(127);
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) break;
    }
}

static unsigned cu_key(unsigned x) {
    const unsigned xcc = x >> 16, hw = x & 0xffff;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    return ((xcc * 8 + se) * 2 + sh) * 16 + cu;
}
static std::set<unsigned> report(const char* what, const std::vector<unsigned>& v) {
    std::map<unsigned, int> per_cu, per_xcc;
    for (unsigned x : v) { per_cu[cu_key(x)]++; per_xcc[x >> 16]++; }
    int max_cu = 0;
    for (auto& kv : per_cu) max_cu = kv.second > max_cu ? kv.second : max_cu;
    printf("%s: %zu workgroups on %zu distinct CUs, max per CU %d; per XCC:", what, v.size(), per_cu.size(), max_cu);
    for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
    printf("\n");
    std::set<unsigned> s;
    for (auto& kv : per_cu) s.insert(kv.first);
    return s;
}

int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&census_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    unsigned *out, *arrived; int* flag;
    hipMalloc(&out, 2048 * sizeof(unsigned));
    hipMalloc(&arrived, 2 * sizeof(unsigned));
    hipMalloc(&flag, sizeof(int));
    hipStream_t plain, masked, rest;
    hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
    // 2 CUs of every XCD?  The mask is indexed by the runtime's CU numbering: try "every 16th CU and its neighbour"
    for (int variant = 0; variant < 2; ++variant) {
        uint32_t m16[8] = {0}, m240[8];
        for (int i = 0; i < 256; ++i) {
            const bool in16 = variant == 0 ? (i < 16) : (i % 16 < 1);
            if (in16) m16[i / 32] |= 1u << (i % 32);
        }
        for (int w = 0; w < 8; ++w) m240[w] = ~m16[w];
        hipError_t e1 = hipExtStreamCreateWithCUMask(&masked, 8, m16);
        hipError_t e2 = hipExtStreamCreateWithCUMask(&rest, 8, m240);
        printf("variant %d: create masked %s, rest %s\n", variant, hipGetErrorString(e1), hipGetErrorString(e2));
        if (e1 != hipSuccess || e2 != hipSuccess) continue;
        for (int order = 0; order < 2; ++order) {
            hipMemset(flag, 0, sizeof(int));
            hipMemset(arrived, 0, 2 * sizeof(unsigned));
            hipMemset(out, 0, 2048 * sizeof(unsigned));
            hipDeviceSynchronize();
            auto small = [&]() { hipLaunchKernelGGL(census_kernel, dim3(32), dim3(512), 70 * 1024, masked, out + 1024, arrived + 1, flag, 3000000ull); };
            auto big = [&]() { hipLaunchKernelGGL(census_kernel, dim3(240), dim3(512), 152 * 1024, order < 2 ? rest : plain, out, arrived, flag, 3000000ull); };
            if (order == 0) { small(); big(); } else { big(); small(); }
            // give everything 5 ms to become resident, then read the arrival counts while the kernels still spin
            hipStream_t s3;
            hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
            unsigned host_arr[2] = {0, 0};
            for (int k = 0; k < 50; ++k) {
                hipMemcpyAsync(host_arr, arrived, sizeof(host_arr), hipMemcpyDeviceToHost, s3);
                hipStreamSynchronize(s3);
                if (host_arr[0] == 240 && host_arr[1] == 32) break;
                struct timespec ts = {0, 100000};
                nanosleep(&ts, nullptr);
            }
            printf("  order %d: resident together: big %u / 240, small %u / 32\n", order, host_arr[0], host_arr[1]);
            int one = 1;
            hipMemcpyAsync(flag, &one, sizeof(int), hipMemcpyHostToDevice, s3);
            hipDeviceSynchronize();
            std::vector<unsigned> b(240), s(32);
            hipMemcpy(b.data(), out, 240 * sizeof(unsigned), hipMemcpyDeviceToHost);
            hipMemcpy(s.data(), out + 1024, 32 * sizeof(unsigned), hipMemcpyDeviceToHost);
            auto sb = report("  240 whole-CU workgroups on the complement mask", b);
            auto ss = report("  32 half-CU workgroups on the 16-CU mask", s);
            int common = 0;
            for (unsigned c : ss) common += sb.count(c);
            printf("  CUs used by both: %d\n", common);
            hipStreamDestroy(s3);
        }
        hipStreamDestroy(masked);
        hipStreamDestroy(rest);
    }
    printf("done: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
