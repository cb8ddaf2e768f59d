// Where do the workgroups of a small LDS-heavy launch land?  Prints (XCC, SE, SH, CU) of every workgroup of
// launches shaped like the CU-reservation sleepers (reserve.hip), alone and while persistent 152 KB workgroups
// occupy the rest of the chip.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/cu_census.bin tools/cu_census.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void census_kernel(unsigned* out, const int* flag, unsigned long long timeout_ticks) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        lds[0] = 0;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(127);
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) break;
    }
}

static void report(const char* what, const std::vector<unsigned>& v) {
    std::map<unsigned, int> per_se, per_cu;
    for (unsigned x : v) {
        const unsigned xcc = x >> 16, hw = x & 0xffff;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_se[xcc * 8 + se]++;
        per_cu[((xcc * 8 + se) * 2 + sh) * 16 + cu]++;
    }
    int max_se = 0, max_cu = 0;
    for (auto& kv : per_se) max_se = kv.second > max_se ? kv.second : max_se;
    for (auto& kv : per_cu) max_cu = kv.second > max_cu ? kv.second : max_cu;
    printf("%s: %zu workgroups on %zu distinct CUs in %zu (XCC, SE) groups; max per SE %d, max per CU %d\n", what,
           v.size(), per_cu.size(), per_se.size(), max_se, max_cu);
}

int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&census_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    unsigned* out; int* flag;
    hipMalloc(&out, 1024 * sizeof(unsigned));
    hipMalloc(&flag, sizeof(int));
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    for (int n : {32, 24, 16, 64}) {
        for (int lds_kb : {64, 84}) {
            hipMemset(flag, 0, sizeof(int));
            hipLaunchKernelGGL(census_kernel, dim3(n), dim3(64), lds_kb * 1024, a, out, flag, 2000000ull);   // 20 ms bound
            hipStreamSynchronize(a);
            std::vector<unsigned> v(n);
            hipMemcpy(v.data(), out, n * sizeof(unsigned), hipMemcpyDeviceToHost);
            char name[64];
            snprintf(name, sizeof(name), "%d sleepers of %d KB", n, lds_kb);
            report(name, v);
        }
    }
    // sleepers launched while 224 big workgroups are resident, then the other way round
    hipMemset(flag, 0, sizeof(int));
    hipLaunchKernelGGL(census_kernel, dim3(224), dim3(512), 152 * 1024, a, out, flag, 3000000ull);
    hipLaunchKernelGGL(census_kernel, dim3(32), dim3(64), 64 * 1024, b, out + 512, flag, 2000000ull);
    hipDeviceSynchronize();
    std::vector<unsigned> big(224), sl(32);
    hipMemcpy(big.data(), out, 224 * sizeof(unsigned), hipMemcpyDeviceToHost);
    hipMemcpy(sl.data(), out + 512, 32 * sizeof(unsigned), hipMemcpyDeviceToHost);
    report("224 big workgroups (152 KB, 512 threads)", big);
    report("32 sleepers launched right after them", sl);
    printf("done: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
