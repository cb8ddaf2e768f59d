// CU reservation for stream-level pipelining (gfx950).
//
// tts_synthesize can overlap the latency-bound decoder loop of call k+1 (a chain of ~2200 small
// dependent launches that use a few dozen CUs for a few microseconds each) with the Griffin-Lim
// iterations of call k (throughput bound: every workgroup needs a whole CU's LDS and registers).
// Left alone, the Griffin-Lim workgroups refill every CU the moment one frees up and the decoder's
// launches starve; stream priorities and CU masks did not change that on MI355X / ROCm 7.2.
//
// What does work is occupancy arithmetic: a `cu_hold_kernel` workgroup allocates 64 KB of LDS and
// then sleeps.  A CU that hosts one can no longer admit a Griffin-Lim workgroup (147 KB LDS) but
// still has room for the decoder's and encoder's workgroups (4-37 KB LDS, moderate registers).
// `reserve_cus` such workgroups therefore reserve that many CUs for the front stream without any
// driver support.  The dispatcher deals workgroups round-robin over the 32 shader engines (8 XCDs x 4 SEs of 8
// CUs) and a launch stalls while ANY engine has no CU able to take its next workgroup, so the useful
// reservations are multiples of 32 (one CU per engine: tools/cu_census.hip shows 32 sleepers landing exactly so;
// with fewer free engines the decoder's launches wait for Griffin-Lim workgroups to exit).  They poll one flag word with s_sleep between polls and ALWAYS terminate: either
// the flag is set (the decoder finished) or the wall-clock bound expires.
#include "tts_common.h"

namespace tts {

__global__ __launch_bounds__(64) void cu_hold_kernel(const int* flag, unsigned long long timeout_ticks) {
    extern __shared__ char hold[];
    if (threadIdx.x == 0) hold[0] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(127);
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) break;
    }
}

// Function attributes are per device: every handle calls this once on its own device.
hipError_t cu_hold_configure() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&cu_hold_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_cu_hold(hipStream_t s, int n_cus, const int* flag, double timeout_ms, int lds_kb) {
    const unsigned long long ticks = (unsigned long long)(timeout_ms * 1e5);
    hipLaunchKernelGGL(cu_hold_kernel, dim3(n_cus), dim3(64), (size_t)lds_kb * 1024, s, flag, ticks);
    return hipGetLastError();
}

}  // namespace tts
