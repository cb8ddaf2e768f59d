// Hand-off latency between two workgroups on the SAME XCD and on DIFFERENT XCDs of an MI355X, for the scopes a hand-off can use
// (tools only; round 6, decoder_ws.hip's cluster placement):
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/xcd_pingpong tools/xcd_pingpong.hip && tools/bin/xcd_pingpong
// 64 workgroups are launched; each reads HW_REG_XCC_ID; the host then picks pairs.  A round trip = A stores a 16-byte payload
// (aux = SC), adds to a counter (scope S); B polls the counter (scope S), loads the payload (aux = SC), stores its own, adds to
// the second counter; A polls that.  MODE 0: sc1 stores / loads + agent-scope atomics (what dec_ws_kernel does).  MODE 1 (sc0 +
// workgroup scope) is kept to show that it does NOT work: a workgroup-scope load may hit in the compute unit's L1, the poll never
// sees the other workgroup's arrival (reported as TIMED OUT) -- there is no scope between "one compute unit" and "the device".
// Measured (round 6): 1.70 us per round trip inside an XCD, 1.90-2.11 us across XCDs: 0.85 against 0.95-1.05 us per hand-off --
// a decoder cluster placed on one XCD would save a tenth of a microsecond of the 3.07 us its phase takes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)0xFFFFFFF0u, 0x00020000);
}
__global__ void xcc_kernel(unsigned* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x & 0xf;
}
template <int MODE>
__global__ void pingpong(int wg_a, int wg_b, unsigned* cnt, float* buf, int rounds, unsigned long long* ticks, unsigned* xcc_seen, float* sink) {
    if ((int)blockIdx.x != wg_a && (int)blockIdx.x != wg_b) return;
    const bool is_a = (int)blockIdx.x == wg_a;
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) xcc_seen[is_a ? 0 : 1] = x & 0xf;
    unsigned* mine = cnt + (is_a ? 0 : 32);
    unsigned* theirs = cnt + (is_a ? 32 : 0);
    const __amdgpu_buffer_rsrc_t rs = rsrc(buf);
    constexpr int AUX = MODE == 0 ? 16 : 1;   // sc1 : sc0
    float acc = 0.f;
    __shared__ int dead[1];
    if (threadIdx.x == 0) dead[0] = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= rounds; ++r) {
        if (is_a) {
            u32x4 v = {(unsigned)r, (unsigned)threadIdx.x, 0u, 0u};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(threadIdx.x * 16), 0, AUX);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) {
                if (MODE == 0) __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        // wait for the other side's r-th arrival (A waits after publishing, B before)
        if (threadIdx.x == 0 && !dead[0]) {
            unsigned spins = 0;
            const __amdgpu_buffer_rsrc_t rc = rsrc(theirs);
            // (MODE 1: an L1-bypassing load -- a workgroup-scope atomic load is a plain, cacheable one)
            while ((MODE == 0 ? __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                              : (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rc, 0, 0, 1)) < (unsigned)r) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 200000u) { dead[0] = 1; break; }
            }
        }
        __syncthreads();
        u32x4 got = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((is_a ? 4096 : 0) + threadIdx.x * 16), 0, AUX);
        acc += (float)got[0];
        if (!is_a) {
            u32x4 v = {(unsigned)r, (unsigned)threadIdx.x, 1u, 0u};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(4096 + threadIdx.x * 16), 0, AUX);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) {
                if (MODE == 0) __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    if (threadIdx.x == 0 && is_a) ticks[0] = dead[0] ? 0ull : __builtin_amdgcn_s_memrealtime() - t0;
    // payload check: the last value read must be the last round (stale reads show as a smaller sum)
    if (threadIdx.x == 0) sink[is_a ? 0 : 1] = acc;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int N = 64, rounds = 2000;
    unsigned *xcc, *cnt, *seen; float *buf, *sink; unsigned long long* ticks;
    CHECK(hipMalloc(&xcc, N * 4)); CHECK(hipMalloc(&cnt, 64 * 4)); CHECK(hipMalloc(&seen, 8)); CHECK(hipMalloc(&buf, 8192)); CHECK(hipMalloc(&sink, 8)); CHECK(hipMalloc(&ticks, 8));
    std::vector<unsigned> hx(N);
    hipLaunchKernelGGL(xcc_kernel, dim3(N), dim3(64), 0, 0, xcc);
    CHECK(hipMemcpy(hx.data(), xcc, N * 4, hipMemcpyDeviceToHost));
    printf("XCC of workgroups 0..15:"); for (int i = 0; i < 16; ++i) printf(" %u", hx[i]); printf("\n");
    const double want = 0.5 * rounds * (rounds + 1.0);
    for (int mode = 0; mode < 2; ++mode)
        for (int pair = 0; pair < 3; ++pair) {
            const int a = 0, b = pair == 0 ? 8 : (pair == 1 ? 1 : 4);
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipMemset(cnt, 0, 64 * 4)); CHECK(hipMemset(buf, 0, 8192));
                if (mode == 0) hipLaunchKernelGGL(pingpong<0>, dim3(N), dim3(256), 0, 0, a, b, cnt, buf, rounds, ticks, seen, sink);
                else hipLaunchKernelGGL(pingpong<1>, dim3(N), dim3(256), 0, 0, a, b, cnt, buf, rounds, ticks, seen, sink);
                CHECK(hipDeviceSynchronize());
                unsigned long long t; unsigned s[2]; float k[2];
                CHECK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(s, seen, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(k, sink, 8, hipMemcpyDeviceToHost));
                if (rep) printf(t == 0 ? "TIMED OUT  %s  workgroups %d (XCC %u) <-> %d (XCC %u)\n" : "%s  workgroups %d (XCC %u) <-> %d (XCC %u): %.2f us per round trip (two hand-offs); payload sums %.0f %.0f of %.0f\n",
                                mode == 0 ? "sc1 + agent scope    " : "sc0 + workgroup scope", a, s[0], b, s[1], t * 0.01 / rounds, k[0], k[1], want);
            }
        }
    return 0;
}
