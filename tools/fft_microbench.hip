// Micro benchmark: cost of one wave-level 1024-point complex FFT (fft1024 of griffin_lim.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/fft_mb tools/fft_microbench.hip && /tmp/fft_mb
#include "../single-speaker-tts_amd/csrc/griffin_lim.hip"
#include <cstdio>
#include <vector>
using namespace tts;

template <int NW>
__global__ __launch_bounds__(NW * 64) void fft_loop_kernel(const cf* tw1024, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* ex_all = reinterpret_cast<cf*>(smem_raw);
    cf* twA = ex_all + NW * EX_CPLX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cf* ex = ex_all + wave * EX_CPLX;
    for (int i = tid; i < 15 * 64; i += NW * 64) twA[i] = tw1024[(i & 63) * ((i >> 6) + 1)];
    FftTw tw;
    for (int d = 1; d < 4; ++d) tw.b[d - 1] = tw1024[16 * (lane & 15) * d];
    tw.a = twA + lane;
    __syncthreads();
    cf v[16];
    for (int j = 0; j < 16; ++j) v[j] = cmk(0.001f * (lane + j), 0.002f * (lane - j));
    for (int it = 0; it < iters; ++it) {
        fft1024(v, ex, tw, lane);
        for (int j = 0; j < 16; ++j) v[j] = cscale(v[j], 1.0f / 32.0f);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += v[j].x + v[j].y;
    out[blockIdx.x * NW * 64 + tid] = s;
}

template <int NW>
void run(const cf* tw, float* out, int blocks, int iters, int extra_lds) {
    const size_t lds = (size_t)(NW * EX_CPLX + 15 * 64) * sizeof(cf) + extra_lds;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_loop_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(fft_loop_kernel<NW>, dim3(blocks), dim3(NW * 64), lds, 0, tw, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(fft_loop_kernel<NW>, dim3(blocks), dim3(NW * 64), lds, 0, tw, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("NW=%d blocks=%d lds=%zu: %.1f us total, %.3f us per FFT per wave (%d iters), err=%s\n", NW, blocks, lds,
           ms * 1e3, ms * 1e3 / iters, iters, hipGetErrorString(hipGetLastError()));
}

int main() {
    std::vector<cf> t1(1024);
    for (int k = 0; k < 1024; ++k) {
        const double a = -2.0 * M_PI * k / 1024.0;
        t1[k] = make_float2((float)cos(a), (float)sin(a));
    }
    cf* tw; float* out;
    hipMalloc(&tw, 1024 * sizeof(cf));
    hipMalloc(&out, 4096 * 1024 * sizeof(float));
    hipMemcpy(tw, t1.data(), 1024 * sizeof(cf), hipMemcpyHostToDevice);
    const int iters = 200;
    run<4>(tw, out, 256, iters, 100 * 1024);   // 1 WG/CU, 1 wave per SIMD
    run<8>(tw, out, 256, iters, 60 * 1024);    // 1 WG/CU, 2 waves per SIMD
    run<8>(tw, out, 512, iters, 60 * 1024);    // 2 rounds of WGs
    run<4>(tw, out, 512, iters, 20 * 1024);    // 2 WGs/CU of 4 waves
    run<4>(tw, out, 768, iters, 4 * 1024);     // 3 WGs/CU of 4 waves: 3 waves per SIMD
    run<8>(tw, out, 512, iters, 0);            // 2 WGs/CU of 8 waves = 4 waves per SIMD (fits only with -DE1S=72 -DEX_CPLX=1152)
    return 0;
}
