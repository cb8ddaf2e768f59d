// Ceiling probe for the three-way-split GEMM of gemm_f32.hip with BOTH operands pre-split into bf16 planes and staged by
// LDS-DMA (no vector instructions in the loader):  C[M][N] = A[M][K] . W[N][K]^T at f32 accuracy, six
// v_mfma_f32_32x32x16_bf16 per 16-deep step and 32 x 32 block.
//   tile 256 x 256, 512 threads = 8 waves (2 x 4), wave tile 128 x 64 = 4 x 2 blocks; k step 16 = one LDS stage of
//   3 planes x (256 + 256) rows x 32 B = 48 KB; NBUF stages in a ring, the LDS-DMA of stage s + NBUF - 1 issued at step s,
//   counted vmcnt, one raw barrier per step.
//   images: [tile][k step][plane][256 rows][2 slots of 8 bf16], slot = k half ^ ((row >> 3) & 1)  (made by pack_kernel)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/gemm_split_mb.bin tools/gemm_split_microbench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
#define TM 256
#define KS 16
#define PLANE_BYTES (TM * KS * 2)          // 8 KB
#define OP_BYTES (3 * PLANE_BYTES)         // 24 KB: one operand's stage
#define STAGE_BYTES (2 * OP_BYTES)         // 48 KB

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x);
    const float r1 = x - __uint_as_float(h & 0xFFFF0000u);
    m = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = __float_as_uint(r2);
}

// src [R][K] f32 row-major -> image; one thread per (row, 8-k chunk); rows past R are zero
__global__ void pack_kernel(const float* __restrict__ src, int R, int K, unsigned short* __restrict__ img, int Rpad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kchunks = K / 8;
    if (idx >= (size_t)Rpad * kchunks) return;
    const int kc = (int)(idx % kchunks);
    const int r = (int)(idx / kchunks);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = r < R ? src[(size_t)r * K + kc * 8 + i] : 0.f;
    const int tile = r / TM, rr = r % TM, ks = kc >> 1, half = kc & 1;
    const int slot = half ^ ((rr >> 3) & 1);
    const size_t base = ((size_t)tile * (K / KS) + ks) * 3 * (TM * KS);
    unsigned short out[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        unsigned h, m, l;
        split3(v[i], h, m, l);
        out[0][i] = (unsigned short)(h >> 16); out[1][i] = (unsigned short)(m >> 16); out[2][i] = (unsigned short)(l >> 16);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 8; ++i) img[base + (size_t)p * TM * KS + rr * KS + slot * 8 + i] = out[p][i];
}

template <int NBUF, int MODE = 0>
__global__ __launch_bounds__(512) void gemm_kernel(const unsigned char* __restrict__ Aimg, const unsigned char* __restrict__ Bimg,
                                                   float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    const int ksteps = K / KS;
    const unsigned char* Ab = Aimg + (size_t)blockIdx.x * ksteps * OP_BYTES;
    const unsigned char* Bb = Bimg + (size_t)blockIdx.y * ksteps * OP_BYTES;

    // LDS-DMA by inline asm: hipcc does not count an asm load, so the only vmcnt waits in the loop are the counted ones below
    // (with the builtin it waits vmcnt(0) in front of the first ds_read of every step: the DMA of stage s + 2 drained at step s)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto glds16 = [&](const unsigned char* src, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    auto issue = [&](int s, int buf) {
        const unsigned char* a = Ab + (size_t)s * OP_BYTES + tid * 16;
        const unsigned char* b = Bb + (size_t)s * OP_BYTES + tid * 16;
        const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + wave * 1024);
#pragma unroll
        for (int i = 0; i < 3; ++i) glds16(a + i * 8192, d + i * 8192);
#pragma unroll
        for (int i = 0; i < 3; ++i) glds16(b + i * 8192, d + OP_BYTES + i * 8192);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment byte offsets inside a plane (the same for every stage)
    int a_off[4], b_off[2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int row = wm * 128 + mb * 32 + li;
        a_off[mb] = row * 32 + ((lh ^ ((row >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int row = wn * 64 + nb * 32 + li;
        b_off[nb] = OP_BYTES + row * 32 + ((lh ^ ((row >> 3) & 1)) << 4);
    }

#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        if (s < ksteps) issue(s, s);
    int buf = 0;
    for (int s = 0; s < ksteps; ++s) {
        // stage s has landed when at most the later stages' DMAs of this wave are outstanding
        const int later = (ksteps - 1 - s) < (NBUF - 2) ? (ksteps - 1 - s) : (NBUF - 2);
        if (later >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (MODE != 1 && s + NBUF - 1 < ksteps) {   // MODE 1: no DMA in the loop (what ds_read + MFMA + barrier cost alone)
            int nb_ = buf + NBUF - 1;
            if (nb_ >= NBUF) nb_ -= NBUF;
            issue(s + NBUF - 1, nb_);
        }
        if (MODE == 2) { if (++buf == NBUF) buf = 0; continue; }   // MODE 2: no fragment reads, no MFMAs (the DMA stream alone)
        const unsigned char* st = smem + buf * STAGE_BYTES;
        uint4 fa[4][3], fb[2][3];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[mb][p] = *reinterpret_cast<const uint4*>(st + p * PLANE_BYTES + a_off[mb]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[nb][p] = *reinterpret_cast<const uint4*>(st + p * PLANE_BYTES + b_off[nb]);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
#define MMA(SA, SB)                                                                                              \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[mb][SA]), \
                                                                      __builtin_bit_cast(bf16x8_t, fb[nb][SB]), acc[mb][nb], 0, 0, 0);
                MMA(0, 2) MMA(2, 0) MMA(1, 1) MMA(0, 1) MMA(1, 0) MMA(0, 0)
#undef MMA
            }
        if (++buf == NBUF) buf = 0;
    }
    // C/D map of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int m0 = blockIdx.x * TM + wm * 128, n0 = blockIdx.y * TM + wn * 64;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + nb * 32 + li;
                if (m < M && n < N) C[(size_t)m * N + n] = acc[mb][nb][r];
            }
}

// STAGGERED form: the two waves of every SIMD (wm = 0 / 1) run half a step apart -- one reads its fragments while the other
// multiplies.  Two barriers per step; group 1 is one barrier behind.  A wave waits for its own DMAs of stage s + 1 BEFORE the
// mid-step barrier of step s, so that whoever passes the next barrier finds the whole stage in LDS.
template <int NBUF, int MODE>
__global__ __launch_bounds__(512) void gemm_stag_kernel(const unsigned char* __restrict__ Aimg, const unsigned char* __restrict__ Bimg,
                                                        float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    const int ksteps = K / KS;
    const unsigned char* Ab = Aimg + (size_t)blockIdx.x * ksteps * OP_BYTES;
    const unsigned char* Bb = Bimg + (size_t)blockIdx.y * ksteps * OP_BYTES;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto glds16 = [&](const unsigned char* src, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    auto issue = [&](int s, int buf) {
        const unsigned char* a = Ab + (size_t)s * OP_BYTES + tid * 16;
        const unsigned char* b = Bb + (size_t)s * OP_BYTES + tid * 16;
        const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + wave * 1024);
#pragma unroll
        for (int i = 0; i < 3; ++i) glds16(a + i * 8192, d + i * 8192);
#pragma unroll
        for (int i = 0; i < 3; ++i) glds16(b + i * 8192, d + OP_BYTES + i * 8192);
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int a_off[4], b_off[2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int row = wm * 128 + mb * 32 + li;
        a_off[mb] = row * 32 + ((lh ^ ((row >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int row = wn * 64 + nb * 32 + li;
        b_off[nb] = OP_BYTES + row * 32 + ((lh ^ ((row >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        if (s < ksteps) issue(s, s);
    // stage 0 of this wave has landed: at most the later prologue stages outstanding
    if (NBUF == 3 && ksteps > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wm == 1) __builtin_amdgcn_s_barrier();
    int buf = 0;
    for (int s = 0; s < ksteps; ++s) {
        __builtin_amdgcn_s_barrier();
        if (MODE != 1 && s + NBUF - 1 < ksteps) {
            int nb_ = buf + NBUF - 1;
            if (nb_ >= NBUF) nb_ -= NBUF;
            issue(s + NBUF - 1, nb_);
        }
        const unsigned char* st = smem + buf * STAGE_BYTES;
        uint4 fa[4][3], fb[2][3];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[mb][p] = *reinterpret_cast<const uint4*>(st + p * PLANE_BYTES + a_off[mb]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[nb][p] = *reinterpret_cast<const uint4*>(st + p * PLANE_BYTES + b_off[nb]);
        // own DMAs of stage s + 1 landed (the ones of stage s + 2, just issued, may stay in flight), fragments in registers
        if (NBUF == 3 && s + 2 < ksteps) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
#define MMA(SA, SB)                                                                                              \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[mb][SA]), \
                                                                      __builtin_bit_cast(bf16x8_t, fb[nb][SB]), acc[mb][nb], 0, 0, 0);
                MMA(0, 2) MMA(2, 0) MMA(1, 1) MMA(0, 1) MMA(1, 0) MMA(0, 0)
#undef MMA
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (++buf == NBUF) buf = 0;
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    const int m0 = blockIdx.x * TM + wm * 128, n0 = blockIdx.y * TM + wn * 64;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + nb * 32 + li;
                if (m < M && n < N) C[(size_t)m * N + n] = acc[mb][nb][r];
            }
}

template <int NBUF, int MODE = 0, bool STAG = false>
static void run(const unsigned char* Aimg, const unsigned char* Bimg, float* C, int M, int N, int K, const std::vector<float>& hA,
                const std::vector<float>& hW) {
    const size_t lds = (size_t)NBUF * STAGE_BYTES;
    auto kern = STAG ? &gemm_stag_kernel<NBUF, MODE> : &gemm_kernel<NBUF, MODE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((M + TM - 1) / TM, (N + TM - 1) / TM);
    CK(hipMemset(C, 0, (size_t)M * N * 4));
    kern<<<grid, 512, lds>>>(Aimg, Bimg, C, M, N, K);
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (int t = 0; t < 4000; ++t) {
        const int m = (int)(((long long)t * 7919 + 13) % M), n = (t * 131 + 7) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
        const double d = hC[(size_t)m * N + n] - ref;
        num += d * d; den += ref * ref;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) kern<<<grid, 512, lds>>>(Aimg, Bimg, C, M, N, K);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("M %d N %d K %d  %s NBUF %d MODE %d: %8.1f us  %6.1f TFLOP/s f32-equivalent (%6.1f bf16 MFMA)  rel-L2 on 4000 samples %.2e\n", M, N, K, STAG ? "staggered" : "in step  ", NBUF, MODE,
           ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, 12.0 * M * N * K / (ms * 1e-3) / 1e12, std::sqrt(num / den));
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64000, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 3072;
    const int Mp = (M + TM - 1) / TM * TM, Np = (N + TM - 1) / TM * TM;
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& x : hA) x = rnd();
    for (auto& x : hW) x = rnd() * 0.05f;
    float *dA, *dW, *dC;
    unsigned short *iA, *iW;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&iA, (size_t)Mp * K * 6)); CK(hipMalloc(&iW, (size_t)Np * K * 6));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    {
        const size_t na = (size_t)Mp * (K / 8), nw = (size_t)Np * (K / 8);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        pack_kernel<<<(unsigned)((na + 255) / 256), 256>>>(dA, M, K, iA, Mp);
        CK(hipEventRecord(e1));
        pack_kernel<<<(unsigned)((nw + 255) / 256), 256>>>(dW, N, K, iW, Np);
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("pack A (%d x %d): %.1f us (a naive kernel: 16-byte stores of 2-byte elements)\n", M, K, ms * 1e3);
    }
    run<2>((const unsigned char*)iA, (const unsigned char*)iW, dC, M, N, K, hA, hW);
    run<3>((const unsigned char*)iA, (const unsigned char*)iW, dC, M, N, K, hA, hW);
    run<2, 0, true>((const unsigned char*)iA, (const unsigned char*)iW, dC, M, N, K, hA, hW);
    run<3, 0, true>((const unsigned char*)iA, (const unsigned char*)iW, dC, M, N, K, hA, hW);
    run<2, 1, true>((const unsigned char*)iA, (const unsigned char*)iW, dC, M, N, K, hA, hW);
    run<2, 1>((const unsigned char*)iA, (const unsigned char*)iW, dC, M, N, K, hA, hW);
    run<2, 2>((const unsigned char*)iA, (const unsigned char*)iW, dC, M, N, K, hA, hW);
    run<3, 2>((const unsigned char*)iA, (const unsigned char*)iW, dC, M, N, K, hA, hW);
    return 0;
}
