// Ceiling probe, second form: post-net projection 1 as it is in the network -- a k = 3 'SAME' convolution over (B, T, C)
// activations, C[m][n] = sum_tap sum_c X[b][t + tap - 1][c] W[n][tap * C + c] -- with the ACTIVATIONS as row-major bf16 planes
// (hi / mid / lo, [plane][B (T + 2) rows, one zero row before and after every utterance][C]: what a producing layer's
// epilogue could write) and the weights as pre-split tile images; both staged by LDS-DMA, no vector instructions in the loader.
//   tile 256 (M) x 128 (N), 512 threads = 8 waves (4 x 2), wave tile 64 x 64; k step 32 = one LDS stage of
//   3 planes x (256 + 128) rows x 64 B = 72 KB, two stages; a row's four 16-byte slots swizzled by (row >> 2) & 3 on the
//   SOURCE side (the LDS image is lane-linear); LDS-DMA issued by inline asm, counted by hand, one raw barrier per step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/gemm_conv_split_mb.bin tools/gemm_conv_split_microbench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
#define TMR 256
#define TNR 128
#define BK 32
#define A_PLANE (TMR * BK * 2)             // 16 KB
#define B_PLANE (TNR * BK * 2)             //  8 KB
#define A_BYTES (3 * A_PLANE)
#define B_BYTES (3 * B_PLANE)
#define STAGE_BYTES (A_BYTES + B_BYTES)    // 72 KB

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x);
    const float r1 = x - __uint_as_float(h & 0xFFFF0000u);
    m = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = __float_as_uint(r2);
}

// X [B T][C] f32 -> planes [3][B (T + 2)][C] bf16, zero rows around every utterance; one thread per (padded row, 8 channels)
__global__ void split_rows_kernel(const float* __restrict__ X, int B, int T, int C, unsigned short* __restrict__ planes) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cch = C / 8;
    const size_t rows = (size_t)B * (T + 2);
    if (idx >= rows * cch) return;
    const int kc = (int)(idx % cch);
    const size_t pr = idx / cch;
    const int b = (int)(pr / (T + 2)), tp = (int)(pr % (T + 2));
    const bool live = tp >= 1 && tp <= T;
    unsigned short out[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = live ? X[((size_t)b * T + tp - 1) * C + kc * 8 + i] : 0.f;
        unsigned h, m, l;
        split3(v, h, m, l);
        out[0][i] = (unsigned short)(h >> 16); out[1][i] = (unsigned short)(m >> 16); out[2][i] = (unsigned short)(l >> 16);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint4*>(planes + ((size_t)p * rows + pr) * C + kc * 8) = *reinterpret_cast<const uint4*>(out[p]);
}

// W [N][K] f32 -> image [n tile of 128][k step of 32][plane][128 rows][4 slots of 8 bf16], slot = chunk ^ ((row >> 2) & 3)
__global__ void pack_w_kernel(const float* __restrict__ W, int N, int K, unsigned short* __restrict__ img, int Npad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kch = K / 8;
    if (idx >= (size_t)Npad * kch) return;
    const int kc = (int)(idx % kch), n = (int)(idx / kch);
    const int tile = n / TNR, rr = n % TNR, ks = kc >> 2, chunk = kc & 3;
    const int slot = chunk ^ ((rr >> 2) & 3);
    unsigned short out[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = n < N ? W[(size_t)n * K + kc * 8 + i] : 0.f;
        unsigned h, m, l;
        split3(v, h, m, l);
        out[0][i] = (unsigned short)(h >> 16); out[1][i] = (unsigned short)(m >> 16); out[2][i] = (unsigned short)(l >> 16);
    }
    const size_t base = ((size_t)tile * (K / BK) + ks) * 3 * (TNR * BK);
#pragma unroll
    for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint4*>(img + base + (size_t)p * TNR * BK + rr * BK + slot * 8) = *reinterpret_cast<const uint4*>(out[p]);
}

// MODE 0: the product; 1: no DMA in the loop; 2: DMA only
template <int MODE>
__global__ __launch_bounds__(512) void conv_kernel(const unsigned char* __restrict__ Apl, const unsigned char* __restrict__ Bimg,
                                                   float* __restrict__ Cout, int Bn, int T, int C, int N, int taps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int M = Bn * T;
    const int csteps = C / BK, ksteps = taps * csteps;
    const size_t plane_stride = (size_t)Bn * (T + 2) * C * 2;   // bytes
    // workgroup -> tile: the two N tiles of an M tile next to each other in launch order (the second finds the rows in L2)
    const int m_tile = blockIdx.x >> 1, n_tile = blockIdx.x & 1;
    const int m0 = m_tile * TMR;

    // this thread's two A rows (tile rows tid >> 2 and + 128) and its slot; padded row of tap 0 = b (T + 2) + t (tap - 1 + 1)
    const unsigned char* a_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 2) + 128 * i;
        int m = m0 + row;
        if (m >= M) m = M - 1;   // (rows past M: any valid address, never stored)
        const int b = m / T, t = m - b * T;
        const int lslot = (tid & 3) ^ ((row >> 2) & 3);
        a_src[i] = Apl + ((size_t)b * (T + 2) + t) * C * 2 + lslot * 16;
    }
    const unsigned char* b_src = Bimg + (size_t)n_tile * ksteps * B_BYTES + tid * 16;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto glds16 = [&](const unsigned char* src, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    auto issue = [&](int s, int buf) {
        const int tap = s / csteps, cs = s - tap * csteps;
        const size_t aoff = (size_t)tap * C * 2 + (size_t)cs * (BK * 2);
        const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + wave * 1024);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16(a_src[i] + p * plane_stride + aoff, d + p * A_PLANE + i * 8192);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) glds16(b_src + (size_t)s * B_BYTES + p * B_PLANE, d + A_BYTES + p * B_PLANE);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int a_off[2][2], b_off[2][2];   // [block][k16 half]
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ra = wm * 64 + blk * 32 + li, rb = wn * 64 + blk * 32 + li;
            a_off[blk][q] = ra * 64 + (((2 * q + lh) ^ ((ra >> 2) & 3)) << 4);
            b_off[blk][q] = A_BYTES + rb * 64 + (((2 * q + lh) ^ ((rb >> 2) & 3)) << 4);
        }

    issue(0, 0);
    int buf = 0;
    for (int s = 0; s < ksteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (MODE != 1 && s + 1 < ksteps) issue(s + 1, buf ^ 1);
        if (MODE != 2) {
            const unsigned char* st = smem + buf * STAGE_BYTES;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint4 fa[2][3], fb[2][3];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        fa[blk][p] = *reinterpret_cast<const uint4*>(st + p * A_PLANE + a_off[blk][q]);
                        fb[blk][p] = *reinterpret_cast<const uint4*>(st + p * B_PLANE + b_off[blk][q]);
                    }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
#define MMA(SA, SB)                                                                                              \
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[mb][SA]), \
                                                                              __builtin_bit_cast(bf16x8_t, fb[nb][SB]), acc[mb][nb], 0, 0, 0);
                        MMA(0, 2) MMA(2, 0) MMA(1, 1) MMA(0, 1) MMA(1, 0) MMA(0, 0)
#undef MMA
                    }
            }
        }
        buf ^= 1;
    }
    const int mw = m0 + wm * 64, nw = n_tile * TNR + wn * 64;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mw + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = nw + nb * 32 + li;
                if (m < M && n < N) Cout[(size_t)m * N + n] = acc[mb][nb][r];
            }
}


// the same for k steps of 16: [n tile][k step][plane][128 rows][2 slots of 8 bf16], slot = k half ^ ((row >> 3) & 1)
__global__ void pack_w16_kernel(const float* __restrict__ W, int N, int K, unsigned short* __restrict__ img, int Npad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kch = K / 8;
    if (idx >= (size_t)Npad * kch) return;
    const int kc = (int)(idx % kch), n = (int)(idx / kch);
    const int tile = n / TNR, rr = n % TNR, ks = kc >> 1, half = kc & 1;
    const int slot = half ^ ((rr >> 3) & 1);
    unsigned short out[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = n < N ? W[(size_t)n * K + kc * 8 + i] : 0.f;
        unsigned h, m, l;
        split3(v, h, m, l);
        out[0][i] = (unsigned short)(h >> 16); out[1][i] = (unsigned short)(m >> 16); out[2][i] = (unsigned short)(l >> 16);
    }
    const size_t base = ((size_t)tile * (K / 16) + ks) * 3 * (TNR * 16);
#pragma unroll
    for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint4*>(img + base + (size_t)p * TNR * 16 + rr * 16 + slot * 8) = *reinterpret_cast<const uint4*>(out[p]);
}

// k steps of 16: a stage is 3 planes x (256 + 128) rows x 32 B = 36 KB, FOUR stages in the ring, three in flight; the A planes by
// all 512 threads (one 16-byte piece each per plane), the B planes by waves 0..3; counted vmcnt per wave, one raw barrier per step
template <int MODE>
__global__ __launch_bounds__(512) void conv16_kernel(const unsigned char* __restrict__ Apl, const unsigned char* __restrict__ Bimg,
                                                     float* __restrict__ Cout, int Bn, int T, int C, int N, int taps) {
    constexpr int AP = TMR * 32, BP = TNR * 32, AB = 3 * AP, ST = AB + 3 * BP, NB = 4;   // bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int M = Bn * T;
    const int csteps = C / 16, ksteps = taps * csteps;
    const size_t plane_stride = (size_t)Bn * (T + 2) * C * 2;
    const int m_tile = blockIdx.x >> 1, n_tile = blockIdx.x & 1;
    const int m0 = m_tile * TMR;
    const unsigned char* a_src;
    {
        const int row = tid >> 1;
        int m = m0 + row;
        if (m >= M) m = M - 1;
        const int b = m / T, t = m - b * T;
        a_src = Apl + ((size_t)b * (T + 2) + t) * C * 2 + (((tid & 1) ^ ((row >> 3) & 1)) << 4);
    }
    const unsigned char* b_src = Bimg + (size_t)n_tile * ksteps * (3 * BP) + (tid & 255) * 16;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto glds16 = [&](const unsigned char* src, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    auto issue = [&](int s, int buf) {
        const int tap = s / csteps, cs = s - tap * csteps;
        const size_t aoff = (size_t)tap * C * 2 + (size_t)cs * 32;
        const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + buf * ST + wave * 1024);
#pragma unroll
        for (int p = 0; p < 3; ++p) glds16(a_src + p * plane_stride + aoff, d + p * AP);
        if (wave < 4) {
#pragma unroll
            for (int p = 0; p < 3; ++p) glds16(b_src + (size_t)s * (3 * BP) + p * BP, d + AB + p * BP);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int a_off[2], b_off[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int ra = wm * 64 + blk * 32 + li, rb = wn * 64 + blk * 32 + li;
        a_off[blk] = ra * 32 + ((lh ^ ((ra >> 3) & 1)) << 4);
        b_off[blk] = AB + rb * 32 + ((lh ^ ((rb >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int s = 0; s < NB - 1; ++s)
        if (s < ksteps) issue(s, s);
    int buf = 0;
    for (int s = 0; s < ksteps; ++s) {
        const int later = (ksteps - 1 - s) < (NB - 2) ? (ksteps - 1 - s) : (NB - 2);
        if (wave < 4) {
            if (later >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (later >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (MODE != 1 && s + NB - 1 < ksteps) issue(s + NB - 1, (buf + NB - 1) & (NB - 1));
        if (MODE != 2) {
            const unsigned char* st = smem + buf * ST;
            uint4 fa[2][3], fb[2][3];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    fa[blk][p] = *reinterpret_cast<const uint4*>(st + p * AP + a_off[blk]);
                    fb[blk][p] = *reinterpret_cast<const uint4*>(st + p * BP + b_off[blk]);
                }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
#define MMA(SA, SB)                                                                                              \
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[mb][SA]), \
                                                                          __builtin_bit_cast(bf16x8_t, fb[nb][SB]), acc[mb][nb], 0, 0, 0);
                    MMA(0, 2) MMA(2, 0) MMA(1, 1) MMA(0, 1) MMA(1, 0) MMA(0, 0)
#undef MMA
                }
        }
        buf = (buf + 1) & (NB - 1);
    }
    const int mw = m0 + wm * 64, nw = n_tile * TNR + wn * 64;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mw + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = nw + nb * 32 + li;
                if (m < M && n < N) Cout[(size_t)m * N + n] = acc[mb][nb][r];
            }
}

template <int MODE, bool K16 = false>
static void run(const unsigned char* Apl, const unsigned char* Bimg, float* dC, int B, int T, int C, int N, int taps,
                const std::vector<float>& hX, const std::vector<float>& hW) {
    const int M = B * T, K = taps * C;
    const size_t lds = K16 ? 4 * 36864 : 2 * STAGE_BYTES;
    auto kern = K16 ? &conv16_kernel<MODE> : &conv_kernel<MODE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = ((M + TMR - 1) / TMR) * 2;
    CK(hipMemset(dC, 0, (size_t)M * N * 4));
    kern<<<grid, 512, lds>>>(Apl, Bimg, dC, B, T, C, N, taps);
    CK(hipDeviceSynchronize());
    double num = 0, den = 0;
    if (MODE == 0) {
        std::vector<float> hC((size_t)M * N);
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        for (int s = 0; s < 3000; ++s) {
            // (rows at utterance edges among the samples: every 50th)
            const int m = s % 50 == 0 ? (int)(((long long)s * 37) % B) * T + (s % 100 == 0 ? 0 : T - 1) : (int)(((long long)s * 7919 + 13) % M);
            const int n = (s * 131 + 7) % N;
            const int b = m / T, t = m % T;
            double ref = 0;
            for (int tap = 0; tap < taps; ++tap) {
                const int ts = t + tap - (taps - 1) / 2;
                if (ts < 0 || ts >= T) continue;
                for (int c = 0; c < C; ++c) ref += (double)hX[((size_t)b * T + ts) * C + c] * (double)hW[(size_t)n * K + tap * C + c];
            }
            const double d = hC[(size_t)m * N + n] - ref;
            num += d * d; den += ref * ref;
        }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) kern<<<grid, 512, lds>>>(Apl, Bimg, dC, B, T, C, N, taps);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("B %d T %d C %d N %d taps %d  %s MODE %d: %8.1f us  %6.1f TFLOP/s f32-equivalent (%6.1f bf16 MFMA)  rel-L2 %.2e\n", B, T, C, N, taps, K16 ? "k16 x 4 stages" : "k32 x 2 stages", MODE,
           ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, 12.0 * M * N * K / (ms * 1e-3) / 1e12, MODE == 0 ? std::sqrt(num / den) : 0.0);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 1000, C = argc > 3 ? atoi(argv[3]) : 1024;
    const int N = argc > 4 ? atoi(argv[4]) : 256, taps = argc > 5 ? atoi(argv[5]) : 3;
    const int M = B * T, K = taps * C, Np = (N + TNR - 1) / TNR * TNR;
    if (Np != 2 * TNR || C % BK || (taps != 3 && taps != 1)) { printf("N must fill two 128-column tiles, C a multiple of 32, taps 1 or 3\n"); return 1; }
    std::vector<float> hX((size_t)M * C), hW((size_t)N * K);
    unsigned s = 4321u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& x : hX) x = rnd();
    for (auto& x : hW) x = rnd() * 0.05f;
    float *dX, *dW, *dC;
    unsigned short *pl, *iw;
    const size_t rows = (size_t)B * (T + 2);
    CK(hipMalloc(&dX, hX.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&pl, 3 * rows * C * 2 + 4096)); CK(hipMalloc(&iw, (size_t)Np * K * 6));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    {
        const size_t na = rows * (C / 8), nw = (size_t)Np * (K / 8);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        split_rows_kernel<<<(unsigned)((na + 255) / 256), 256>>>(dX, B, T, C, pl);   // (first touch)
        CK(hipEventRecord(e0));
        split_rows_kernel<<<(unsigned)((na + 255) / 256), 256>>>(dX, B, T, C, pl);
        CK(hipEventRecord(e1));
        pack_w_kernel<<<(unsigned)((nw + 255) / 256), 256>>>(dW, N, K, iw, Np);
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("split pass over the activations (%d x %d f32 -> three bf16 planes): %.1f us\n", M, C, ms * 1e3);
    }
    // tap 0 of row (b, t) is padded row b (T + 2) + t for three taps ('SAME': t - 1), b (T + 2) + t + 1 for one
    const unsigned char* Apl = (const unsigned char*)pl + (taps == 1 ? (size_t)C * 2 : 0);
    run<0>(Apl, (const unsigned char*)iw, dC, B, T, C, N, taps, hX, hW);
    run<1>(Apl, (const unsigned char*)iw, dC, B, T, C, N, taps, hX, hW);
    run<2>(Apl, (const unsigned char*)iw, dC, B, T, C, N, taps, hX, hW);
    {
        const size_t nw = (size_t)Np * (K / 8);
        pack_w16_kernel<<<(unsigned)((nw + 255) / 256), 256>>>(dW, N, K, iw, Np);
        CK(hipDeviceSynchronize());
    }
    run<0, true>(Apl, (const unsigned char*)iw, dC, B, T, C, N, taps, hX, hW);
    run<1, true>(Apl, (const unsigned char*)iw, dC, B, T, C, N, taps, hX, hW);
    run<2, true>(Apl, (const unsigned char*)iw, dC, B, T, C, N, taps, hX, hW);
    return 0;
}
