// Design-space probe for the fp32 MFMA GEMM of gemm_f32.hip on dense operands: C[M][N] = A[M][K] . Wt[N][K]^T.
//   V0: the shipped main loop (single LDS buffer, two barriers per k tile)
//   V1: two LDS buffers, one barrier per k tile, the LDS stores of the next tile between the MFMAs of this one
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/gemm_mb.bin tools/gemm_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define BM 128
#define BN 128


__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int V>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ Wt, float* __restrict__ C, int M, int N,
                                                   int K) {
    const int nyb = gridDim.y;
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, seq = lin >> 3;
    const int m_blk = (seq / nyb) * 8 + xcd;
    const int m0 = m_blk * BM;
    const int n0 = (seq % nyb) * BN;
    if (n0 >= N || m0 >= M) return;
    constexpr int BK = V == 2 ? 64 : 32;
    constexpr int LDS_LD = BK + 4;
    constexpr int NBUF = V == 1 ? 2 : 1;
    constexpr int NLD = BK / 8;      // float4 per thread and operand per tile (4 rows x BK/32 column groups)
    __shared__ __attribute__((aligned(16))) float As[NBUF][BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[NBUF][BN * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int kq = tid & 7;
    const float* a_row[4];
    const float* b_row[4];
    float4 ra[NLD], rb[NLD];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        a_row[i] = A + (size_t)min(m0 + row, M - 1) * K;
        b_row[i] = Wt + (size_t)min(n0 + row, N - 1) * K;
    }
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt), 0, (int)0xFFFFFFF0u, 0x00020000);
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (V == 9) {   // what gemm_f32.hip does: raw buffer loads, 32-bit byte offsets
                const unsigned oa = (unsigned)((a_row[i & 3] - A) + kt + 4 * kq + 32 * (i >> 2)) * 4u;
                const unsigned ob = (unsigned)((b_row[i & 3] - Wt) + kt + 4 * kq + 32 * (i >> 2)) * 4u;
                ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, (int)oa, 0, 0));
                rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, (int)ob, 0, 0));
            } else {
                ra[i] = ld4(a_row[i & 3] + kt + 4 * kq + 32 * (i >> 2));
                rb[i] = ld4(b_row[i & 3] + kt + 4 * kq + 32 * (i >> 2));
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int row = (tid >> 3) + 32 * (i & 3);
            *reinterpret_cast<float4*>(&As[buf][row * LDS_LD + 4 * kq + 32 * (i >> 2)]) = ra[i];
            *reinterpret_cast<float4*>(&Bs[buf][row * LDS_LD + 4 * kq + 32 * (i >> 2)]) = rb[i];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute = [&](int buf, int q) {
        float4 a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            a[t] = *reinterpret_cast<const float4*>(&As[buf][(wm * 64 + t * 32 + li) * LDS_LD + 8 * q + 4 * lh]);
            b[t] = *reinterpret_cast<const float4*>(&Bs[buf][(wn * 64 + t * 32 + li) * LDS_LD + 8 * q + 4 * lh]);
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].x, b[tn].x, acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].y, b[tn].y, acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].z, b[tn].z, acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].w, b[tn].w, acc[tm][tn], 0, 0, 0);
            }
    };
    auto frag = [&](int buf, int q, float4 (&a)[2], float4 (&b)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            a[t] = *reinterpret_cast<const float4*>(&As[buf][(wm * 64 + t * 32 + li) * LDS_LD + 8 * q + 4 * lh]);
            b[t] = *reinterpret_cast<const float4*>(&Bs[buf][(wn * 64 + t * 32 + li) * LDS_LD + 8 * q + 4 * lh]);
        }
    };
    auto mma = [&](const float4 (&a)[2], const float4 (&b)[2]) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].x, b[tn].x, acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].y, b[tn].y, acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].z, b[tn].z, acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].w, b[tn].w, acc[tm][tn], 0, 0, 0);
            }
    };
    if (V == 7 || V == 8 || V == 9) {   // LDS fragments of step q + 1 requested before the MFMAs of step q (two register sets)
        load_tile(0);
        if (V == 8) { store_tile(0); __syncthreads(); }
        for (int kt = 0; kt < K; kt += BK) {
            if (V == 7 || V == 9) {
                store_tile(0);
                __syncthreads();
                if (kt + BK < K) load_tile(kt + BK);
            }
            float4 a0[2], b0[2], a1[2], b1[2];
            frag(0, 0, a0, b0);
            frag(0, 1, a1, b1);
            mma(a0, b0);
            frag(0, 2, a0, b0);
            mma(a1, b1);
            frag(0, 3, a1, b1);
            mma(a0, b0);
            mma(a1, b1);
            if (V == 7 || V == 9) __syncthreads();
        }
    } else if (V >= 4) {   // ablations (wrong results): 4 = no global loads in the loop, 5 = also no LDS stores, 6 = also no barriers
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < K; kt += BK) {
            if (V == 4) store_tile(0);
            if (V <= 5) __syncthreads();
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) compute(0, q);
            if (V <= 5) __syncthreads();
        }
    } else if (V != 1) {
        load_tile(0);
        for (int kt = 0; kt < K; kt += BK) {
            store_tile(0);
            __syncthreads();
            if (kt + BK < K) load_tile(kt + BK);
            if (V == 3) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) compute(0, q);
            if (V == 3) __builtin_amdgcn_s_setprio(0);
            __syncthreads();
        }
    } else {
        load_tile(0);
        store_tile(0);
        if (BK < K) load_tile(BK);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < K; kt += BK) {
            compute(cur, 0);
            compute(cur, 1);
            if (kt + BK < K) store_tile(cur ^ 1);
            if (kt + 2 * BK < K) load_tile(kt + 2 * BK);
            compute(cur, 2);
            compute(cur, 3);
            __syncthreads();
            cur ^= 1;
        }
    }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int n = n0 + wn * 64 + tn * 32 + li;
        if (n >= N) continue;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) C[(size_t)m * N + n] = acc[tm][tn][r];
            }
    }
}

// V10: operands go global -> LDS directly (buffer_load ... lds, 16 bytes per lane: a wave fills 1 KB = 8 rows of 32 floats),
// no register staging and no ds_write; rows are unpadded (128 B) and the 16-byte chunks of a row are XOR-swizzled with the
// row index (the lane picks WHICH global chunk it fetches, the LDS position is fixed by the lane), two LDS buffers, one
// barrier per k tile.  Rows past M / N get the byte offset 0xFFFFFFFF (range-checked: zeros).
typedef __attribute__((address_space(3))) void lds_void;
__global__ __launch_bounds__(256) void gemm_dma_kernel(const float* __restrict__ A, const float* __restrict__ Wt, float* __restrict__ C, int M,
                                                       int N, int K) {
    const int nyb = gridDim.y;
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, seq = lin >> 3;
    const int m_blk = (seq / nyb) * 8 + xcd;
    const int m0 = m_blk * BM;
    const int n0 = (seq % nyb) * BN;
    if (n0 >= N || m0 >= M) return;
    constexpr int BK = 32;
    __shared__ __attribute__((aligned(1024))) float As[2][BM * BK];
    __shared__ __attribute__((aligned(1024))) float Bs[2][BN * BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt), 0, (int)0xFFFFFFF0u, 0x00020000);
    // this wave fills rows [32 wave, 32 wave + 32) of both tiles: 4 instructions of 8 rows each per operand
    unsigned a_off[4], b_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 32 * wave + 8 * i + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);            // global 16-byte chunk that belongs at LDS position lane & 7
        a_off[i] = m0 + row < M ? (unsigned)((size_t)(m0 + row) * K + 4 * chunk) * 4u : 0xFFFFFFFFu;
        b_off[i] = n0 + row < N ? (unsigned)((size_t)(n0 + row) * K + 4 * chunk) * 4u : 0xFFFFFFFFu;
    }
    auto load_tile = [&](int buf, int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lds_void* la = (lds_void*)(&As[buf][(32 * wave + 8 * i) * BK]);
            lds_void* lb = (lds_void*)(&Bs[buf][(32 * wave + 8 * i) * BK]);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, la, 16, a_off[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : a_off[i] + kt * 4u, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, lb, 16, b_off[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : b_off[i] + kt * 4u, 0, 0, 0);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto frag = [&](int buf, int q, float4 (&a)[2], float4 (&b)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ra_ = wm * 64 + t * 32 + li, rb_ = wn * 64 + t * 32 + li;
            a[t] = *reinterpret_cast<const float4*>(&As[buf][ra_ * BK + 4 * ((2 * q + lh) ^ (ra_ & 7))]);
            b[t] = *reinterpret_cast<const float4*>(&Bs[buf][rb_ * BK + 4 * ((2 * q + lh) ^ (rb_ & 7))]);
        }
    };
    auto mma = [&](const float4 (&a)[2], const float4 (&b)[2]) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].x, b[tn].x, acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].y, b[tn].y, acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].z, b[tn].z, acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].w, b[tn].w, acc[tm][tn], 0, 0, 0);
            }
    };
    load_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < K; kt += BK) {
        if (kt + BK < K) load_tile(cur ^ 1, kt + BK);
        float4 a0[2], b0[2], a1[2], b1[2];
        frag(cur, 0, a0, b0);
        frag(cur, 1, a1, b1);
        mma(a0, b0);
        frag(cur, 2, a0, b0);
        mma(a1, b1);
        frag(cur, 3, a1, b1);
        mma(a0, b0);
        mma(a1, b1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int n = n0 + wn * 64 + tn * 32 + li;
        if (n >= N) continue;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) C[(size_t)m * N + n] = acc[tm][tn][r];
            }
    }
}

double run_dma(const float* A, const float* Wt, float* C, int M, int N, int K, std::vector<float>* out = nullptr) {
    const int m_blocks = (M + BM - 1) / BM;
    dim3 grid((m_blocks + 7) / 8 * 8, (N + BN - 1) / BN);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(gemm_dma_kernel, grid, dim3(256), 0, 0, A, Wt, C, M, N, K);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(gemm_dma_kernel, grid, dim3(256), 0, 0, A, Wt, C, M, N, K);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    if (out) { out->resize((size_t)M * N); (void)hipMemcpy(out->data(), C, out->size() * 4, hipMemcpyDeviceToHost); }
    printf("V10 (global -> LDS direct) %6dx%5dx%5d: %8.1f us  %6.1f TFLOP/s (%s)\n", M, N, K, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12,
           hipGetErrorString(hipGetLastError()));
    return best;
}

template <int V>
double run(const float* A, const float* Wt, float* C, int M, int N, int K, std::vector<float>* out = nullptr) {
    const int m_blocks = (M + BM - 1) / BM;
    dim3 grid((m_blocks + 7) / 8 * 8, (N + BN - 1) / BN);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(gemm_kernel<V>, grid, dim3(256), 0, 0, A, Wt, C, M, N, K);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(gemm_kernel<V>, grid, dim3(256), 0, 0, A, Wt, C, M, N, K);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    if (out) { out->resize((size_t)M * N); (void)hipMemcpy(out->data(), C, out->size() * 4, hipMemcpyDeviceToHost); }
    printf("V%d %6dx%5dx%5d: %8.1f us  %6.1f TFLOP/s (%s)\n", V, M, N, K, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12,
           hipGetErrorString(hipGetLastError()));
    return best;
}

int main() {
    const int shapes[][3] = {{8192, 8192, 1024}, {64000, 256, 3072}, {64000, 1024, 256}, {1000, 200, 256}};
    for (int coarse = 0; coarse < 1; ++coarse)
    for (auto& s : shapes) {
        printf("%s operands\n", coarse ? "16-bit" : "full-mantissa");
        const int M = s[0], N = s[1], K = s[2];
        float *A, *Wt, *C;
        (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&Wt, (size_t)N * K * 4); (void)hipMalloc(&C, (size_t)M * N * 4);
        std::vector<float> ha((size_t)M * K), hw((size_t)N * K);
        unsigned x = 12345;
        // full-mantissa operands: the matrix pipes' power draw (and with it the clock) depends on the data
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; unsigned y = x; x = x * 1664525u + 1013904223u; return ((float)(y >> 8) + (float)(x >> 8) / 16777216.0f) / 16777216.0f - 0.5f; };
        for (auto& v : ha) v = coarse ? (float)((int)(rnd() * 65536.0f)) / 65536.0f : rnd();
        for (auto& v : hw) v = coarse ? (float)((int)(rnd() * 65536.0f)) / 65536.0f : rnd();
        (void)hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(Wt, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> c0, c1;
        run<0>(A, Wt, C, M, N, K, &c0);
        run<1>(A, Wt, C, M, N, K, &c1);
        run<6>(A, Wt, C, M, N, K, &c1);
        run<8>(A, Wt, C, M, N, K, &c1);
        run<7>(A, Wt, C, M, N, K, &c1);
        run<9>(A, Wt, C, M, N, K, &c1);
        {
            std::vector<float> c10;
            run_dma(A, Wt, C, M, N, K, &c10);
            double md10 = 0;
            for (size_t i = 0; i < c0.size(); i += 97) md10 = fmax(md10, fabs((double)c0[i] - c10[i]));
            printf("   max |V0 - V10| = %g\n", md10);
        }
        {
            double md7 = 0;
            for (size_t i = 0; i < c0.size(); i += 97) md7 = fmax(md7, fabs((double)c0[i] - c1[i]));
            printf("   max |V0 - V7| = %g\n", md7);
        }
        double md = 0;
        for (size_t i = 0; i < c0.size(); i += 97) md = fmax(md, fabs((double)c0[i] - c1[i]));
        printf("   max |V0 - V1| = %g\n", md);
        (void)hipFree(A); (void)hipFree(Wt); (void)hipFree(C);
    }
    return 0;
}
