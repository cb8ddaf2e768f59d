// The tail of a CBHG between the projections' residual and the bi-GRU (gfx950), as ONE launch:
//   lifter Dense + relu            reference tacotron/layers.py:546-553 (highway_network's units != input width)
//   n highway layers               reference tacotron/layers.py:236-259 (y = relu(xW_H + b_H) t + x (1 - t), t = sigmoid(xW_T + b_T))
//   GRU input projections          the x-halves of both directions' gate and candidate matrices (reference layers.py:560-594,
//                                  gru.hip consumes them as [r | u | c] per direction)
// As separate GEMMs (gemm_f32.hip) these are n + 2 launches whose 128-float rows make a round trip through HBM between
// each pair, with K = 128 deep tiles that never reach the matrix pipes' rate.  Here a workgroup keeps its 128 rows in
// LDS for the whole chain: every stage is a 128 x 256 x 128 GEMM whose A operand is that tile and whose weights stream
// through a double-buffered LDS image in k-chunks of 32 (all workgroups read the same 0.9 MB of weights: L2 hits); a
// highway stage writes its output back into the tile (between two workgroup barriers: every wave has read the old rows
// by then), the three GRU passes store to the projection buffer.  Same MFMA step, LDS image and k permutation as
// gemm_f32.hip (v_mfma_f32_32x32x2_f32, [row][32 + 4] floats, one ds_read_b128 per operand per four MFMAs).
// 512 threads = 8 waves (2 x 4), a wave owns 64 rows x 64 columns; the highway packing puts the 32 H columns and the 32 T
// columns of the same units in one 64-column span, so the gate mix is lane-local.
#include "tts_common.h"

namespace tts {

#define CT_BM 128
#define CT_U 128                 // highway units = width of the tile
#define CT_XLD (CT_U + 4)        // LDS row stride of the activation tile
#define CT_BK 32
#define CT_BLD (CT_BK + 4)
#define CT_BN 256
#define CT_THREADS 512

size_t cbhg_tail_lds_bytes() { return ((size_t)CT_BM * CT_XLD + 2 * (size_t)CT_BN * CT_BLD) * sizeof(float); }

bool cbhg_tail_supports(int c_in, int units, int gru_units, int n_hw, long long M) {
    // (32-bit byte offsets into the projection buffer)
    return units == CT_U && gru_units == 128 && c_in >= 4 && c_in <= CT_U && c_in % 4 == 0 && n_hw >= 0 && n_hw <= CBHG_TAIL_MAX_HW &&
           M >= 1 && (M + CT_BM) * 3 * CT_BN * (long long)sizeof(float) < 0xFFFFFFF0ll;
}

__global__ __launch_bounds__(CT_THREADS) void cbhg_tail_kernel(CbhgTailParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;                          // [128][CT_XLD]
    float* Bs = lds + CT_BM * CT_XLD;         // [2][256][CT_BLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * CT_BM;
    const int M = p.M;

    // ---- the rows: [128][c_in] -> LDS, zero beyond c_in (the lifter's K is padded to a multiple of 32) and beyond M
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)0xFFFFFFF0u, 0x00020000);
#pragma unroll
        for (int i = 0; i < (CT_BM * (CT_U / 4)) / CT_THREADS; ++i) {
            const int idx = tid + CT_THREADS * i;
            const int row = idx >> 5, c4 = idx & 31;
            const bool ok = m0 + row < M && 4 * c4 < p.c_in;
            const unsigned off = ok ? (unsigned)((size_t)(m0 + row) * p.ldx + 4 * c4) * 4u : 0xFFFFFFFFu;
            const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
            *reinterpret_cast<float4*>(&Xs[row * CT_XLD + 4 * c4]) = v;
        }
    }

    // ---- the chain as a flat sequence of (job, k-chunk) steps; the weights of the next step are in flight during the
    // MFMAs of this one, across job boundaries too
    const int n_jobs = 1 + p.n_hw + 3;
    const int kq = tid & 7;
    float4 rb[4];
    auto job_w = [&](int j) -> const float* {
        return j == 0 ? p.lifter_wt : (j <= p.n_hw ? p.hw_wt[j - 1] : p.gru_wt + (size_t)(j - 1 - p.n_hw) * CT_BN * CT_U);
    };
    auto job_k = [&](int j) { return j == 0 ? p.c_in : CT_U; };
    auto job_n = [&](int j) { return j == 0 ? CT_U : CT_BN; };
    auto load_chunk = [&](int j, int kc) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(job_w(j)), 0, (int)0xFFFFFFF0u, 0x00020000);
        const int K = job_k(j), N = job_n(j);
        const int kk = kc * CT_BK + 4 * kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + 64 * i;
            const bool ok = row < N && kk < K;
            rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (int)((unsigned)(row * K + kk) * 4u) : -1, 0, 0));
        }
    };
    auto store_chunk = [&](int buf) {
        float* B = Bs + buf * (CT_BN * CT_BLD);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + 64 * i;
            *reinterpret_cast<float4*>(&B[row * CT_BLD + 4 * kq]) = rb[i];
        }
    };

    // Outputs leave through raw buffer stores whose resource ends with row M - 1: one 32-bit offset register per lane plus a
    // compile-time constant per element (as 64-bit addresses the 64 stores of an epilogue held 128 registers), and rows
    // past M are dropped by the range check.
    const __amdgpu_buffer_rsrc_t xp_rs = __builtin_amdgcn_make_buffer_rsrc(
        p.xproj, 0, (int)(unsigned)((size_t)M * 3 * CT_BN * sizeof(float) < 0xFFFFFFF0ull ? (size_t)M * 3 * CT_BN * sizeof(float) : 0xFFFFFFF0ull), 0x00020000);
    const __amdgpu_buffer_rsrc_t hw_rs = __builtin_amdgcn_make_buffer_rsrc(
        p.hw_out, 0, p.hw_out ? (int)(unsigned)((size_t)M * CT_U * sizeof(float)) : 0, 0x00020000);
    const int row_l = wm * 64 + 4 * lh;   // + tm * 32 + (r & 3) + 8 * (r >> 2)

    f32x16 acc[2][2];
    int buf = 0;
    load_chunk(0, 0);
    for (int j = 0; j < n_jobs; ++j) {
        const int nch = (job_k(j) + CT_BK - 1) / CT_BK;
        const bool live = 64 * wn < job_n(j);   // the lifter has 128 columns: the waves of the upper two spans only help loading
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int kc = 0; kc < nch; ++kc) {
            store_chunk(buf);
            __syncthreads();   // (the first one also covers the tile's rows, a job's first one its updated rows)
            if (kc + 1 < nch) load_chunk(j, kc + 1);
            else if (j + 1 < n_jobs) load_chunk(j + 1, 0);
            if (live) {
                const float* B = Bs + buf * (CT_BN * CT_BLD);
                auto frag = [&](int q, float4 (&a)[2], float4 (&b)[2]) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        a[t] = *reinterpret_cast<const float4*>(&Xs[(wm * 64 + t * 32 + li) * CT_XLD + kc * CT_BK + 8 * q + 4 * lh]);
                        b[t] = *reinterpret_cast<const float4*>(&B[(wn * 64 + t * 32 + li) * CT_BLD + 8 * q + 4 * lh]);
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
                float4 a0[2], b0[2], a1[2], b1[2];
                frag(0, a0, b0);
                frag(1, a1, b1);
                mma(a0, b0);
                frag(2, a0, b0);
                mma(a1, b1);
                frag(3, a1, b1);
                mma(a0, b0);
                mma(a1, b1);
            }
            buf ^= 1;
        }

        // ---- epilogue.  C/D map of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        if (j == 0) {                       // lifter: relu(acc + b) -> the tile
            if (live) {
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const float b = p.lifter_b[wn * 64 + tn * 32 + li];
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[tm][tn][r] = fmaxf(acc[tm][tn][r] + b, 0.f);
                }
            }
            __syncthreads();                // every wave has read the input rows
            if (live) {
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ro = tm * 32 + (r & 3) + 8 * (r >> 2);
                            Xs[(row_l + ro) * CT_XLD + wn * 64 + tn * 32 + li] = acc[tm][tn][r];
                        }
            }
            // (visible to the other waves after the first chunk barrier of the next job)
        } else if (j <= p.n_hw) {           // highway: acc[.][0] = H, acc[.][1] = T of unit 32 wn + li
            const int unit = wn * 32 + li;
            const float bh = p.hw_b[j - 1][wn * 64 + li], bt = p.hw_b[j - 1][wn * 64 + 32 + li];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = tm * 32 + (r & 3) + 8 * (r >> 2);
                    const float hh = fmaxf(acc[tm][0][r] + bh, 0.f);
                    const float tt = sigmoidf_(acc[tm][1][r] + bt);
                    const float x = Xs[(row_l + ro) * CT_XLD + unit];
                    acc[tm][0][r] = hh * tt + x * (1.0f - tt);
                }
            __syncthreads();                // every wave has read the old rows
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = tm * 32 + (r & 3) + 8 * (r >> 2);
                    Xs[(row_l + ro) * CT_XLD + unit] = acc[tm][0][r];
                }
            if (j == p.n_hw) {
                const unsigned vb = (unsigned)((m0 + row_l) * CT_U + unit) * 4u;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = tm * 32 + (r & 3) + 8 * (r >> 2);
                        const float y = acc[tm][0][r];   // (a copy: __builtin_bit_cast of a vector ELEMENT reads element 0)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), hw_rs, (int)(vb + (unsigned)(ro * CT_U * 4)), 0, 0);
                    }
            }
        } else {                            // GRU input projections, 256 columns per pass
            const int pass = j - 1 - p.n_hw;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int n = pass * CT_BN + wn * 64 + tn * 32 + li;
                const float b = p.gru_b[n];
                const unsigned vb = (unsigned)((m0 + row_l) * (3 * CT_BN) + n) * 4u;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = tm * 32 + (r & 3) + 8 * (r >> 2);
                        const float y = acc[tm][tn][r] + b;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), xp_rs,
                                                              (int)(vb + (unsigned)(ro * 3 * CT_BN * 4)), 0, 0);
                    }
            }
        }
    }
    // n_hw == 0: the lifter's output is the highway stack's output
    if (p.n_hw == 0 && p.hw_out) {
        __syncthreads();
        for (int idx = tid; idx < CT_BM * (CT_U / 4); idx += CT_THREADS) {
            const int row = idx >> 5, c4 = idx & 31;
            if (m0 + row < M)
                *reinterpret_cast<float4*>(&p.hw_out[(size_t)(m0 + row) * CT_U + 4 * c4]) = *reinterpret_cast<const float4*>(&Xs[row * CT_XLD + 4 * c4]);
        }
    }
}

hipError_t cbhg_tail_configure() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&cbhg_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)cbhg_tail_lds_bytes());
}

hipError_t launch_cbhg_tail(hipStream_t s, const CbhgTailParams& p) {
    if (p.M < 1) return hipSuccess;
    hipLaunchKernelGGL(cbhg_tail_kernel, dim3((p.M + CT_BM - 1) / CT_BM), dim3(CT_THREADS), cbhg_tail_lds_bytes(), s, p);
    return hipGetLastError();
}

}  // namespace tts
