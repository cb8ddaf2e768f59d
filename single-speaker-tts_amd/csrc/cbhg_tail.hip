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
// by then), the three GRU passes store to the projection buffer.  Round 4: the products run on the bf16 matrix pipe
// at f32 accuracy, as in gemm_f32.hip: a k-chunk of the tile (read from its f32 rows in LDS) and of the weights is split
// exactly into three bf16 terms when it is staged, [split][row][4 chunks of 8 bf16] swizzled, and a 16-deep step of a 32 x 32
// block is six v_mfma_f32_32x32x16_bf16 (the kernel was bound by the f32 MFMA rate: 62.9 MFLOP per workgroup = 117 us at
// 256 flop per cycle).  One image per operand, two workgroup barriers per chunk; the next chunk's weights are in flight in
// registers meanwhile.  (The round-3 form on v_mfma_f32_32x32x2_f32 is in git history up to round 5.)
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

// the tile's f32 rows, then the bf16 images of one k-chunk: A [3][128][64 B], B [3][256][64 B]
#define CT_AIMG (3 * CT_BM * 64)
#define CT_BIMG (3 * CT_BN * 64)
size_t cbhg_tail_lds_bytes() { return (size_t)CT_BM * CT_XLD * sizeof(float) + CT_AIMG + CT_BIMG; }
typedef __bf16 ct_bf16x8_t __attribute__((ext_vector_type(8)));
// x = hi + mid + lo exactly, each a bf16 (the top 16 bits of an f32): bits of x, of x - hi, of x - hi - mid (gemm_f32.hip)
__device__ __forceinline__ void ct_split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x);
    const float r1 = x - __uint_as_float(h & 0xFFFF0000u);
    m = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = __float_as_uint(r2);
}
__device__ __forceinline__ unsigned ct_pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
// four consecutive k (quad kq of the chunk's eight) of one row -> 8 bytes per split; `rows` rows per split
__device__ __forceinline__ void ct_store_split4(unsigned char* img, int rows, int row, int kq, float4 v) {
    unsigned h[4], m[4], l[4];
    ct_split3(v.x, h[0], m[0], l[0]);
    ct_split3(v.y, h[1], m[1], l[1]);
    ct_split3(v.z, h[2], m[2], l[2]);
    ct_split3(v.w, h[3], m[3], l[3]);
    const int off = (row * 4 + ((kq >> 1) ^ ((row >> 2) & 3))) * 16 + (kq & 1) * 8;
    *reinterpret_cast<uint2*>(img + off) = make_uint2(ct_pack_hi(h[0], h[1]), ct_pack_hi(h[2], h[3]));
    *reinterpret_cast<uint2*>(img + rows * 64 + off) = make_uint2(ct_pack_hi(m[0], m[1]), ct_pack_hi(m[2], m[3]));
    *reinterpret_cast<uint2*>(img + 2 * rows * 64 + off) = make_uint2(ct_pack_hi(l[0], l[1]), ct_pack_hi(l[2], l[3]));
}

bool cbhg_tail_supports(int c_in, int units, int gru_units, int n_hw, long long M) {
    // (32-bit byte offsets into the projection buffer)
    return units == CT_U && gru_units == 128 && c_in >= 4 && c_in <= CT_U && c_in % 4 == 0 && n_hw >= 0 && n_hw <= CBHG_TAIL_MAX_HW &&
           M >= 1 && (M + CT_BM) * 3 * CT_BN * (long long)sizeof(float) < 0xFFFFFFF0ll;
}

__global__ __launch_bounds__(CT_THREADS) void cbhg_tail_kernel(CbhgTailParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;                          // [128][CT_XLD]
    unsigned char* Ai = reinterpret_cast<unsigned char*>(lds + CT_BM * CT_XLD);   // [3][128][64 B]
    unsigned char* Bi = Ai + CT_AIMG;                                              // [3][256][64 B]
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
    // chunk kc of the weights (registers) and of the tile (its f32 rows in LDS) -> the two bf16 images
    auto stage_chunk = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ct_store_split4(Bi, CT_BN, (tid >> 3) + 64 * i, kq, rb[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 3) + 64 * i;
            ct_store_split4(Ai, CT_BM, row, kq, *reinterpret_cast<const float4*>(&Xs[row * CT_XLD + kc * CT_BK + 4 * kq]));
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
            __syncthreads();   // the images are free (the previous chunk's MFMAs have read them); the tile's rows are written
            stage_chunk(kc);
            __syncthreads();
            if (kc + 1 < nch) load_chunk(j, kc + 1);
            else if (j + 1 < n_jobs) load_chunk(j + 1, 0);
            if (live) {
                // two 16-deep steps: lane (li, lh) of a 32-row block holds k = 16 q + 8 lh .. + 7 of row li: chunk 2 q + lh
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int ch = 2 * q + lh;
                    uint4 fa[2][3];
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm) {
                        const int row = wm * 64 + tm * 32 + li;
                        const int off = (row * 4 + (ch ^ ((row >> 2) & 3))) * 16;
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) fa[tm][sp] = *reinterpret_cast<const uint4*>(Ai + sp * CT_BM * 64 + off);
                    }
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        const int row = wn * 64 + tn * 32 + li;
                        const int off = (row * 4 + (ch ^ ((row >> 2) & 3))) * 16;
                        uint4 fb[3];
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) fb[sp] = *reinterpret_cast<const uint4*>(Bi + sp * CT_BN * 64 + off);
#pragma unroll
                        for (int tm = 0; tm < 2; ++tm) {
#define CT_MMA(SA, SB)                                                                                                   \
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ct_bf16x8_t, fa[tm][SA]), \
                                                                                  __builtin_bit_cast(ct_bf16x8_t, fb[SB]), acc[tm][tn], 0, 0, 0);
                            CT_MMA(0, 2) CT_MMA(2, 0) CT_MMA(1, 1) CT_MMA(0, 1) CT_MMA(1, 0) CT_MMA(0, 0)
#undef CT_MMA
                        }
                    }
                }
            }
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
