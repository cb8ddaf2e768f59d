// Luong-attention GRU decoder loop (gfx950).
//
// Replaces seq2seq.dynamic_decode(BasicDecoder(OutputProjectionWrapper(MultiRNNCell([
//   AttentionWrapper(PrenetWrapper(GRUCell), LuongAttention), Residual(GRUCell) x2])),
//   TacotronInferenceHelper)) -- reference tacotron/model.py:191-331, wrappers.py:94-124,
// helpers.py:83-110,161-205.  n_steps strictly sequential steps; within a step every layer
// depends on the previous one, rows (utterances) are independent.  The loop is latency bound:
// what matters is the number and the length of the dependent launches per step.
//
//  * Each layer of a step is one small-M GEMM launch over all B rows: 16x16 output tiles, K split
//    over the 8 waves of a workgroup (v_mfma_f32_16x16x4_f32, operands loaded straight from L2 as
//    float4 -- the 6 MB of weights stay L2 / Infinity-Cache resident), LDS reduction of the 8
//    partial tiles, fused GRU / activation epilogue whose operands are fetched BEFORE the K loop.
//  * The output projection (OutputProjectionWrapper, model.py:273-277) is taken off the per-step
//    chain: the helper feeds back only the last n_mels outputs (helpers.py:200), so the next
//    step's pre-net input x_t W1x = (y W_o + b_o)[-80:] W1x is folded into one [U -> P1] matrix at
//    weight-load time; all 200 projections y_t W_o + b_o are one large GEMM after the loop.
//  * Attention: TTS_ATT_PARTS workgroups per utterance each score a slice of the memory, keep a
//    local (max, sum) and an unnormalised partial context; the attention-layer GEMM merges the
//    partials in its A loader (flash-decoding style), and the alignment history is normalised
//    once for all steps after the loop.
//  * The whole loop is captured into one hipGraph by the caller (api_stages.hip).
#include "tts_common.h"
#include "decoder.h"
#include <cstring>

namespace tts {

#define DEC_NW 8
#define DEC_THREADS (DEC_NW * 64)

template <bool PARTS>
__device__ __forceinline__ void dec_gemm_body(const DecGemm& p) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int b0 = blockIdx.y * 16;

    __shared__ float red[DEC_NW][16][17];

    // ---- epilogue operands first: their L2 round trip overlaps the operand loads below
    const int orow = (tid >> 4) & 15, ocol = tid & 15;
    const int ob = b0 + orow, on = n0 + ocol;
    const bool owner = tid < 256 && ob < p.B && on < p.N;
    float e_bias = 0.f, e_h = 0.f, e_u = 0.f, e_res = 0.f;
    if (owner) {
        if (p.bias) e_bias = p.bias[on];
        if (p.epi == DEC_EPI_GRU_GATES) {
            if (on < p.U) e_h = p.h[(size_t)ob * p.U + on];
        } else if (p.epi == DEC_EPI_GRU_CAND) {
            const size_t i = (size_t)ob * p.U + on;
            e_h = p.h[i];
            e_u = p.u[i];
            if (p.resid) e_res = p.resid[(size_t)ob * p.ldr + on];
        }
    }

    const int row = b0 + r;
    const bool row_ok = row < p.B;
    const int rr = row_ok ? row : 0;
    const float* a0 = p.a0 + (size_t)rr * p.lda0;
    const float* a1 = p.a1 ? p.a1 + (size_t)rr * p.lda1 - p.k0 : nullptr;   // indexed by absolute k
    const int n = n0 + r;
    const bool n_ok = n < p.N;
    const float* w = p.Wt + (size_t)(n_ok ? n : 0) * p.K;

    // attention-context merge coefficients for this lane's row
    float coef[TTS_ATT_PARTS];
    const float* part_row = nullptr;
    size_t part_stride = 0;
    if (PARTS) {
        const float* st = p.stats + (size_t)rr * TTS_ATT_PARTS * 2;
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < TTS_ATT_PARTS; ++i) m = fmaxf(m, st[2 * i]);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TTS_ATT_PARTS; ++i) {
            coef[i] = __expf(st[2 * i] - m);
            sum += coef[i] * st[2 * i + 1];
        }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < TTS_ATT_PARTS; ++i) coef[i] *= inv;
        part_row = p.parts + (size_t)rr * p.lda1 - p.k0;
        part_stride = (size_t)p.B * p.lda1;
    }

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nchunks = p.K >> 4;
    // chunk c (16 consecutive k) belongs to wave c % DEC_NW; all of a wave's loads are issued
    // before its MFMAs so that one L2 round trip covers them (K <= 512 => one pass of 4 chunks).
    for (int cb = 0; cb < nchunks; cb += 4 * DEC_NW) {
        float4 av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cb + wave + DEC_NW * i;
            const int k = 16 * c + 4 * q;
            av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < nchunks) {
                if (n_ok) bv[i] = *reinterpret_cast<const float4*>(w + k);
                if (row_ok) {
                    if (k < p.k0) {
                        av[i] = *reinterpret_cast<const float4*>(a0 + k);
                    } else if (PARTS) {
                        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int pp = 0; pp < TTS_ATT_PARTS; ++pp) {
                            const float4 t4 = *reinterpret_cast<const float4*>(part_row + pp * part_stride + k);
                            s4.x = fmaf(coef[pp], t4.x, s4.x);
                            s4.y = fmaf(coef[pp], t4.y, s4.y);
                            s4.z = fmaf(coef[pp], t4.z, s4.z);
                            s4.w = fmaf(coef[pp], t4.w, s4.w);
                        }
                        av[i] = s4;
                    } else {
                        av[i] = *reinterpret_cast<const float4*>(a1 + k);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bv[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bv[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, bv[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, bv[i].w, acc, 0, 0, 0);
        }
    }
    // C/D map of 16x16: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][q * 4 + i][r] = acc[i];
    __syncthreads();
    if (!owner) return;

    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < DEC_NW; ++wv) v += red[wv][orow][ocol];   // fixed order
    v += e_bias;

    switch (p.epi) {
        case DEC_EPI_ACT:
            p.out[(size_t)ob * p.ldo + on] = apply_act(v, p.act);
            break;
        case DEC_EPI_GRU_GATES: {
            // columns [r | u]; emits r*h (the candidate's second operand) and u
            const float gte = sigmoidf_(v);
            if (on < p.U) p.rh[(size_t)ob * p.U + on] = gte * e_h;
            else p.u[(size_t)ob * p.U + on - p.U] = gte;
        } break;
        case DEC_EPI_GRU_CAND: {
            const float c = tanhf_(v);
            const float hn = e_u * e_h + (1.0f - e_u) * c;
            p.h[(size_t)ob * p.U + on] = hn;
            if (p.out) p.out[(size_t)ob * p.ldo + on] = e_res + hn;
        } break;
        case DEC_EPI_GRU_CUDNN_PRE: {
            // columns [r | u | hh | xi]: gates on [x;h], hh = h Wch + bch, xi = x Wci + bci
            const int blk = on / p.U, j = on - blk * p.U;
            const size_t i = (size_t)ob * p.U + j;
            if (blk == 0) p.rh[i] = sigmoidf_(v);         // r (not yet multiplied)
            else if (blk == 1) p.u[i] = sigmoidf_(v);
            else if (blk == 2) p.hh[i] = v;
            else p.xi[i] = v;
        } break;
    }
}

// Two entry points so that the common form (no attention merge in the loader) can be held to 64 VGPRs: four
// 512-thread workgroups per CU instead of two, i.e. the 128 workgroups of the N = 512 layers fit the 32 CUs
// the call pipeline reserves for the decoder in one round.
__global__ __launch_bounds__(DEC_THREADS, 8) void dec_gemm_kernel(DecGemm p) { dec_gemm_body<false>(p); }
__global__ __launch_bounds__(DEC_THREADS) void dec_gemm_merge_kernel(DecGemm p) { dec_gemm_body<true>(p); }

// CudnnCompatibleGRUCell tail: c = tanh(xi + r*hh); h' = u h + (1-u) c; y = resid + h'
__global__ void dec_gru_cudnn_combine(const float* r, const float* u, const float* hh, const float* xi, float* h,
                                      const float* resid, int ldr, float* out, int ldo, int B, int U) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * U) return;
    const int b = i / U, j = i - b * U;
    const float c = tanhf_(xi[i] + r[i] * hh[i]);
    const float uu = u[i];
    const float hn = uu * h[i] + (1.0f - uu) * c;
    h[i] = hn;
    if (out) out[(size_t)b * ldo + j] = (resid ? resid[(size_t)b * ldr + j] : 0.f) + hn;
}

// Luong dot attention, one (utterance, slice of the memory) per workgroup:
//   score_j = <q, keys_j> for the slice; local max m, e_j = exp(score_j - m), local sum s = sum e_j,
//   unnormalised partial context sum_j e_j values_j.  softmax over ALL Ts positions (no mask) is
//   recovered by merging the parts: TF-1.8 _luong_score / _compute_attention; dot form documented
//   at reference attention.py:396-400.
__global__ __launch_bounds__(256) void dec_attention_kernel(const float* __restrict__ query,   // [B][D]
                                                            const float* __restrict__ keys,    // [B][Ts][D]
                                                            const float* __restrict__ values,  // [B][Ts][D]
                                                            float* __restrict__ e_out,         // [B][Ts] of this step
                                                            float* __restrict__ parts,         // [PARTS][B][D]
                                                            float* __restrict__ stats,         // [B][PARTS][2] of this step
                                                            int B, int Ts, int Tp, int w_lo, int w_n,
                                                            const float* __restrict__ p_row, int local_d) {
    constexpr int D = 256;
    __shared__ __attribute__((aligned(16))) float qs[D];
    extern __shared__ float sc[];   // Tp scores
    __shared__ float redm[4], reds[4];

    const int part = blockIdx.x;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // the scored positions are [w_lo, w_lo + w_n): the whole memory, or the local window -- whose start the
    // predictive mode derives per utterance from this step's predicted centre (clamped into the memory: a window
    // that leaves it has already raised the error flag, see dec_predict_centre_kernel)
    if (p_row) {
        const int c = (int)floorf(p_row[b]);
        w_lo = min(max(c - local_d, 0), Ts - w_n);
    }
    const int j_lo = w_lo + part * Tp;
    const int nj = min(Tp, w_lo + w_n - j_lo);   // may be <= 0 for an empty slice
    qs[tid] = query[(size_t)b * D + tid];
    __syncthreads();

    // scores: 16 lanes per key, 4 keys per wave per iteration
    const int sub = lane >> 4, l16 = lane & 15;
    const float* kb = keys + ((size_t)b * Ts + j_lo) * D;
    for (int j0 = 0; j0 < nj; j0 += 16) {
        const int j = j0 + wave * 4 + sub;
        float s = 0.f;
        if (j < nj) {
            const float* kr = kb + (size_t)j * D;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d0 = (l16 + 16 * i) * 4;
                const float4 kv = *reinterpret_cast<const float4*>(kr + d0);
                const float4 qv = *reinterpret_cast<const float4*>(qs + d0);
                s = fmaf(kv.x, qv.x, s);
                s = fmaf(kv.y, qv.y, s);
                s = fmaf(kv.z, qv.z, s);
                s = fmaf(kv.w, qv.w, s);
            }
        }
        s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 1);
        if (j < nj && l16 == 0) sc[j] = s;
    }
    __syncthreads();

    float m = -INFINITY;
    for (int j = tid; j < nj; j += 256) m = fmaxf(m, sc[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) redm[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    float sum = 0.f;
    for (int j = tid; j < nj; j += 256) {
        const float e = __expf(sc[j] - m);
        sc[j] = e;
        e_out[(size_t)b * Ts + j_lo + j] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) reds[wave] = sum;
    __syncthreads();
    if (tid == 0) {
        float* st = stats + ((size_t)b * TTS_ATT_PARTS + part) * 2;
        st[0] = m;                                            // -inf for an empty slice
        st[1] = (reds[0] + reds[1]) + (reds[2] + reds[3]);    // 0 for an empty slice
    }

    // partial context: thread d accumulates over the slice (coalesced 1 KB rows)
    const float* vb = values + ((size_t)b * Ts + j_lo) * D + tid;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    int j = 0;
    for (; j + 4 <= nj; j += 4) {
        c0 = fmaf(sc[j + 0], vb[(size_t)(j + 0) * D], c0);
        c1 = fmaf(sc[j + 1], vb[(size_t)(j + 1) * D], c1);
        c2 = fmaf(sc[j + 2], vb[(size_t)(j + 2) * D], c2);
        c3 = fmaf(sc[j + 3], vb[(size_t)(j + 3) * D], c3);
    }
    for (; j < nj; ++j) c0 = fmaf(sc[j], vb[(size_t)j * D], c0);
    parts[((size_t)part * B + b) * D + tid] = (c0 + c1) + (c2 + c3);
}

// LocalLuongAttention in PREDICTIVE mode (reference attention.py:246-258): per utterance
//   p = T_s * sigmoid( v_p^T tanh(W_p h) ),  window [floor(p) - D, floor(p) + D].
// One workgroup per utterance: thread n forms (h W_p)[n], the workgroup reduces v_p . tanh(.).  A window that
// leaves the memory is where the reference stops being well defined (its padding arithmetic, attention.py:288-304,
// fails at run time): the error flag is raised and the caller reports TTS_ERR_UNSUPPORTED.
__global__ __launch_bounds__(256) void dec_predict_centre_kernel(const float* __restrict__ query, const float* __restrict__ wp,
                                                                 const float* __restrict__ vp, float* __restrict__ p_out,
                                                                 int Ts, int local_d, int* __restrict__ err_flag) {
    constexpr int A = 256;
    __shared__ float qs[A];
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    qs[tid] = query[(size_t)b * A + tid];
    __syncthreads();
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int k = 0; k < A; k += 4) {
        a0 = fmaf(qs[k + 0], wp[(size_t)(k + 0) * A + tid], a0);
        a1 = fmaf(qs[k + 1], wp[(size_t)(k + 1) * A + tid], a1);
        a2 = fmaf(qs[k + 2], wp[(size_t)(k + 2) * A + tid], a2);
        a3 = fmaf(qs[k + 3], wp[(size_t)(k + 3) * A + tid], a3);
    }
    float v = tanhf_((a0 + a1) + (a2 + a3)) * vp[tid];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        const float p = (float)Ts * sigmoidf_((red[0] + red[1]) + (red[2] + red[3]));
        p_out[b] = p;
        const int c = (int)floorf(p);
        if (c - local_d < 0 || c + local_d + 1 > Ts) *err_flag = 1;
    }
}

// Window of LocalLuongAttention in MONOTONIC mode at decoder step t (reference attention.py:263-286):
// p = min(max(t, D), Ts - (D + 1)), window [p - D, p + D + 1).  Needs Ts >= 2D + 1 (checked by the caller).
__host__ __device__ static inline int local_center(int t, int Ts, int D) {
    int p = t > D ? t : D;
    const int hi = Ts - (D + 1);
    return p < hi ? p : hi;
}

// alignment_history[t][b][j] = e * exp(m_part - m) / sum, for all steps at once (off the critical path).
// local_d > 0: positions outside the step's window are 0 (the reference pads the 2D+1 window back to
// the memory length, attention.py:85-92) and, with `gaussian`, the window is weighted by
// exp(-(j - p)^2 / 2 * (D/2)^2) -- the reference's expression as written (attention.py:73-80); the
// context vector uses the UNweighted window softmax (attention.py:69-71).
__global__ void dec_align_finalize_kernel(const float* __restrict__ e, const float* __restrict__ stats,
                                          float* __restrict__ align, int B, int Ts, int Tp, int local_d,
                                          int gaussian, const float* __restrict__ p_hist) {
    const size_t tb = blockIdx.x;   // t * B + b
    int w_lo = 0;
    float pc = 0.f;   // window centre as the gaussian sees it: the step index, or the real-valued prediction
    if (local_d > 0) {
        if (p_hist) {
            pc = p_hist[tb];
            w_lo = min(max((int)floorf(pc) - local_d, 0), Ts - (2 * local_d + 1));
        } else {
            const int c = local_center((int)(tb / B), Ts, local_d);
            pc = (float)c;
            w_lo = c - local_d;
        }
    }
    const float gk = 0.5f * (0.5f * local_d) * (0.5f * local_d);
    const float* st = stats + tb * TTS_ATT_PARTS * 2;
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < TTS_ATT_PARTS; ++i) m = fmaxf(m, st[2 * i]);
    float w[TTS_ATT_PARTS];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TTS_ATT_PARTS; ++i) {
        w[i] = __expf(st[2 * i] - m);
        sum += w[i] * st[2 * i + 1];
    }
    const float inv = 1.0f / sum;
    for (int j = threadIdx.x; j < Ts; j += blockDim.x) {
        float a = 0.f;
        const int jw = j - w_lo;
        if (local_d == 0 || (jw >= 0 && jw <= 2 * local_d)) {
            const int part = jw / Tp;
            float wp = w[0];
#pragma unroll
            for (int i = 1; i < TTS_ATT_PARTS; ++i) wp = part == i ? w[i] : wp;
            a = e[tb * Ts + j] * wp * inv;
            if (local_d > 0 && gaussian) {
                const float dist = (float)j - pc;
                a *= __expf(-(dist * dist) * gk);
            }
        }
        align[tb * Ts + j] = a;
    }
}

static inline hipError_t run_gemm(hipStream_t s, const DecGemm& p) {
    dim3 grid((p.N + 15) / 16, (p.B + 15) / 16);
    if (p.parts) hipLaunchKernelGGL(dec_gemm_merge_kernel, grid, dim3(DEC_THREADS), 0, s, p);
    else hipLaunchKernelGGL(dec_gemm_kernel, grid, dim3(DEC_THREADS), 0, s, p);
    return hipGetLastError();
}

static DecGemm mk(const float* a0, int lda0, int k0, const float* a1, int lda1, const float* Wt,
                  const float* bias, int B, int N, int K) {
    DecGemm p;
    memset(&p, 0, sizeof(p));
    p.a0 = a0; p.lda0 = lda0; p.k0 = k0;
    p.a1 = a1; p.lda1 = lda1;
    if (!a1) p.k0 = K;  // single segment
    p.Wt = Wt; p.bias = bias; p.B = B; p.N = N; p.K = K;
    return p;
}

// One GRU cell (both formulations).  x [B][in] (ldx), state h [B][U] updated in place,
// out [B][U] (row stride ldo) = (resid ? resid : 0) + h'.
static hipError_t run_gru(hipStream_t s, const DecoderWeights::Gru& g, const DecoderScratch& sc,
                          const float* x, int ldx, int n_in, float* h, const float* resid, int ldr, float* out,
                          int ldo, int B, int U, int cudnn) {
    hipError_t e;
    if (!cudnn) {
        DecGemm p = mk(x, ldx, n_in, h, U, g.gates_wt, g.gates_b, B, 2 * U, n_in + U);
        p.epi = DEC_EPI_GRU_GATES; p.U = U; p.h = h; p.rh = sc.rh; p.u = sc.u;
        if ((e = run_gemm(s, p)) != hipSuccess) return e;
        DecGemm c = mk(x, ldx, n_in, sc.rh, U, g.cand_wt, g.cand_b, B, U, n_in + U);
        c.epi = DEC_EPI_GRU_CAND; c.U = U; c.h = h; c.u = sc.u; c.resid = resid; c.ldr = ldr; c.out = out; c.ldo = ldo;
        return run_gemm(s, c);
    }
    DecGemm p = mk(x, ldx, n_in, h, U, g.gates_wt, g.gates_b, B, 4 * U, n_in + U);
    p.epi = DEC_EPI_GRU_CUDNN_PRE; p.U = U; p.rh = sc.rh; p.u = sc.u; p.hh = sc.hh; p.xi = sc.xi;
    if ((e = run_gemm(s, p)) != hipSuccess) return e;
    const int n = B * U;
    hipLaunchKernelGGL(dec_gru_cudnn_combine, dim3((n + 255) / 256), dim3(256), 0, s, sc.rh, sc.u, sc.hh, sc.xi, h,
                       resid, ldr, out, ldo, B, U);
    return hipGetLastError();
}

hipError_t decoder_enqueue(hipStream_t s, const DecoderWeights& w, const DecoderScratch& sc,
                           const float* memory, const float* keys, int B, int Ts, int n_steps,
                           float* align, int cudnn) {
    const int A = w.att_units, U = w.dec_units, NM = w.n_mels;
    const int P1 = w.prenet1_units, P2 = w.prenet2_units;
    const int LD = w.local_d;                       // 0: global LuongAttention
    const int Wn = LD > 0 ? 2 * LD + 1 : Ts;        // scored positions per step
    if (LD > 0 && Ts < Wn) return hipErrorInvalidValue;
    const int Tp = (Wn + TTS_ATT_PARTS - 1) / TTS_ATT_PARTS;
    const int yld = n_steps * U;
    hipError_t e;
    // zero states (attention, cell states): TF zero_state
    if ((e = hipMemsetAsync(sc.state, 0, sc.state_bytes, s)) != hipSuccess) return e;
    if (LD > 0 && w.local_predictive && (e = hipMemsetAsync(sc.err_flag, 0, sizeof(int), s)) != hipSuccess) return e;
    float* e_buf = sc.align_raw;
    for (int t = 0; t < n_steps; ++t) {
        // PrenetWrapper on concat([x_t, attention_{t-1}])   (wrappers.py:122-124)
        DecGemm p1;
        if (t == 0) {   // GO frame: zeros (helpers.py:108)
            p1 = mk(sc.zeros, 0, NM, sc.att, A, w.prenet1_wt, w.prenet1_b, B, P1, NM + A);
        } else {        // x_t = (y_{t-1} W_o + b_o)[-n_mels:], folded into the pre-net matrix
            p1 = mk(sc.yhist + (size_t)(t - 1) * U, yld, U, sc.att, A, w.prenet1f_wt, w.prenet1f_b, B, P1, U + A);
        }
        p1.epi = DEC_EPI_ACT; p1.act = ACT_RELU; p1.out = sc.p1; p1.ldo = P1;
        if ((e = run_gemm(s, p1)) != hipSuccess) return e;
        DecGemm p2 = mk(sc.p1, P1, P1, nullptr, 0, w.prenet2_wt, w.prenet2_b, B, P2, P1);
        p2.epi = DEC_EPI_ACT; p2.act = ACT_RELU; p2.out = sc.p2; p2.ldo = P2;
        if ((e = run_gemm(s, p2)) != hipSuccess) return e;
        // attention GRU (no residual); output = new state
        if ((e = run_gru(s, w.att_gru, sc, sc.p2, P2, P2, sc.h_att, nullptr, 0, nullptr, 0, B, A, cudnn)) != hipSuccess)
            return e;
        // Luong attention with the new cell output as query
        // (LocalLuongAttention: AdvancedAttentionWrapper hands the step index to the mechanism,
        //  reference attention.py:563, which then scores only the window around it)
        float* stats_t = sc.att_stats + (size_t)t * B * TTS_ATT_PARTS * 2;
        const int w_lo = LD > 0 ? local_center(t, Ts, LD) - LD : 0;
        const float* p_row = nullptr;
        if (LD > 0 && w.local_predictive) {
            float* p_t = sc.p_hist + (size_t)t * B;
            hipLaunchKernelGGL(dec_predict_centre_kernel, dim3(B), dim3(256), 0, s, sc.h_att, w.local_wp, w.local_vp, p_t,
                               Ts, LD, sc.err_flag);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            p_row = p_t;
        }
        hipLaunchKernelGGL(dec_attention_kernel, dim3(TTS_ATT_PARTS, B), dim3(256), (size_t)Tp * sizeof(float), s,
                           sc.h_att, keys, memory, e_buf + (size_t)t * B * Ts, sc.ctx_parts, stats_t, B, Ts, Tp, w_lo,
                           Wn, p_row, LD);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        // attention_layer(concat([cell_output, context])), no bias; context merged from the parts
        DecGemm al = mk(sc.h_att, A, A, sc.ctx_parts, w.mem_units, w.attn_layer_wt, nullptr, B, A, A + w.mem_units);
        al.parts = sc.ctx_parts; al.stats = stats_t;
        al.epi = DEC_EPI_ACT; al.act = ACT_NONE; al.out = sc.att; al.ldo = A;
        if ((e = run_gemm(s, al)) != hipSuccess) return e;
        // residual GRU stack; the top layer writes straight into the y history
        const float* y = sc.att;
        int ldy = A;
        for (int l = 0; l < w.n_layers; ++l) {
            const bool top = l == w.n_layers - 1;
            float* yo = top ? sc.yhist + (size_t)t * U : ((l & 1) ? sc.y1 : sc.y0);
            const int ldo = top ? yld : U;
            if ((e = run_gru(s, w.gru[l], sc, y, ldy, l == 0 ? A : U, sc.h_dec[l], y, ldy, yo, ldo, B, U, cudnn)) !=
                hipSuccess)
                return e;
            y = yo;
            ldy = ldo;
        }
    }
    if (align) {
        hipLaunchKernelGGL(dec_align_finalize_kernel, dim3(n_steps * B), dim3(64), 0, s, e_buf, sc.att_stats, align, B,
                           Ts, Tp, LD, w.local_gaussian, (LD > 0 && w.local_predictive) ? sc.p_hist : nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace tts
