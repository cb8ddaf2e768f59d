// Luong-attention GRU decoder loop (gfx950).
//
// Replaces seq2seq.dynamic_decode(BasicDecoder(OutputProjectionWrapper(MultiRNNCell([
//   AttentionWrapper(PrenetWrapper(GRUCell), LuongAttention), Residual(GRUCell) x2])),
//   TacotronInferenceHelper)) -- reference tacotron/model.py:191-331, wrappers.py:94-124,
// helpers.py:83-110,161-205.  n_steps strictly sequential steps; within a step every layer
// depends on the previous one, rows (utterances) are independent.
//
// Each layer of a step is one small-M GEMM launch over all B rows: 16x16 output tiles, K
// split over the 4 waves of a workgroup (v_mfma_f32_16x16x4_f32, operands loaded straight
// from L2 as float4 -- weights are 6 MB and stay L2/Infinity-Cache resident), LDS reduction
// of the 4 partial tiles, fused GRU / activation epilogue.  The whole loop is captured into
// one hipGraph by the caller (api.hip) so the per-launch host cost disappears.
#include "tts_common.h"
#include "decoder.h"
#include <cstring>

namespace tts {

__global__ __launch_bounds__(256) void dec_gemm_kernel(DecGemm p) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int b0 = blockIdx.y * 16;

    __shared__ float red[4][16][17];

    const int row = b0 + r;
    const bool row_ok = row < p.B;
    const int rr = row_ok ? row : 0;
    const float* a0 = p.a0 + (size_t)rr * p.lda0;
    const float* a1 = p.a1 + (size_t)rr * p.lda1 - p.k0;   // indexed by absolute k
    const int n = n0 + r;
    const bool n_ok = n < p.N;
    const float* w = p.Wt + (size_t)(n_ok ? n : 0) * p.K;

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nchunks = p.K >> 4;
    // chunk c (16 consecutive k) belongs to wave c % 4; all of a wave's loads are issued before
    // its MFMAs so that one L2 round trip covers them (K <= 512 => one pass).
    for (int cb = 0; cb < nchunks; cb += 32) {
        float4 av[8], bv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = cb + wave + 4 * i;
            const int k = 16 * c + 4 * q;
            av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < nchunks) {
                if (row_ok) av[i] = *reinterpret_cast<const float4*>((k < p.k0 ? a0 : a1) + k);
                if (n_ok) bv[i] = *reinterpret_cast<const float4*>(w + k);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
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

    const int orow = tid >> 4, ocol = tid & 15;
    const int ob = b0 + orow, on = n0 + ocol;
    if (ob >= p.B || on >= p.N) return;
    float v = (red[0][orow][ocol] + red[1][orow][ocol]) + (red[2][orow][ocol] + red[3][orow][ocol]);
    if (p.bias) v += p.bias[on];

    switch (p.epi) {
        case DEC_EPI_ACT:
            p.out[(size_t)ob * p.ldo + on] = apply_act(v, p.act);
            break;
        case DEC_EPI_GRU_GATES: {
            // columns [r | u]; emits r*h (the candidate's second operand) and u
            const float gte = sigmoidf_(v);
            if (on < p.U) {
                p.rh[(size_t)ob * p.U + on] = gte * p.h[(size_t)ob * p.U + on];
            } else {
                p.u[(size_t)ob * p.U + on - p.U] = gte;
            }
        } break;
        case DEC_EPI_GRU_CAND: {
            const size_t i = (size_t)ob * p.U + on;
            const float c = tanhf_(v);
            const float u = p.u[i];
            const float hn = u * p.h[i] + (1.0f - u) * c;
            p.h[i] = hn;
            if (p.out) p.out[(size_t)ob * p.ldo + on] = p.resid ? p.resid[i] + hn : hn;
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

// CudnnCompatibleGRUCell tail: c = tanh(xi + r*hh); h' = u h + (1-u) c; y = resid + h'
__global__ void dec_gru_cudnn_combine(const float* r, const float* u, const float* hh, const float* xi,
                                      float* h, const float* resid, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c = tanhf_(xi[i] + r[i] * hh[i]);
    const float uu = u[i];
    const float hn = uu * h[i] + (1.0f - uu) * c;
    h[i] = hn;
    if (out) out[i] = resid ? resid[i] + hn : hn;
}

// Luong dot attention for one utterance per workgroup:
//   score_j = <q, keys_j>, a = softmax(score) over ALL Ts positions (no mask), ctx = sum_j a_j values_j
// (TF-1.8 _luong_score / _compute_attention; dot form documented at reference attention.py:396-400).
__global__ __launch_bounds__(256) void dec_attention_kernel(const float* __restrict__ query,  // [B][D]
                                                            const float* __restrict__ keys,   // [B][Ts][D]
                                                            const float* __restrict__ values, // [B][Ts][D]
                                                            float* __restrict__ align,        // [B][Ts] slice of step t
                                                            float* __restrict__ ctx,          // [B][D]
                                                            int Ts) {
    constexpr int D = 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* qs = smem;            // D
    float* sc = smem + D;        // Ts (rounded up)
    __shared__ float redm[4], reds[4];

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    qs[tid] = query[(size_t)b * D + tid];
    __syncthreads();

    // scores: 16 lanes per key, 4 keys per wave per iteration
    const int sub = lane >> 4, l16 = lane & 15;
    const float* kb = keys + (size_t)b * Ts * D;
    for (int j0 = 0; j0 < Ts; j0 += 16) {
        const int j = j0 + wave * 4 + sub;
        float s = 0.f;
        if (j < Ts) {
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
        if (j < Ts && l16 == 0) sc[j] = s;
    }
    __syncthreads();

    // softmax over Ts (max-subtracted)
    float m = -INFINITY;
    for (int j = tid; j < Ts; j += 256) m = fmaxf(m, sc[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) redm[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    float sum = 0.f;
    for (int j = tid; j < Ts; j += 256) {
        const float e = __expf(sc[j] - m);
        sc[j] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) reds[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / ((reds[0] + reds[1]) + (reds[2] + reds[3]));
    for (int j = tid; j < Ts; j += 256) {
        const float a = sc[j] * inv;
        sc[j] = a;
        if (align) align[(size_t)b * Ts + j] = a;
    }
    __syncthreads();

    // context: thread d accumulates over all positions (coalesced 1 KB rows)
    const float* vb = values + (size_t)b * Ts * D + tid;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    int j = 0;
    for (; j + 4 <= Ts; j += 4) {
        c0 = fmaf(sc[j + 0], vb[(size_t)(j + 0) * D], c0);
        c1 = fmaf(sc[j + 1], vb[(size_t)(j + 1) * D], c1);
        c2 = fmaf(sc[j + 2], vb[(size_t)(j + 2) * D], c2);
        c3 = fmaf(sc[j + 3], vb[(size_t)(j + 3) * D], c3);
    }
    for (; j < Ts; ++j) c0 = fmaf(sc[j], vb[(size_t)j * D], c0);
    ctx[(size_t)b * D + tid] = (c0 + c1) + (c2 + c3);
}

static inline hipError_t run_gemm(hipStream_t s, const DecGemm& p) {
    dim3 grid((p.N + 15) / 16, (p.B + 15) / 16);
    hipLaunchKernelGGL(dec_gemm_kernel, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

static DecGemm mk(const float* a0, int lda0, int k0, const float* a1, int lda1, const float* Wt,
                  const float* bias, int B, int N, int K) {
    DecGemm p;
    memset(&p, 0, sizeof(p));
    p.a0 = a0; p.lda0 = lda0; p.k0 = k0;
    p.a1 = a1 ? a1 : a0; p.lda1 = a1 ? lda1 : lda0;
    if (!a1) p.k0 = K;  // single segment
    p.Wt = Wt; p.bias = bias; p.B = B; p.N = N; p.K = K;
    return p;
}

// One GRU cell (both formulations).  x [B][in] (ldx), state h [B][U] updated in place,
// out [B][U] = (resid ? resid : 0) + h'.
static hipError_t run_gru(hipStream_t s, const DecoderWeights::Gru& g, const DecoderScratch& sc,
                          const float* x, int ldx, int n_in, float* h, const float* resid, float* out,
                          int B, int U, int cudnn) {
    hipError_t e;
    if (!cudnn) {
        DecGemm p = mk(x, ldx, n_in, h, U, g.gates_wt, g.gates_b, B, 2 * U, n_in + U);
        p.epi = DEC_EPI_GRU_GATES; p.U = U; p.h = h; p.rh = sc.rh; p.u = sc.u;
        if ((e = run_gemm(s, p)) != hipSuccess) return e;
        DecGemm c = mk(x, ldx, n_in, sc.rh, U, g.cand_wt, g.cand_b, B, U, n_in + U);
        c.epi = DEC_EPI_GRU_CAND; c.U = U; c.h = h; c.u = sc.u; c.resid = resid; c.out = out; c.ldo = U;
        return run_gemm(s, c);
    }
    DecGemm p = mk(x, ldx, n_in, h, U, g.gates_wt, g.gates_b, B, 4 * U, n_in + U);
    p.epi = DEC_EPI_GRU_CUDNN_PRE; p.U = U; p.rh = sc.rh; p.u = sc.u; p.hh = sc.hh; p.xi = sc.xi;
    if ((e = run_gemm(s, p)) != hipSuccess) return e;
    const int n = B * U;
    hipLaunchKernelGGL(dec_gru_cudnn_combine, dim3((n + 255) / 256), dim3(256), 0, s, sc.rh, sc.u, sc.hh,
                       sc.xi, h, resid, out, n);
    return hipGetLastError();
}

hipError_t decoder_enqueue(hipStream_t s, const DecoderWeights& w, const DecoderScratch& sc,
                           const float* memory, const float* keys, int B, int Ts, int n_steps,
                           float* mel, float* align, int cudnn) {
    const int A = w.att_units, U = w.dec_units, NM = w.n_mels, R = w.reduction;
    const int OUT = NM * R;
    const int P1 = w.prenet1_units, P2 = w.prenet2_units;
    hipError_t e;
    // zero states (attention, alignments irrelevant, cell states): TF zero_state
    if ((e = hipMemsetAsync(sc.state, 0, sc.state_bytes, s)) != hipSuccess) return e;
    const size_t lds_attn = (size_t)(256 + ((Ts + 3) & ~3)) * sizeof(float);
    const size_t mel_ld = (size_t)n_steps * OUT;
    for (int t = 0; t < n_steps; ++t) {
        // PrenetWrapper on concat([x_t, attention_{t-1}])   (wrappers.py:122-124)
        const float* x = t == 0 ? sc.zeros : mel + (size_t)(t - 1) * OUT + (OUT - NM);
        const int ldx = t == 0 ? 0 : (int)mel_ld;
        DecGemm p1 = mk(x, ldx, NM, sc.att, A, w.prenet1_wt, w.prenet1_b, B, P1, NM + A);
        p1.epi = DEC_EPI_ACT; p1.act = ACT_RELU; p1.out = sc.p1; p1.ldo = P1;
        if ((e = run_gemm(s, p1)) != hipSuccess) return e;
        DecGemm p2 = mk(sc.p1, P1, P1, nullptr, 0, w.prenet2_wt, w.prenet2_b, B, P2, P1);
        p2.epi = DEC_EPI_ACT; p2.act = ACT_RELU; p2.out = sc.p2; p2.ldo = P2;
        if ((e = run_gemm(s, p2)) != hipSuccess) return e;
        // attention GRU (no residual); output = new state
        if ((e = run_gru(s, w.att_gru, sc, sc.p2, P2, P2, sc.h_att, nullptr, nullptr, B, A, cudnn)) != hipSuccess)
            return e;
        // Luong attention with the new cell output as query
        hipLaunchKernelGGL(dec_attention_kernel, dim3(B), dim3(256), lds_attn, s, sc.h_att, keys, memory,
                           align ? align + (size_t)t * B * Ts : nullptr, sc.ctx, Ts);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        // attention_layer(concat([cell_output, context])), no bias
        DecGemm al = mk(sc.h_att, A, A, sc.ctx, w.mem_units, w.attn_layer_wt, nullptr, B, A, A + w.mem_units);
        al.epi = DEC_EPI_ACT; al.act = ACT_NONE; al.out = sc.att; al.ldo = A;
        if ((e = run_gemm(s, al)) != hipSuccess) return e;
        // residual GRU stack
        const float* y = sc.att;
        for (int l = 0; l < w.n_layers; ++l) {
            float* yo = (l & 1) ? sc.y1 : sc.y0;
            if ((e = run_gru(s, w.gru[l], sc, y, l == 0 ? A : U, l == 0 ? A : U, sc.h_dec[l], y, yo, B, U, cudnn)) != hipSuccess)
                return e;
            y = yo;
        }
        // OutputProjectionWrapper -> mel[:, t, :]
        DecGemm op = mk(y, U, U, nullptr, 0, w.out_wt, w.out_b, B, OUT, U);
        op.epi = DEC_EPI_ACT; op.act = ACT_NONE; op.out = mel + (size_t)t * OUT; op.ldo = (int)mel_ld;
        if ((e = run_gemm(s, op)) != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace tts
