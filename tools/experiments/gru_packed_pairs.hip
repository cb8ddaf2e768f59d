// Recurrent half of the CBHG bidirectional GRU (gfx950).
//
// Replaces tf.nn.bidirectional_dynamic_rnn(GRUCell fw, GRUCell bw) / CudnnGRU at reference
// tacotron/layers.py:560-592: both directions run over the FULL padded length from a zero
// state (no sequence_length), outputs concatenated [fw | bw].
//
// The input halves (x W_x + b for r, u, c and both directions) are one big MFMA GEMM done
// beforehand; this kernel is the strictly sequential part.  It is latency bound (T dependent
// steps of a 128 -> 384 mat-vec), so the design minimises the dependent chain of one step:
//   * one 1024-thread workgroup per (utterance, direction): 16 waves = 4 per SIMD hide the LDS and
//     transcendental latencies of each other;
//   * the recurrent weights live in VGPRs for the whole sequence; every thread owns a pair of adjacent
//     output columns and a slice of K (gates: 8 lanes x 16 elements, candidate: 16 lanes x 8), multiplies
//     with packed FMAs and the partial sums are combined with lane shuffles inside the wave (no LDS trip);
//   * the state lives in LDS in a padded layout whose eight K-slices fall into different banks;
//   * the next step's input projections are prefetched while the current step computes.
//
//   GRUCell [TF-1.8]       : [r|u] = sig(xg + h Wgh);  c = tanh(xc + (r*h) Wch);  h' = u h + (1-u) c
//   CudnnCompatibleGRUCell : c = tanh(xc + r * (h Wch + bch))
#include "tts_common.h"

namespace tts {

#define GRU_THREADS 1024
#define GRU_SPAD 20   // floats per 16-element slice of the state in LDS (16 + 4: the eight slices hit different banks)

typedef float gru_f2 __attribute__((ext_vector_type(2)));

// Sums over groups of 8 (16) consecutive lanes, valid in the FIRST lane of each group, with DPP operands
// (v_add_f32_dpp: no LDS round trip, unlike __shfl_xor, which hipcc lowers to ds_bpermute_b32 here).
template <int CTRL>
__device__ __forceinline__ float gru_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float gru_lane_sum8(float v) {
    v += gru_dpp<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
    v += gru_dpp<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
    v += gru_dpp<0x104>(v);   // row_shl:4: lane i takes lane i + 4
    return v;
}
__device__ __forceinline__ float gru_lane_sum16(float v) {
    v = gru_lane_sum8(v);
    v += gru_dpp<0x108>(v);   // row_shl:8
    return v;
}

// The step is bound by VALU issue (one instruction per ~4 cycles and SIMD, sixteen waves): every thread owns
// a PAIR of adjacent output columns and a slice of K, so that the multiply-accumulates are v_pk_fma_f32 on
// (column a, column b) with the state element broadcast -- half the instructions of the scalar form.
//   phase 1, gates:     128 column pairs x 8 slices of 16 state elements (16 packed FMAs per thread)
//   phase 2, candidate:  64 column pairs x 16 slices of 8 (4 packed multiplies r*h + 8 packed FMAs)
template <int H, bool CUDNN>
__global__ __launch_bounds__(GRU_THREADS) void bigru_kernel(const float* __restrict__ xproj, int xld,
                                                            const float* __restrict__ wrec,
                                                            float* __restrict__ out, int B, int T) {
    static_assert(H == 128, "thread mapping assumes 128 units");
    const int b = blockIdx.x;
    const int d = blockIdx.y;        // 0 = forward, 1 = backward
    const int tid = threadIdx.x;
    const int gp = tid >> 3, ks = tid & 7;     // phase 1: gate columns 2 gp, 2 gp + 1; state elements [16 ks, +16)
    const int cp = tid >> 4, kc = tid & 15;    // phase 2: candidate columns 2 cp, 2 cp + 1; elements [8 kc, +8)

    __shared__ __attribute__((aligned(16))) float hs[8 * GRU_SPAD];   // state, slice-padded
    __shared__ __attribute__((aligned(16))) float rs[8 * GRU_SPAD];   // reset gate, same layout
    __shared__ float us[H];

    const size_t wstride = (size_t)H * 2 * H + (size_t)H * H + (CUDNN ? H : 0);
    const float* wg_g = wrec + d * wstride;          // [H][2H]
    const float* wc_g = wg_g + (size_t)H * 2 * H;    // [H][H]
    const float* bch_g = wc_g + (size_t)H * H;       // [H] (cudnn)

    gru_f2 wg[16];   // Wgh[16 ks + i][2 gp, 2 gp + 1]
    gru_f2 wc[8];    // Wch[8 kc + i][2 cp, 2 cp + 1]
#pragma unroll
    for (int i = 0; i < 16; ++i) wg[i] = *reinterpret_cast<const gru_f2*>(wg_g + (size_t)(16 * ks + i) * (2 * H) + 2 * gp);
#pragma unroll
    for (int i = 0; i < 8; ++i) wc[i] = *reinterpret_cast<const gru_f2*>(wc_g + (size_t)(8 * kc + i) * H + 2 * cp);
    gru_f2 bch = {0.f, 0.f};
    if (CUDNN) bch = *reinterpret_cast<const gru_f2*>(bch_g + 2 * cp);

    if (tid < 8 * GRU_SPAD) { hs[tid] = 0.f; rs[tid] = 0.f; }
    __syncthreads();

    const float* xb = xproj + (size_t)b * T * xld + (size_t)d * 3 * H;
    float* ob = out + (size_t)b * T * 2 * H + (size_t)d * H;

    int t = d ? T - 1 : 0;
    const int dt = d ? -1 : 1;
    // lane ks == 0 of every gate pair / lane kc == 0 of every candidate pair owns the input term
    const gru_f2 zero2 = {0.f, 0.f};
    gru_f2 xg = (ks == 0) ? *reinterpret_cast<const gru_f2*>(xb + (size_t)t * xld + 2 * gp) : zero2;
    gru_f2 xc = (kc == 0) ? *reinterpret_cast<const gru_f2*>(xb + (size_t)t * xld + 2 * H + 2 * cp) : zero2;
    gru_f2 hreg = zero2;   // h[2 cp], h[2 cp + 1] (valid in the kc == 0 lanes)
    const float* hq = hs + ks * GRU_SPAD;                          // phase 1 slice
    const int o2 = (kc >> 1) * GRU_SPAD + (kc & 1) * 8;            // phase 2 slice (8 elements)
    const int hpos = (cp >> 3) * GRU_SPAD + 2 * (cp & 7);          // where h[2 cp], h[2 cp + 1] live

    for (int s = 0; s < T; ++s, t += dt) {
        gru_f2 xg_n = zero2, xc_n = zero2;
        if (s + 1 < T) {
            const size_t o = (size_t)(t + dt) * xld;
            if (ks == 0) xg_n = *reinterpret_cast<const gru_f2*>(xb + o + 2 * gp);
            if (kc == 0) xc_n = *reinterpret_cast<const gru_f2*>(xb + o + 2 * H + 2 * cp);
        }

        // ---- phase 1: gates
        gru_f2 a0 = zero2, a1 = zero2, a2 = zero2, a3 = zero2;   // four independent chains
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(hq + i);
            a0 = __builtin_elementwise_fma((gru_f2){hv.x, hv.x}, wg[i + 0], a0);
            a1 = __builtin_elementwise_fma((gru_f2){hv.y, hv.y}, wg[i + 1], a1);
            a2 = __builtin_elementwise_fma((gru_f2){hv.z, hv.z}, wg[i + 2], a2);
            a3 = __builtin_elementwise_fma((gru_f2){hv.w, hv.w}, wg[i + 3], a3);
        }
        const gru_f2 acc = (a0 + a1) + (a2 + a3);
        const float g0 = gru_lane_sum8(acc.x), g1 = gru_lane_sum8(acc.y);
        // this thread's eight state elements of phase 2, read before the state is republished
        const float4 h2a = *reinterpret_cast<const float4*>(hs + o2), h2b = *reinterpret_cast<const float4*>(hs + o2 + 4);
        gru_f2 pc = zero2;
#define GRU_CAND8(Q0, Q1, Q2, Q3, Q4, Q5, Q6, Q7)                                          \
    {                                                                                      \
        gru_f2 c0_ = (gru_f2){Q0, Q0} * wc[0], c1_ = (gru_f2){Q1, Q1} * wc[1];             \
        gru_f2 c2_ = (gru_f2){Q2, Q2} * wc[2], c3_ = (gru_f2){Q3, Q3} * wc[3];             \
        c0_ = __builtin_elementwise_fma((gru_f2){Q4, Q4}, wc[4], c0_);                     \
        c1_ = __builtin_elementwise_fma((gru_f2){Q5, Q5}, wc[5], c1_);                     \
        c2_ = __builtin_elementwise_fma((gru_f2){Q6, Q6}, wc[6], c2_);                     \
        c3_ = __builtin_elementwise_fma((gru_f2){Q7, Q7}, wc[7], c3_);                     \
        pc = (c0_ + c1_) + (c2_ + c3_);                                                    \
    }
        if (CUDNN) {
            // the candidate's recurrent part does not depend on r here: compute it in the same phase
            GRU_CAND8(h2a.x, h2a.y, h2a.z, h2a.w, h2b.x, h2b.y, h2b.z, h2b.w)
        }
        if (ks == 0) {
            const float ga = sigmoidf_(xg.x + g0), gb = sigmoidf_(xg.y + g1);
            const int c0 = 2 * gp;
            if (c0 < H) {
                *reinterpret_cast<gru_f2*>(rs + (c0 >> 4) * GRU_SPAD + (c0 & 15)) = (gru_f2){ga, gb};
            } else {
                *reinterpret_cast<gru_f2*>(us + c0 - H) = (gru_f2){ga, gb};
            }
        }
        __syncthreads();

        if (!CUDNN) {
            // phase 2 on r*h, formed on the fly (same products as an explicit r*h vector)
            const float4 r2a = *reinterpret_cast<const float4*>(rs + o2), r2b = *reinterpret_cast<const float4*>(rs + o2 + 4);
            const float q0 = r2a.x * h2a.x, q1 = r2a.y * h2a.y, q2 = r2a.z * h2a.z, q3 = r2a.w * h2a.w;
            const float q4 = r2b.x * h2b.x, q5 = r2b.y * h2b.y, q6 = r2b.z * h2b.z, q7 = r2b.w * h2b.w;
            GRU_CAND8(q0, q1, q2, q3, q4, q5, q6, q7)
        }
#undef GRU_CAND8
        const float p0 = gru_lane_sum16(pc.x), p1 = gru_lane_sum16(pc.y);
        if (kc == 0) {
            const int c0 = 2 * cp;
            float ca, cb;
            if (CUDNN) {
                const gru_f2 r2 = *reinterpret_cast<const gru_f2*>(rs + (c0 >> 4) * GRU_SPAD + (c0 & 15));
                ca = tanhf_(xc.x + r2.x * (p0 + bch.x));
                cb = tanhf_(xc.y + r2.y * (p1 + bch.y));
            } else {
                ca = tanhf_(xc.x + p0);
                cb = tanhf_(xc.y + p1);
            }
            const gru_f2 u2 = *reinterpret_cast<const gru_f2*>(us + c0);
            hreg = (gru_f2){u2.x * hreg.x + (1.0f - u2.x) * ca, u2.y * hreg.y + (1.0f - u2.y) * cb};
            *reinterpret_cast<gru_f2*>(ob + (size_t)t * 2 * H + c0) = hreg;
        }
        xg = xg_n;
        xc = xc_n;
        // every read of hs of this step happened before the barrier above: publish the new state
        if (kc == 0) *reinterpret_cast<gru_f2*>(hs + hpos) = hreg;
        __syncthreads();
    }
}

size_t bigru_wrec_floats(int H, int cudnn) {
    return 2 * ((size_t)H * 2 * H + (size_t)H * H + (cudnn ? H : 0));
}

hipError_t launch_bigru(hipStream_t s, const float* xproj, int xld, const float* wrec, float* out,
                        int B, int T, int H, int cudnn) {
    if (H != 128) return hipErrorInvalidValue;
    dim3 grid(B, 2);
    if (cudnn)
        hipLaunchKernelGGL((bigru_kernel<128, true>), grid, dim3(GRU_THREADS), 0, s, xproj, xld, wrec, out, B, T);
    else
        hipLaunchKernelGGL((bigru_kernel<128, false>), grid, dim3(GRU_THREADS), 0, s, xproj, xld, wrec, out, B, T);
    return hipGetLastError();
}

}  // namespace tts
