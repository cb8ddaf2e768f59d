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
//   * the recurrent weights live in VGPRs for the whole sequence; every gate column is split
//     over 4 lanes (K/4 = 32 FMAs each), every candidate column over 8 lanes (16 FMAs each), and
//     the partial sums are combined with DPP shuffles inside the wave (no LDS round trip);
//   * the state lives in LDS in a padded layout whose four K-quarters fall into different banks;
//   * the next step's input projections are prefetched while the current step computes.
//
//   GRUCell [TF-1.8]       : [r|u] = sig(xg + h Wgh);  c = tanh(xc + (r*h) Wch);  h' = u h + (1-u) c
//   CudnnCompatibleGRUCell : c = tanh(xc + r * (h Wch + bch))
#include "tts_common.h"

namespace tts {

#define GRU_THREADS 1024
#define GRU_QPAD 36   // floats per K-quarter of the state in LDS (32 + 4: quarters hit different banks)

// Lane sums with DPP operands (v_add_f32_dpp, no LDS round trip: hipcc lowers __shfl_xor to ds_bpermute_b32, a
// dependent LDS access per reduction level).  After gru_sum4 every lane of a quad holds the quad's sum; gru_sum8
// is valid in the first lane of each group of eight.
template <int CTRL>
__device__ __forceinline__ float gru_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float gru_sum4(float v) {
    v += gru_dpp<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
    v += gru_dpp<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
    return v;
}
__device__ __forceinline__ float gru_sum8(float v) {
    v = gru_sum4(v);
    v += gru_dpp<0x104>(v);   // row_shl:4: lane i takes lane i + 4
    return v;
}

template <int H, bool CUDNN>
__global__ __launch_bounds__(GRU_THREADS) void bigru_kernel(const float* __restrict__ xproj, int xld,
                                                            const float* __restrict__ wrec,
                                                            float* __restrict__ out, int B, int T) {
    static_assert(H == 128, "thread mapping assumes 128 units");
    const int b = blockIdx.x;
    const int d = blockIdx.y;        // 0 = forward, 1 = backward
    const int tid = threadIdx.x;
    // phase 1 mapping: gate column gc (0..2H-1), K quarter kq
    const int gc = tid >> 2, kq = tid & 3;
    // phase 2 mapping: candidate column cc (0..H-1), K eighth ke
    const int cc = tid >> 3, ke = tid & 7;

    __shared__ __attribute__((aligned(16))) float hs[4 * GRU_QPAD];    // state, quarter-padded
    __shared__ __attribute__((aligned(16))) float rhs[4 * GRU_QPAD];   // r*h (GRUCell) for phase 2
    __shared__ float us[H];
    __shared__ float rs[H];

    const size_t wstride = (size_t)H * 2 * H + (size_t)H * H + (CUDNN ? H : 0);
    const float* wg_g = wrec + d * wstride;          // [H][2H]
    const float* wc_g = wg_g + (size_t)H * 2 * H;    // [H][H]
    const float* bch_g = wc_g + (size_t)H * H;       // [H] (cudnn)

    float wg[32];   // Wgh[32 kq + i][gc]
    float wc[16];   // Wch[16 ke + i][cc]
#pragma unroll
    for (int i = 0; i < 32; ++i) wg[i] = wg_g[(size_t)(32 * kq + i) * (2 * H) + gc];
#pragma unroll
    for (int i = 0; i < 16; ++i) wc[i] = wc_g[(size_t)(16 * ke + i) * H + cc];
    float bch = 0.f;
    if (CUDNN) bch = bch_g[cc];

    if (tid < 4 * GRU_QPAD) { hs[tid] = 0.f; rhs[tid] = 0.f; }
    __syncthreads();

    const float* xb = xproj + (size_t)b * T * xld + (size_t)d * 3 * H;
    float* ob = out + (size_t)b * T * 2 * H + (size_t)d * H;

    int t = d ? T - 1 : 0;
    const int dt = d ? -1 : 1;
    // lane kq == 0 of every gate column / lane ke == 0 of every candidate column owns the input term
    float xg = (kq == 0) ? xb[(size_t)t * xld + gc] : 0.f;
    float xc = (ke == 0) ? xb[(size_t)t * xld + 2 * H + cc] : 0.f;
    float hreg = 0.f;   // h[cc] (valid in the ke == 0 lanes)

    for (int s = 0; s < T; ++s, t += dt) {
        float xg_n = 0.f, xc_n = 0.f;
        if (s + 1 < T) {
            const size_t o = (size_t)(t + dt) * xld;
            if (kq == 0) xg_n = xb[o + gc];
            if (ke == 0) xc_n = xb[o + 2 * H + cc];
        }

        // ---- phase 1: gates.  4 lanes per column, 32 FMAs each, DPP-reduced.
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const float* hq = hs + kq * GRU_QPAD;
#pragma unroll
        for (int i = 0; i < 32; i += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(hq + i);
            a0 = fmaf(hv.x, wg[i + 0], a0);
            a1 = fmaf(hv.y, wg[i + 1], a1);
            a2 = fmaf(hv.z, wg[i + 2], a2);
            a3 = fmaf(hv.w, wg[i + 3], a3);
        }
        const float g = gru_sum4((a0 + a1) + (a2 + a3));
        float pc = 0.f;
        if (CUDNN) {
            // candidate's recurrent part does not depend on r here: compute it in the same phase
            const float* he = hs + (ke >> 1) * GRU_QPAD + (ke & 1) * 16;
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                const float4 hv = *reinterpret_cast<const float4*>(he + i);
                p0 = fmaf(hv.x, wc[i + 0], p0);
                p1 = fmaf(hv.y, wc[i + 1], p1);
                p2 = fmaf(hv.z, wc[i + 2], p2);
                p3 = fmaf(hv.w, wc[i + 3], p3);
            }
            pc = gru_sum8((p0 + p1) + (p2 + p3));
        }
        if (kq == 0) {
            const float gate = sigmoidf_(xg + g);
            if (gc < H) {
                rs[gc] = gate;
            } else {
                us[gc - H] = gate;
            }
        }
        __syncthreads();

        if (!CUDNN) {
            // r*h in the quarter-padded layout (one thread per unit), then phase 2 on it
            if (tid < H) rhs[(tid >> 5) * GRU_QPAD + (tid & 31)] = rs[tid] * hs[(tid >> 5) * GRU_QPAD + (tid & 31)];
            __syncthreads();
            const float* he = rhs + (ke >> 1) * GRU_QPAD + (ke & 1) * 16;
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                const float4 hv = *reinterpret_cast<const float4*>(he + i);
                p0 = fmaf(hv.x, wc[i + 0], p0);
                p1 = fmaf(hv.y, wc[i + 1], p1);
                p2 = fmaf(hv.z, wc[i + 2], p2);
                p3 = fmaf(hv.w, wc[i + 3], p3);
            }
            pc = gru_sum8((p0 + p1) + (p2 + p3));
        }
        if (ke == 0) {
            const float c = CUDNN ? tanhf_(xc + rs[cc] * (pc + bch)) : tanhf_(xc + pc);
            const float u = us[cc];
            const float hn = u * hreg + (1.0f - u) * c;
            hreg = hn;
            ob[(size_t)t * 2 * H + cc] = hn;
        }
        xg = xg_n;
        xc = xc_n;
        // every read of hs of this step happened before the barrier(s) above: publish the new state
        if (ke == 0) hs[(cc >> 5) * GRU_QPAD + (cc & 31)] = hreg;
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
