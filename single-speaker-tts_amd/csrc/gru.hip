// Recurrent half of the CBHG bidirectional GRU (gfx950).
//
// Replaces tf.nn.bidirectional_dynamic_rnn(GRUCell fw, GRUCell bw) / CudnnGRU at reference
// tacotron/layers.py:560-592: both directions run over the FULL padded length from a zero
// state (no sequence_length), outputs concatenated [fw | bw].
//
// The input halves (x W_x + b for r, u, c and both directions) are one big MFMA GEMM done
// beforehand; this kernel is the strictly sequential part.  It is latency bound, so the
// design keeps everything on chip: one 256-thread workgroup per (utterance, direction), the
// recurrent weights live in VGPRs for the whole sequence (128 + 64 floats per thread), the
// state lives in LDS and is broadcast-read as float4, and the next step's input projections
// are prefetched while the current step computes.
//
//   GRUCell [TF-1.8]       : [r|u] = sig(xg + h Wgh);  c = tanh(xc + (r*h) Wch);  h' = u h + (1-u) c
//   CudnnCompatibleGRUCell : c = tanh(xc + r * (h Wch + bch))
#include "tts_common.h"

namespace tts {

template <int H, bool CUDNN>
__global__ __launch_bounds__(2 * H) void bigru_kernel(const float* __restrict__ xproj, int xld,
                                                      const float* __restrict__ wrec,
                                                      float* __restrict__ out, int B, int T) {
    constexpr int NT = 2 * H;        // threads
    constexpr int HALF = H / 2;      // k-range of one candidate partial sum
    const int b = blockIdx.x;
    const int d = blockIdx.y;        // 0 = forward, 1 = backward
    const int n = threadIdx.x;
    const int half = n / H;
    const int col = n % H;

    __shared__ __attribute__((aligned(16))) float hs[H];
    __shared__ __attribute__((aligned(16))) float rhs[H];
    __shared__ __attribute__((aligned(16))) float us[H];
    __shared__ __attribute__((aligned(16))) float part[2][H];

    const size_t wstride = (size_t)H * 2 * H + (size_t)H * H + (CUDNN ? H : 0);
    const float* wg_g = wrec + d * wstride;          // [H][2H]
    const float* wc_g = wg_g + (size_t)H * 2 * H;    // [H][H]
    const float* bch_g = wc_g + (size_t)H * H;       // [H] (cudnn)

    float wg[H];
    float wc[HALF];
#pragma unroll
    for (int k = 0; k < H; ++k) wg[k] = wg_g[(size_t)k * NT + n];
#pragma unroll
    for (int k = 0; k < HALF; ++k) wc[k] = wc_g[(size_t)(half * HALF + k) * H + col];
    float bch = 0.f;
    if (CUDNN && n < H) bch = bch_g[n];

    if (n < H) hs[n] = 0.f;
    float hreg = 0.f;
    __syncthreads();

    const float* xb = xproj + (size_t)b * T * xld + (size_t)d * 3 * H;
    float* ob = out + (size_t)b * T * 2 * H + (size_t)d * H;

    int t = d ? T - 1 : 0;
    const int dt = d ? -1 : 1;
    float xg = xb[(size_t)t * xld + n];
    float xc = (n < H) ? xb[(size_t)t * xld + 2 * H + n] : 0.f;

    for (int s = 0; s < T; ++s, t += dt) {
        // prefetch the next step's input projections
        float xg_n = 0.f, xc_n = 0.f;
        if (s + 1 < T) {
            const size_t o = (size_t)(t + dt) * xld;
            xg_n = xb[o + n];
            if (n < H) xc_n = xb[o + 2 * H + n];
        }

        // phase 1: gates
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(&hs[k]);
            a0 = fmaf(hv.x, wg[k + 0], a0);
            a1 = fmaf(hv.y, wg[k + 1], a1);
            a2 = fmaf(hv.z, wg[k + 2], a2);
            a3 = fmaf(hv.w, wg[k + 3], a3);
        }
        const float gate = sigmoidf_(xg + ((a0 + a1) + (a2 + a3)));
        float r = 0.f;
        if (n < H) {
            r = gate;
            if (!CUDNN) rhs[n] = gate * hreg;
        } else {
            us[n - H] = gate;
        }
        if (CUDNN) {
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
            for (int k = 0; k < HALF; k += 4) {
                const float4 hv = *reinterpret_cast<const float4*>(&hs[half * HALF + k]);
                p0 = fmaf(hv.x, wc[k + 0], p0);
                p1 = fmaf(hv.y, wc[k + 1], p1);
                p2 = fmaf(hv.z, wc[k + 2], p2);
                p3 = fmaf(hv.w, wc[k + 3], p3);
            }
            part[half][col] = (p0 + p1) + (p2 + p3);
        }
        __syncthreads();

        if (!CUDNN) {
            // phase 2: candidate on r*h, K split in two halves across the 2H threads
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
            for (int k = 0; k < HALF; k += 4) {
                const float4 hv = *reinterpret_cast<const float4*>(&rhs[half * HALF + k]);
                p0 = fmaf(hv.x, wc[k + 0], p0);
                p1 = fmaf(hv.y, wc[k + 1], p1);
                p2 = fmaf(hv.z, wc[k + 2], p2);
                p3 = fmaf(hv.w, wc[k + 3], p3);
            }
            part[half][col] = (p0 + p1) + (p2 + p3);
            __syncthreads();
        }

        if (n < H) {
            const float hc = part[0][n] + part[1][n];
            const float c = CUDNN ? tanhf_(xc + r * (hc + bch)) : tanhf_(xc + hc);
            const float u = us[n];
            const float hn = u * hreg + (1.0f - u) * c;
            hreg = hn;
            hs[n] = hn;
            ob[(size_t)t * 2 * H + n] = hn;
        }
        xg = xg_n;
        xc = xc_n;
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
        hipLaunchKernelGGL((bigru_kernel<128, true>), grid, dim3(256), 0, s, xproj, xld, wrec, out, B, T);
    else
        hipLaunchKernelGGL((bigru_kernel<128, false>), grid, dim3(256), 0, s, xproj, xld, wrec, out, B, T);
    return hipGetLastError();
}

}  // namespace tts
