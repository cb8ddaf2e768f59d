// Recurrent half of the CBHG bidirectional GRU (gfx950).
//
// Replaces tf.nn.bidirectional_dynamic_rnn(GRUCell fw, GRUCell bw) / CudnnGRU at reference
// tacotron/layers.py:560-592: both directions run over the FULL padded length from a zero
// state (no sequence_length), outputs concatenated [fw | bw].
//
// The input halves (x W_x + b for r, u, c and both directions) are one big MFMA GEMM done
// beforehand; this kernel is the strictly sequential part.  It is latency bound (T dependent
// steps of a 128 -> 384 mat-vec), so the design minimises the dependent chain of one step:
//   * one 512-thread workgroup per (utterance, direction): thread (unit j, K quarter kq) owns the r, u AND candidate
//     columns of unit j over 32 of the 128 k -- 96 recurrent weights in VGPRs for the whole sequence, so r, u, the
//     candidate and the state of a unit all end up in the same four lanes and only r*h has to cross the workgroup;
//   * the three dot products run on v_pk_fma_f32 (two consecutive k per instruction) and are reduced over the quad
//     with DPP operands (no LDS round trip);
//   * two workgroup barriers per step (GRUCell form: after r*h is published, after the new state is; the
//     CudnnCompatibleGRUCell form needs only the second) of 8 waves instead of three of 16 (the 1024-thread kernel
//     it replaces: 2540 cycles per step, 900 of them at barriers);
//   * the state lives in LDS, ping-pong, in a padded layout whose four K-quarters fall into different banks;
//   * the next step's input projections are prefetched while the current step computes.
//
//   GRUCell [TF-1.8]       : [r|u] = sig(xg + h Wgh);  c = tanh(xc + (r*h) Wch);  h' = u h + (1-u) c
//   CudnnCompatibleGRUCell : c = tanh(xc + r * (h Wch + bch))
#include "tts_common.h"

namespace tts {

#define GRU_THREADS 512
#define GRU_QPAD 36   // floats per K-quarter of the state in LDS (32 + 4: quarters hit different banks)

// Quad sum with DPP operands (v_add_f32_dpp, no LDS round trip: hipcc lowers __shfl_xor to ds_bpermute_b32, a
// dependent LDS access per reduction level).  Every lane of the quad ends up with the sum.
template <int CTRL>
__device__ __forceinline__ float gru_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float gru_sum4(float v) {
    v += gru_dpp<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
    v += gru_dpp<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
    return v;
}
// Workgroup barrier that orders LDS accesses only: nothing one thread writes to global memory is read by another
// thread of this kernel, so a step must not wait for its output store (or the prefetched inputs) at the barrier.
__device__ __forceinline__ void gru_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef float gru_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gru_f2 gru_pk_fma(gru_f2 a, gru_f2 b, gru_f2 c) {   // (a.x b.x + c.x, a.y b.y + c.y)
    gru_f2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// sum_k q[k] * w[k] over this lane's 32 k (w as 16 pairs), two packed accumulators
__device__ __forceinline__ float gru_dot32(const float* q, const gru_f2 (&w)[16]) {
    gru_f2 a = {0.f, 0.f}, b = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32; i += 4) {
        const float4 hv = *reinterpret_cast<const float4*>(q + i);
        a = gru_pk_fma((gru_f2){hv.x, hv.y}, w[i / 2], a);
        b = gru_pk_fma((gru_f2){hv.z, hv.w}, w[i / 2 + 1], b);
    }
    return (a.x + a.y) + (b.x + b.y);
}

template <int H, bool CUDNN>
__global__ __launch_bounds__(GRU_THREADS) void bigru_kernel(const float* __restrict__ xproj, int xld,
                                                            const float* __restrict__ wrec,
                                                            float* __restrict__ out, int B, int T) {
    static_assert(H == 128, "thread mapping assumes 128 units");
    const int b = blockIdx.x;
    const int d = blockIdx.y;        // 0 = forward, 1 = backward
    const int tid = threadIdx.x;
    const int j = tid >> 2, kq = tid & 3;   // unit, K quarter

    // state, quarter-padded, two copies: step s reads copy s & 1 and writes the other one, so the barrier at the end of
    // a step is all that separates a step's reads from the next write of the same words
    __shared__ __attribute__((aligned(16))) float hs[2][4 * GRU_QPAD];
    __shared__ __attribute__((aligned(16))) float rhs[4 * GRU_QPAD];   // r*h (GRUCell form)

    const size_t wstride = (size_t)H * 2 * H + (size_t)H * H + (CUDNN ? H : 0);
    const float* wg_g = wrec + d * wstride;          // [H][2H]: columns [r | u]
    const float* wc_g = wg_g + (size_t)H * 2 * H;    // [H][H]
    const float* bch_g = wc_g + (size_t)H * H;       // [H] (cudnn)

    gru_f2 wr[16], wu[16], wc[16];   // (W[32 kq + 2 i][col], W[32 kq + 2 i + 1][col]) for col = r_j, u_j, c_j
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const size_t k0 = 32 * kq + 2 * i;
        wr[i] = (gru_f2){wg_g[k0 * (2 * H) + j], wg_g[(k0 + 1) * (2 * H) + j]};
        wu[i] = (gru_f2){wg_g[k0 * (2 * H) + H + j], wg_g[(k0 + 1) * (2 * H) + H + j]};
        wc[i] = (gru_f2){wc_g[k0 * H + j], wc_g[(k0 + 1) * H + j]};
    }
    float bch = 0.f;
    if (CUDNN) bch = bch_g[j];

    if (tid < 4 * GRU_QPAD) { hs[0][tid] = 0.f; hs[1][tid] = 0.f; rhs[tid] = 0.f; }
    __syncthreads();

    const float* xb = xproj + (size_t)b * T * xld + (size_t)d * 3 * H + j;
    float* ob = out + (size_t)b * T * 2 * H + (size_t)d * H + j;
    const int slot = (j >> 5) * GRU_QPAD + (j & 31);   // unit j in the quarter-padded layout

    int t = d ? T - 1 : 0;
    const int dt = d ? -1 : 1;
    // Input terms of step s (the four lanes of a unit load the same three words: one request each after coalescing)
    // live in three register sets that rotate by NAME (the loop is unrolled by three): the set a step has consumed is
    // refilled for step s + 3 by unconditional loads (the step index is clamped, not tested).  A load under a branch,
    // or a register-to-register rotation of the sets, makes the compiler wait with vmcnt(0) inside the step -- for the
    // loads it has just issued, i.e. the whole global-load latency on the critical path of every step.
    auto xrow = [&](int s) {
        const int sc = s < T ? s : T - 1;
        return xb + (size_t)(d ? T - 1 - sc : sc) * xld;
    };
    float xa[3] = {xrow(0)[0], xrow(0)[H], xrow(0)[2 * H]};
    float xb1[3] = {xrow(1)[0], xrow(1)[H], xrow(1)[2 * H]};
    float xc2[3] = {xrow(2)[0], xrow(2)[H], xrow(2)[2 * H]};
    float hreg = 0.f;   // h[j], kept by all four lanes of the unit

    int s = 0;
#define GRU_STEP(X)                                                                            \
    {                                                                                          \
        const float* hq = hs[s & 1] + kq * GRU_QPAD;                                           \
        const float r = sigmoidf_(X[0] + gru_sum4(gru_dot32(hq, wr)));                         \
        const float u = sigmoidf_(X[1] + gru_sum4(gru_dot32(hq, wu)));                         \
        float c;                                                                               \
        if (CUDNN) { /* the candidate's recurrent part does not depend on r: same pass over h */ \
            c = tanhf_(X[2] + r * (gru_sum4(gru_dot32(hq, wc)) + bch));                        \
        } else {                                                                               \
            if (kq == 0) rhs[slot] = r * hreg;                                                 \
            gru_lds_barrier();                                                                 \
            c = tanhf_(X[2] + gru_sum4(gru_dot32(rhs + kq * GRU_QPAD, wc)));                   \
        }                                                                                      \
        hreg = u * hreg + (1.0f - u) * c;                                                      \
        if (kq == 0) {                                                                         \
            ob[(size_t)t * 2 * H] = hreg;                                                      \
            hs[(s + 1) & 1][slot] = hreg;                                                      \
        }                                                                                      \
        const float* xn = xrow(s + 3);                                                         \
        X[0] = xn[0]; X[1] = xn[H]; X[2] = xn[2 * H];                                          \
        ++s; t += dt;                                                                          \
        gru_lds_barrier();                                                                     \
    }
    while (s + 3 <= T) {
        GRU_STEP(xa)
        GRU_STEP(xb1)
        GRU_STEP(xc2)
    }
    if (s < T) GRU_STEP(xa)
    if (s < T) GRU_STEP(xb1)
#undef GRU_STEP
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
