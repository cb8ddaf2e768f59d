// Internal declarations shared by the HIP translation units of libsstts_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <map>
#include <string>
#include <vector>

#include "../../include/sstts_hip.h"

namespace tts {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_TANH = 3 };
enum Epi { EPI_STD = 0, EPI_HIGHWAY = 1 };

// Reciprocals are v_rcp_f32 (1 ulp): `1.0f / x` compiles to the IEEE division sequence (v_div_scale, v_rcp, four
// fmas, v_div_fmas, v_div_fixup: ten dependent instructions), which sits three times on the critical path of every
// GRU step.  rcp(inf) = 0 and rcp(1) = 1, so the limits stay exact.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1); exact limits at +-inf, abs error ~1e-7.
    float e = __expf(2.0f * x);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
// reference tacotron/inference.py:96-101,175 + audio/conversion.py:102,51: clip -> dB -> magnitude -> ** power
__device__ __forceinline__ float denorm_db(float x, float ref_db, float range_db) {
    const float c = fminf(fmaxf(x, 0.f), 1.f);
    return (c - 1.0f) * range_db + ref_db;
}
__device__ __forceinline__ float db_pow(float db, float power) {
    // (10^(db/20)) ** power == 2^(db * power * log2(10) / 20): one exp2 instead of exp2 + powf
    return exp2f(db * (power * (0.05f * 3.3219280948873623f)));
}
__device__ __forceinline__ float denorm_pow(float x, float ref_db, float range_db, float power) {
    return db_pow(denorm_db(x, ref_db, range_db), power);
}
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    if (act == ACT_SIGMOID) return sigmoidf_(v);
    if (act == ACT_TANH) return tanhf_(v);
    return v;
}

// ----------------------------------------------------------------------------- GEMM (gemm_f32.hip)
// C[m, coff+n] = epi( sum_k A_op[m,k] * Wt[n,k] ), fp32 MFMA.  A_op is an implicit im2col of a
// channels-last (rows = B*T) activation: element (m, k) = x[m - padl + k / Cin][k % Cin] when the
// shifted time index stays inside the row's sequence, else 0 (TF 'SAME' conv1d, stride 1).
// Dense layers are the Cin == K, padl == 0 case.
struct GemmGroup {
    const float* A;
    const int32_t* gather;  // optional: row m reads A + gather[m] * lda (embedding lookup)
    int gather_rows;        // rows of the gathered table: ids outside [0, gather_rows) read as a zero row
    const float* Wt;        // packed weights [N][K], k contiguous
    const unsigned char* Wimg;   // the same weights pre-split into the kernel's bf16 LDS images (gemm_f32.hip, PRE), or null
    const float* bias;      // [N] or null
    const float* scale;     // [N] folded batch-norm scale or null
    const float* shift;     // [N] folded batch-norm shift (used with scale)
    const float* R;         // residual [M][ldr] or null
    float* C;
    int M, N, K;
    int lda, T, Cin, padl, pool;
    int ldc, coff, ldr;
    int act, epi;
    // optional second output: C2[m][n] = (10^(((clip(v,0,1)-1)*d_range + d_ref)/20)) ** d_pow for n < N,
    // 0 for N <= n < N2 (the de-normalised, power-raised magnitude of the final Dense layer)
    float* C2;
    int ldc2, N2;
    float d_ref, d_range, d_pow;
    int* d_flag;            // optional: set to 1 when a de-normalised value lies below -100 dB (reference
                            // audio/conversion.py:47-49); only passed when the constants allow that at all
    // split-K: this group accumulates only the k tiles [kt0, kt1) (multiples of the tile depth 32; 0, 0 = all of
    // K) and is launched with a plain epilogue into a partial-sum buffer; see launch_gemm_splitk
    int kt0, kt1;
};
#define TTS_GEMM_MAX_GROUPS 16
struct GemmBatch {
    GemmGroup g[TTS_GEMM_MAX_GROUPS];
    int ps;   // 1: the producer / consumer form of the kernel (512 threads: four multiplying, four staging waves; gemm_f32.hip, PS)
};
// Launches one grouped GEMM; all groups must share M (grid.x) and have N <= max_n.
hipError_t launch_gemm(hipStream_t s, const GemmBatch& b, int n_groups);
bool gemm_experiments_built();   // gemm_f32.hip compiled with -DGEMM_EXPERIMENTS (tools): the PRE / PS variants exist
// One GEMM whose K range is cut into `slices` parts computed by separate workgroups (for problems with too
// few output tiles to fill the GPU); `partial` holds slices*M*N floats.  The partial sums are added in slice
// order and the group's epilogue (bias, activation, affine, residual) is applied by a second small kernel.
hipError_t launch_gemm_splitk(hipStream_t s, const GemmGroup& g, int slices, float* partial, int ps = 0);
// the pre-split image of a weight matrix used with this (K, Cin): gemm_weight_image_bytes(N, K) bytes
size_t gemm_weight_image_bytes(int N, int K);
hipError_t launch_gemm_pack_weights(hipStream_t s, const float* Wt, unsigned char* img, int N, int K, int Cin);
int gemm_splitk_slices(int K);   // 1 = not worth splitting; a function of the layer only, never of the batch

// ----------------------------------------------------------------------------- bi-GRU (gru.hip)
// xproj [B*T][xld]: per direction d a block of 3*H input projections (bias included) at column
// d*3H: [r | u | c].  wpack: per direction, recurrent weights packed by pack_gru_recurrent().
// out [B*T][2H] = [fw | bw].
hipError_t launch_bigru(hipStream_t s, const float* xproj, int xld, const float* wrec, float* out,
                        int B, int T, int H, int cudnn);
size_t bigru_wrec_floats(int H, int cudnn);

// ----------------------------------------------------------------------------- CBHG tail (cbhg_tail.hip)
// lifter + highway stack + GRU input projections of a CBHG in one launch; the rows stay in LDS between the layers.
#define CBHG_TAIL_MAX_HW 8
struct CbhgTailParams {
    const float* X; int ldx; int c_in;        // [M][ldx], c_in valid floats per row (the projections' residual output)
    const float* lifter_wt; const float* lifter_b;                       // [128][c_in], [128]
    const float* hw_wt[CBHG_TAIL_MAX_HW]; const float* hw_b[CBHG_TAIL_MAX_HW]; int n_hw;   // [256][128] / [256], H|T packed per 64-row span
    const float* gru_wt; const float* gru_b;  // [768][128], [768]
    float* hw_out;                            // [M][128] the highway stack's output (null: not wanted)
    float* xproj;                             // [M][768]
    int M;
};
bool cbhg_tail_supports(int c_in, int units, int gru_units, int n_hw, long long M);
hipError_t cbhg_tail_configure();             // per device, before the first launch
hipError_t launch_cbhg_tail(hipStream_t s, const CbhgTailParams& p);

// ----------------------------------------------------------------------------- CU reservation (reserve.hip)
hipError_t cu_hold_configure();   // per device, before the first launch_cu_hold
hipError_t launch_cu_hold(hipStream_t s, int n_cus, const int* flag, double timeout_ms, int lds_kb = 64);

// ----------------------------------------------------------------------------- helpers
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace tts
