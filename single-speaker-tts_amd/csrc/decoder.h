// Decoder-loop internals shared between decoder.hip and the api_*.hip files.
#pragma once
#include "tts_common.h"

namespace tts {

enum DecEpi { DEC_EPI_ACT = 0, DEC_EPI_GRU_GATES = 1, DEC_EPI_GRU_CAND = 2, DEC_EPI_GRU_CUDNN_PRE = 3 };

#define TTS_ATT_PARTS 4   // attention positions are split over this many workgroups per utterance

// out[B][N] = epi( [a0 | a1][B][K] . Wt[N][K]^T + bias ); the A operand is the concatenation of
// two row-major segments: columns [0,k0) from a0 (row stride lda0), [k0,K) from a1 (lda1).
// With `parts` set, segment 1 is the attention context assembled on the fly from the
// TTS_ATT_PARTS partial contexts and their softmax statistics (see dec_attention_kernel).
struct DecGemm {
    const float* a0;
    const float* a1;
    const float* Wt;
    const float* bias;
    float* out;
    float* h;            // GRU state [B][U]
    float* rh;           // r*h (GRUCell) or r (cudnn)
    float* u;
    float* hh;
    float* xi;
    const float* resid;
    const float* parts;  // [TTS_ATT_PARTS][B][lda1] partial (unnormalised) contexts or null
    const float* stats;  // [B][TTS_ATT_PARTS][2] = (max, sum) per part
    int lda0, lda1, k0;
    int B, N, K;
    int ldo, ldr, U, act, epi;
};

struct DecoderWeights {
    struct Gru {
        const float* gates_wt;  // [2U][in+U]  (cudnn: [4U][in+U] = r|u|hh|xi zero-padded)
        const float* gates_b;
        const float* cand_wt;   // [U][in+U]   (GRUCell only)
        const float* cand_b;
    };
    const float* prenet1_wt; const float* prenet1_b;     // [P1][n_mels + A]   (step 0: GO frame)
    const float* prenet1f_wt; const float* prenet1f_b;   // [P1][U + A] with the output projection folded in
    const float* prenet2_wt; const float* prenet2_b;     // [P2][P1]
    Gru att_gru;
    const float* attn_layer_wt;                          // [A][A + mem]
    Gru gru[4];
    const float* out_wt; const float* out_b;             // [r*n_mels][U]
    int n_layers, att_units, dec_units, mem_units, n_mels, reduction, prenet1_units, prenet2_units;
    int local_d;         // > 0: LocalLuongAttention (dot score) with window 2*local_d + 1
    int local_gaussian;  // luong_force_gaussian (affects the reported alignments only)
    int local_predictive;            // window centre predicted per utterance instead of the step index
    const float* local_wp;           // [A][A]  (row-major as TF stores it: q @ W_p)
    const float* local_vp;           // [A]
    const float* ws_wimg;            // decoder_ws.hip: the weights in register order (decoder_ws_pack), or null
    const float* ws_bimg;            // ... and the biases per workgroup
};

struct DecoderScratch {
    float* state;        // att | h_att | h_dec[0..n_layers) contiguous, zeroed per call
    size_t state_bytes;
    float* att; float* h_att; float* h_dec[4];
    float* h_att_alt; float* h_dec_alt[4];   // second copies of the states (persistent decoder: double-buffered by step parity)
    float* p1; float* p2; float* rh; float* u; float* hh; float* xi; float* y0; float* y1;
    float* ctx_parts;    // [TTS_ATT_PARTS][B][mem]
    float* att_stats;    // [n_steps][B][TTS_ATT_PARTS][2]
    float* yhist;        // [B][n_steps][U]: top-layer outputs of every step (feeds the deferred output projection)
    float* align_raw;    // [n_steps][B][Ts] unnormalised exp(score - part max), used when the caller wants no alignments
    const float* zeros;  // >= n_mels zero floats
    float* p_hist;       // [n_steps][B] predicted window centres (predictive local attention) or null
    int* err_flag;       // set to 1 when a predicted window leaves the memory (predictive local attention)
};

// Enqueues the whole n_steps loop on stream s (capturable: no syncs, no allocations), including
// the deferred alignment normalisation.  The output projection of all steps (yhist -> mel) is
// one large GEMM issued by the caller.
hipError_t decoder_enqueue(hipStream_t s, const DecoderWeights& w, const DecoderScratch& sc,
                           const float* memory, const float* keys, int B, int Ts, int n_steps,
                           float* align, int cudnn);

// ---- persistent form (decoder_persistent.hip): one launch for the whole loop; both GRU formulations, global and local attention
struct PdParams {
    const float *w1, *b1, *w1f, *b1f, *w2, *b2;   // pre-net (step 0 / folded / layer 2)
    const float *ag_w, *ag_b, *ac_w, *ac_b;       // attention GRU gates / candidate
    const float* al_w;                            // attention layer
    const float *g_gw[2], *g_gb[2], *g_cw[2], *g_cb[2];
    const float *memory, *keys;                   // [B][Ts][256]
    float *att, *p1, *p2, *rh, *ctx, *y0, *yhist;   // hand-off buffers [B][.] (state zeroed per call)
    float *h_att2[2], *h_dec2[2][2];              // recurrent states, double-buffered by step parity (zeroed per call)
    int dbg_delay;                                // tests only: workgroup 3 of every cluster sleeps this long before it stages
    float* align;                                 // [n_steps][B][Ts] or null
    unsigned* counters;                           // [clusters][64]: one arrival counter per cluster, zeroed per call
    unsigned* resident;                           // workgroups that have started
    int* status;                                  // set to 1 when a wait timed out (results are then invalid)
    int* hold_flag;                               // optional: raised once every workgroup is resident (reserve.hip sleepers)
    int B, Ts, n_steps, n_mels;
    int cudnn;                                    // CudnnCompatibleGRUCell arithmetic (gates_wt holds [r | u | hh | xi])
    int local_d, local_gaussian, local_predictive;   // LocalLuongAttention (0 = global attention)
    const float *local_wp, *local_vp;
    float* p_hist;                                // [n_steps][B] predicted centres (predictive mode)
    int* err_flag;                                // raised when a predicted window leaves the memory
};
bool decoder_persistent_supports(const DecoderWeights& w, int cudnn, int B, int Ts);
int decoder_persistent_workgroups(int B);         // compute units the launch needs all to itself
hipError_t decoder_persistent_configure();        // per device
// `sync`: 64 * ceil(B / 16) + 2 unsigned words; `dbg_delay`: tests only (tts_set_option "pd_debug_delay", per handle),
// see PdParams::dbg_delay
hipError_t decoder_persistent_enqueue(hipStream_t s, const DecoderWeights& w, const DecoderScratch& sc, const float* memory,
                                      const float* keys, int B, int Ts, int n_steps, float* align, unsigned* sync,
                                      int* hold_flag, int cudnn, int dbg_delay);

// ---- weight-stationary persistent form (decoder_ws.hip, round 5): clusters of 16 workgroups x 32 utterances, every
// workgroup's share of the weights resident in registers; both GRU formulations, global and (round 6) local attention
#define DEC_WS_NREG 176          // weight registers per lane (172 used by the GRUCell form, 176 by the CudnnCompatibleGRUCell form)
#define DEC_WS_BIAS_SLOTS 10     // b1 folded | b1 (step 0) | b2 | attention GRU gates, candidate | (attention layer: none) | 2 x (gates, candidate)
struct WsParams {
    const float* wimg; const float* bimg;
    const float *memory, *keys;                   // [B][Ts][256]
    float* state;                                 // [clusters][8 x 32 x 256]: att | h_att x 2 | h_dec1 x 2 | h_dec2 x 2 (by step parity) | y, zeroed per call
    float* rest;                                  // [clusters][p1 | r*h | ctx | y0 | p2]
    float* yhist;                                 // [B][n_steps][256]
    float* align;                                 // [n_steps][B][Ts] or null
    unsigned* counters; unsigned* resident; int* status; int* hold_flag;   // as PdParams
    int B, Ts, n_steps, dbg_delay;
    int cudnn;                                    // CudnnCompatibleGRUCell arithmetic (the register image is packed for it)
    int local_d, local_gaussian, local_predictive;   // LocalLuongAttention (0 = global attention), as PdParams
    const float *local_wp, *local_vp;
    float* p_hist;                                // [n_steps][B] predicted window centres (predictive mode)
    int* err_flag;                                // raised when a predicted window leaves the memory
};
struct DecWsHostWeights {                         // host pointers to the packed [N][K] matrices and biases of DecoderWeights
    const float *w1f, *b1f, *b1, *w2, *b2, *ag_w, *ag_b, *ac_w, *ac_b, *al_w;
    const float *g_gw[2], *g_gb[2], *g_cw[2], *g_cb[2];
    int cudnn;                                    // CudnnCompatibleGRUCell: *g_w / *g_b are the [4U][K] / [4U] blocks r | u | hh | xi, *c_* unused
};
size_t decoder_ws_wimg_floats();
size_t decoder_ws_bimg_floats();
void decoder_ws_pack(const DecWsHostWeights& hw, float* wimg, float* bimg);
bool decoder_ws_supports(const DecoderWeights& w, int cudnn, int B, int Ts);
// rows = utterances per cluster of 16 workgroups: 32 (round 5) or 16 (round 6: twice the compute units, ~0.8 of the time; the
// same bits per utterance)
int decoder_ws_workgroups(int B, int rows = 32);  // compute units the launch needs all to itself
int decoder_ws_clusters(int B, int rows = 32);
size_t decoder_ws_scratch_floats(int B, int rows = 32);
hipError_t decoder_ws_configure();                // per device
// `sync`: 64 * max(clusters, sync_clusters) + 2 unsigned words (counters, resident count, sticky status word)
hipError_t decoder_ws_enqueue(hipStream_t s, const DecoderWeights& w, float* scratch, float* yhist, const float* memory,
                              const float* keys, int B, int Ts, int n_steps, float* align, unsigned* sync, int* hold_flag,
                              int cudnn, int dbg_delay, int rows = 32, int sync_clusters = 0, float* p_hist = nullptr, int* err_flag = nullptr);

}  // namespace tts
