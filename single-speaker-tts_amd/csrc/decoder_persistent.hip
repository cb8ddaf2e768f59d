// Persistent form of the Luong-attention GRU decoder loop (gfx950): ONE launch for all n_steps steps.
//
// Same arithmetic as decoder.hip (reference tacotron/model.py:191-331, wrappers.py:94-124, helpers.py:83-110,
// 161-205; GRUCell form, global LuongAttention) -- what changes is who waits for whom.  decoder.hip issues ten
// dependent launches per step; under the call pipeline every one of them queues behind whatever else the chip is
// running.  Here a CLUSTER of PD_W workgroups (one per CU, 1024 threads) owns 16 utterances for the whole loop:
//   * every layer of a step is a phase; workgroup j of the cluster computes the same slice of output units in
//     every step (32 of 256 units, 16 of the 128 of pre-net 2; for a GRU its r AND u columns, so the update gate
//     and the cell state of its units never leave its LDS), K split over the waves, v_mfma_f32_16x16x4_f32;
//   * the weights of the NEXT phase are fetched from L2 into registers BEFORE the workgroup waits for its peers
//     (they do not depend on data);
//   * activations travel between the workgroups of a cluster through small global buffers with the write-through
//     hand-off of MI355X_MICROARCH.md ("valid forms"): every byte stored sc1; every storing wave waits vmcnt(0) and
//     adds to the cluster's counter (agent scope) for itself -- a GEMM phase's stores come from its first one or two
//     waves, and no workgroup barrier holds the other fourteen back from the next phase's weight requests (round 4:
//     14.9 -> 13.6 ms alone; the attention phase, where every wave stores, keeps drain -> barrier -> one add);
//     consumers poll that counter with an sc1 load in one lane, then a workgroup barrier, then sc1 loads of the
//     bytes.  No cache-wide release or acquire anywhere, no grid-wide barrier: clusters never talk to each other;
//   (Round 4 measured the alternative hand-off of cdna_hip_programming.md Guideline 16 R2 -- every value an 8-byte
//   {value, tag} granule, the consumers sweeping the tile until the tags match, no counter and no drained stores --
//   on one box against this form: 15.48 against 14.86 ms alone, 16.1 against 15.5 ms beside Griffin-Lim.  A phase is bound
//   by the bytes its compute unit moves (128 KB of weights, the staged tile, the attention rows), not by the round trips
//   of the protocol, and the granules double the tile's bytes.  Requesting the next phase's weights a phase ahead did not
//   pay either: at 128 registers it spills, and scratch accesses queue behind the weight loads; with 512-thread
//   workgroups and 256 registers the dependent MFMA chain of a wave doubles: 18.3 ms.  git history has both.)
//   * attention: workgroup j scores, normalises and contracts two of the cluster's 16 rows itself, so softmax
//     needs no cross-workgroup merge and the alignments are written normalised.
// Every wait is bounded (PD_SPIN_LIMIT polls); on a timeout the status word is set, every workgroup of the grid
// sees it at its next wait and the kernel drains.  All workgroups must be co-resident: the host only uses this
// path when 8 * ceil(B / 16) compute units are free for it (api_stages.hip).
#include "tts_common.h"
#include "decoder.h"
#include <cstdio>

namespace tts {

#define PD_W 8
#define PD_NW 16
#define PD_THREADS (PD_NW * 64)
#define PD_D 256                  // attention units = decoder units = memory depth = pre-net 1 units
#define PD_P2 128                 // pre-net 2 units
#define PD_LDA 516                // LDS row stride (floats) of the staged A tile: K <= 512; +4 keeps b128 reads conflict-free
#define PD_RED_LD 20              // row stride of the per-wave partial tiles (b128 aligned)
#define PD_SPIN_LIMIT 2000000u      // polls of ~1 us each
// attention: key passes (32 positions each) / value rows per wave in flight together.  Same-box decoder times alone /
// beside Griffin-Lim: one pass and two rows at a time (round-2 start) 15.16 / 17.29 ms, 2 and 4: 14.87 / 17.24, 2 and 8:
// 15.05 / 17.27, 3 and 8: 15.78 / 17.60 (the prefetched keys live across the wait for the cluster and spill).  The phase
// is bound by the cluster's arrival skew, not by its dependent memory trips.
#ifndef PD_KB
#define PD_KB 2
#endif
#ifndef PD_VB
#define PD_VB 4
#endif

typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned pd_u32x4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pd_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)0xFFFFFFF0u, 0x00020000);
}
// 16-byte sc1 (write-through / L1-bypassing) accesses: aux bit 4
__device__ __forceinline__ float4 pd_ld4(const __amdgpu_buffer_rsrc_t& rs, unsigned byte_off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16));
}
__device__ __forceinline__ void pd_st4(const __amdgpu_buffer_rsrc_t& rs, unsigned byte_off, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pd_u32x4, v), rs, (int)byte_off, 0, 16);
}

// PD_GATES + PD_CAND: TF GRUCell (candidate on [x ; r*h], two hops per cell).  PD_CUDNN_RU + PD_CUDNN_HX:
// CudnnCompatibleGRUCell (reference layers.py:560-577, model.py:226-227,257-259): c = tanh(x Wci + bci + r*(h Wch + bch));
// all four column blocks [r | u | hh | xi] come from the same staged [x ; h] tile, r and u never leave the
// workgroup, so the cell is ONE hop: the second pass continues on the tile of the first (`cont`).
enum PdEpi { PD_ACT = 0, PD_GATES = 1, PD_CAND = 2, PD_CUDNN_RU = 3, PD_CUDNN_HX = 4 };

#ifdef PD_TIMELINE   // tools only: s_memrealtime stamps (100 MHz) of workgroup 0 in step 100, [phase][8], kept in LDS
__device__ unsigned long long* pd_dbg = nullptr;
__device__ __shared__ int pd_tl_step, pd_tl_phase;
__device__ __shared__ float* pd_tl_lds;
#define PD_STAMP(I)                                                                                              \
    if (blockIdx.x == 0 && threadIdx.x == 0 && pd_tl_step == 100)                                               \
        reinterpret_cast<unsigned long long*>(pd_tl_lds)[pd_tl_phase * 8 + (I)] = __builtin_amdgcn_s_memrealtime();
#else
#define PD_STAMP(I)
#endif

struct PdPhase {
    const float* a0; int lda0; int k0;   // A columns [0, k0): a0 (null = zeros), row stride lda0
    const float* a1; int lda1;           // A columns [k0, K)
    int K;
    const float* Wt;                     // [N][K]
    const float* bias;                   // [N] or null
    int row0;                            // first of the (one or two) 256-row weight blocks this phase uses
    int cont;                            // continues on the A tile the previous phase staged: no wait, no staging
    int more;                            // a `cont` phase follows: nothing to publish yet
    int ub;                              // units of this layer owned per workgroup (32 or 16)
    int epi, act, layer;
    int delay;                           // tests only: workgroup 3 sleeps delay x ~3.4 us between the wait and the staging
    float* out; int ldo;                 // PD_ACT: activations; PD_GATES: r*h; PD_CAND: new state h
    float* yout; int ldy;                // PD_CAND with residual: y = x + h'
};

// LDS map (floats)
#define PD_OFF_AS 0
#define PD_OFF_RED (16 * PD_LDA)
#define PD_OFF_H (PD_OFF_RED + PD_NW * 16 * PD_RED_LD)
#define PD_OFF_U (PD_OFF_H + 3 * 16 * 32)
#define PD_OFF_R (PD_OFF_U + 16 * 32)
#define PD_OFF_CTRL (PD_OFF_R + 16 * 32)
#define PD_OFF_SC (PD_OFF_CTRL + 16)
#ifdef PD_TIMELINE
size_t pd_lds_bytes(int Ts) { return ((size_t)PD_OFF_SC + 2 * (size_t)((Ts + 3) & ~3)) * sizeof(float) + 1024; }
#else
size_t pd_lds_bytes(int Ts) { return ((size_t)PD_OFF_SC + 2 * (size_t)((Ts + 3) & ~3)) * sizeof(float); }
#endif

// Start of a phase: wait until `target` arrivals have been counted on the cluster's counter (lane 0 polls, everybody
// meets at the barrier).  The arrival of THIS workgroup for the previous phase was signalled when that phase ended
// (pd_publish): until round 4 it was signalled here, behind the next phase's weight requests and address arithmetic --
// 1.0-1.4 us in which the peers could not yet see that this workgroup was done, on every phase's critical path.
__device__ __forceinline__ void pd_wait(unsigned* cnt, unsigned target, int* status, int* ctrl) {
    if (threadIdx.x == 0 && target > 0) {
        if (ctrl[0] == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
#ifndef PD_NO_POLL_SLEEP
                __builtin_amdgcn_s_sleep(1);
#endif
                if ((++spins & 1023u) == 0 &&
                    (spins > PD_SPIN_LIMIT || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ctrl[0] = 1;   // drain: no further waits in this workgroup
                    break;
                }
            }
        }
    }
    __syncthreads();
}

// End of a phase: every storing wave waits for its (write-through) stores, then the workgroup barrier -- after it
// one lane signals for all of them: the arrival on the cluster's counter (agent scope), at once.
// A workgroup counts PD_ARRIVALS on the counter per phase, however many of its waves store.
#define PD_ARRIVALS 2u
__device__ __forceinline__ void pd_publish(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, PD_ARRIVALS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The same for a phase whose stores all come from its first one or two waves (the GEMM phases' epilogue): every storing wave
// signals for itself when ITS stores have drained (MI355X_MICROARCH.md, valid forms: "each storing wave for itself"), and
// no workgroup barrier stands between the other fourteen waves and the next phase's weight requests: those are in
// flight while the epilogue runs, not behind it.  (What the barrier also did -- keep the partial tiles and the staged tile
// from being overwritten under the epilogue's reads -- the next phase's own barrier does: nobody writes either before
// pd_wait, and a `cont` phase is entered through the barrier of the `more` phase in front of it.)
__device__ __forceinline__ void pd_publish_wave(unsigned* cnt, unsigned arrivals) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One GEMM-shaped phase of the cluster: out[16 rows][this workgroup's units] = epi([a0 | a1] . Wt^T + bias).
__device__ __forceinline__ void pd_phase(const PdPhase& ph, float* lds, int j, int b0, int B, unsigned* cnt, unsigned target,
                                         int* status) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    float* As = lds + PD_OFF_AS;
    float* red = lds + PD_OFF_RED;
    float* h_loc = lds + PD_OFF_H + ph.layer * (16 * 32);
    float* u_loc = lds + PD_OFF_U;
    float* r_loc = lds + PD_OFF_R;
    int* ctrl = reinterpret_cast<int*>(lds + PD_OFF_CTRL);

    const int gates = (ph.epi == PD_GATES || ph.epi == PD_CUDNN_RU || ph.epi == PD_CUDNN_HX) ? 2 : 1;
    const int tpg = ph.ub >> 4;                 // 16-column tiles per gate
    const int tiles = gates * tpg;              // 1, 2 or 4
    const int ksl = PD_NW / tiles;              // K slices
    const int tile = wave % tiles, slice = wave / tiles;
    const int gate = tile / tpg, within = tile - gate * tpg;
    const int nch = ph.K >> 4;

    // ---- this wave's weight fragments: independent of every other workgroup, so they are in flight during the wait
    const int n = ph.row0 + gate * PD_D + j * ph.ub + 16 * within + r;
    const float* wrow = ph.Wt + (size_t)n * ph.K + 4 * q;
    float4 bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = slice + ksl * i;
        bv[i] = *reinterpret_cast<const float4*>(wrow + 16 * (c < nch ? c : nch - 1));
    }

    // ---- where this thread's (at most two) float4 of the cluster's A tile come from and go to: computed before the
    // wait as well (an integer division per element; behind the wait it was ~0.6 us of every phase)
    unsigned soff[2];   // byte offset into a0 (ssel 0) or a1 (ssel 1); ssel 2 = zeros
    int sdst[2], ssel[2];
    if (!ph.cont) {
        const int k4 = ph.K >> 2;                 // float4 per row
        const int total = 16 * k4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * PD_THREADS;
            const int ic = i < total ? i : 0;
            const int row = ic / k4, kk = (ic - row * k4) * 4;
            const int b = b0 + row < B ? b0 + row : B - 1;
            sdst[u] = i < total ? row * PD_LDA + kk : -1;
            const bool first = kk < ph.k0;
            ssel[u] = i >= total ? 3 : (first ? (ph.a0 ? 0 : 2) : 1);
            soff[u] = first ? (unsigned)(b * ph.lda0 + kk) * 4u : (unsigned)(b * ph.lda1 + kk - ph.k0) * 4u;
        }
    }

    PD_STAMP(0)
    if (!ph.cont) pd_wait(cnt, target, status, ctrl);
    PD_STAMP(1)
    if (ph.delay && !ph.cont && j == 3)   // a late stager: what a workgroup that clears its poll late looks like to its peers
        for (int i = 0; i < ph.delay; ++i) __builtin_amdgcn_s_sleep(127);

    // ---- stage the cluster's A tile (16 rows x K) in LDS: sc1 loads of the handed-off activations.  K <= 512: at most
    // two float4 per thread, BOTH requested before either is written to LDS (written as a loop of load-then-store the
    // second request waited for the first: two fabric round trips per phase instead of one)
    if (!ph.cont) {
        const __amdgpu_buffer_rsrc_t r0 = pd_rsrc(ph.a0 ? ph.a0 : ph.a1), r1 = pd_rsrc(ph.a1);
        float4 sv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            sv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ssel[u] == 0) sv[u] = pd_ld4(r0, soff[u]);
            else if (ssel[u] == 1) sv[u] = pd_ld4(r1, soff[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (sdst[u] >= 0) *reinterpret_cast<float4*>(As + sdst[u]) = sv[u];
        __syncthreads();
    }
    PD_STAMP(2)

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = slice + ksl * i;
        if (c < nch) {   // wave-uniform
            const float4 av = *reinterpret_cast<const float4*>(As + r * PD_LDA + 16 * c + 4 * q);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[i].w, acc, 0, 0, 0);
        }
    }
    // C/D map of 16x16: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < 4; ++i) red[(wave * 16 + q * 4 + i) * PD_RED_LD + r] = acc[i];
    __syncthreads();
    PD_STAMP(3)

    // ---- epilogue: thread e owns (row, 4 consecutive units) of every gate; K slices added in a fixed order
    if (tid < tpg * 64) {
        const int ew = tid >> 6, row = (tid >> 2) & 15, c4 = (tid & 3) * 4;
        const int cl = 16 * ew + c4;             // column inside this workgroup's unit block
        const int unit = j * ph.ub + cl;
        float4 v[2];
#pragma unroll
        for (int eg = 0; eg < 2; ++eg) {
            v[eg] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (eg < gates) {
                const int et = eg * tpg + ew;
                for (int s = 0; s < ksl; ++s) {
                    const float4 t4 = *reinterpret_cast<const float4*>(red + ((et + tiles * s) * 16 + row) * PD_RED_LD + c4);
                    v[eg].x += t4.x; v[eg].y += t4.y; v[eg].z += t4.z; v[eg].w += t4.w;
                }
                if (ph.bias) {
                    const float4 b4 = *reinterpret_cast<const float4*>(ph.bias + ph.row0 + eg * PD_D + unit);
                    v[eg].x += b4.x; v[eg].y += b4.y; v[eg].z += b4.z; v[eg].w += b4.w;
                }
            }
        }
        const bool row_ok = b0 + row < B;
        const int b = b0 + row;
        float* hl = h_loc + row * 32 + cl;
        float* ul = u_loc + row * 32 + cl;
        float* rl = r_loc + row * 32 + cl;
        if (ph.epi == PD_ACT) {
            float4 o = v[0];
            o.x = apply_act(o.x, ph.act); o.y = apply_act(o.y, ph.act); o.z = apply_act(o.z, ph.act); o.w = apply_act(o.w, ph.act);
            if (row_ok) pd_st4(pd_rsrc(ph.out), (unsigned)(b * ph.ldo + unit) * 4u, o);
        } else if (ph.epi == PD_GATES || ph.epi == PD_CUDNN_RU) {
            float4 rr4, uu4;
            rr4.x = sigmoidf_(v[0].x); rr4.y = sigmoidf_(v[0].y); rr4.z = sigmoidf_(v[0].z); rr4.w = sigmoidf_(v[0].w);
            uu4.x = sigmoidf_(v[1].x); uu4.y = sigmoidf_(v[1].y); uu4.z = sigmoidf_(v[1].z); uu4.w = sigmoidf_(v[1].w);
            *reinterpret_cast<float4*>(ul) = uu4;             // u stays in this workgroup
            if (ph.epi == PD_GATES) {                          // r*h is the candidate's K operand: hand it over
                const float4 h4 = *reinterpret_cast<const float4*>(hl);
                rr4.x *= h4.x; rr4.y *= h4.y; rr4.z *= h4.z; rr4.w *= h4.w;
                if (row_ok) pd_st4(pd_rsrc(ph.out), (unsigned)(b * ph.ldo + unit) * 4u, rr4);
            } else {
                *reinterpret_cast<float4*>(rl) = rr4;
            }
        } else {   // PD_CAND / PD_CUDNN_HX: h' = u h + (1 - u) tanh(.)
            const float4 h4 = *reinterpret_cast<const float4*>(hl);
            const float4 u4 = *reinterpret_cast<const float4*>(ul);
            float4 cin = v[0];
            if (ph.epi == PD_CUDNN_HX) {   // v[0] = h Wch + bch, v[1] = x Wci + bci
                const float4 r4 = *reinterpret_cast<const float4*>(rl);
                cin.x = v[1].x + r4.x * v[0].x; cin.y = v[1].y + r4.y * v[0].y;
                cin.z = v[1].z + r4.z * v[0].z; cin.w = v[1].w + r4.w * v[0].w;
            }
            float4 hn;
            hn.x = u4.x * h4.x + (1.0f - u4.x) * tanhf_(cin.x);
            hn.y = u4.y * h4.y + (1.0f - u4.y) * tanhf_(cin.y);
            hn.z = u4.z * h4.z + (1.0f - u4.z) * tanhf_(cin.z);
            hn.w = u4.w * h4.w + (1.0f - u4.w) * tanhf_(cin.w);
            *reinterpret_cast<float4*>(hl) = hn;
            if (row_ok) pd_st4(pd_rsrc(ph.out), (unsigned)(b * ph.ldo + unit) * 4u, hn);
            if (ph.yout) {   // ResidualWrapper: y = x + h'; x is the first K segment of the staged tile
                const float4 x4 = *reinterpret_cast<const float4*>(As + row * PD_LDA + unit);
                hn.x += x4.x; hn.y += x4.y; hn.z += x4.z; hn.w += x4.w;
                if (row_ok) pd_st4(pd_rsrc(ph.yout), (unsigned)(b * ph.ldy + unit) * 4u, hn);
            }
        }
    }
    PD_STAMP(4)
    if (ph.more) __syncthreads();   // r / u are in LDS, the partial tiles may be overwritten
    else if (tid < tpg * 64) pd_publish_wave(cnt, PD_ARRIVALS / (unsigned)tpg);   // (tpg = 1 or 2 storing waves)
    PD_STAMP(5)
}

// Luong dot attention for rows 2j and 2j+1 of the cluster (TF-1.8 _luong_score / _compute_attention; dot form at
// reference attention.py:396-400): softmax over ALL Ts positions, context = alignments . memory.
// LocalLuongAttention (reference attention.py:32-342; decoder.hip has the launch-per-layer form): only the window
// of 2D+1 positions around the step index (monotonic) or around the predicted centre p = Ts sigmoid(v_p . tanh(W_p h))
// is scored; the reported alignments are zero outside it and, with `gaussian`, weighted as the reference writes it.
struct PdLocal {
    int d, gaussian, predictive, step;
    const float* wp; const float* vp;   // [256][256] (q @ W_p), [256]
    float* p_hist_t;                    // [B] predicted centres of this step
    int* err_flag;
};
template <bool LOCAL>   // the global form stays inline (one instance in the step loop); the windowed one is a call
__device__ __forceinline__ void pd_attention_body(const float* __restrict__ query, const float* __restrict__ keys,
                                             const float* __restrict__ values, float* ctx, float* align_t, int Ts, float* lds,
                                             int j, int b0, int B, unsigned* cnt, unsigned target, int* status,
                                             const PdLocal lc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = wave >> 3, hw = wave & 7, t512 = tid & 511;
    float* qs = lds + PD_OFF_AS + half * PD_D;                        // [2][256]
    float* part = lds + PD_OFF_AS + 2 * PD_D + half * (8 * PD_D);     // [2][8][256]
    float* redm = lds + PD_OFF_RED + half * 16;                       // [2][8] max, then [2][8] sum at +8... see below
    const int Tsp = (Ts + 3) & ~3;
    float* sc = lds + PD_OFF_SC + half * Tsp;
    int* ctrl = reinterpret_cast<int*>(lds + PD_OFF_CTRL);

    const int row = b0 + 2 * j + half;
    const bool row_ok = row < B;
    const int rr = row_ok ? row : B - 1;
#ifdef PD_ABL_ONE_MEMORY   // tools only (wrong results): every utterance attends over utterance 0's memory, 0.3 MB instead of 19 MB
    const int mr = 0;
#else
    const int mr = rr;
#endif

    // Scores: 16 lanes per key, 32 keys per pass of the row's 8 waves, PD_KB passes requested together (a pass per
    // round trip to L2 / the Infinity Cache was five dependent trips per step at Ts = 150).  The keys do not depend on
    // the query: with the global form the first PD_KB passes are requested BEFORE the wait for the cluster.
    const int sub = lane >> 4, l16 = lane & 15;
    float4 kpre[PD_KB][4];
    auto load_keys = [&](const float* kbase, int j0, int n_pos) {
#pragma unroll
        for (int q = 0; q < PD_KB; ++q) {
            const int jj = j0 + 32 * q + hw * 4 + sub;
            const float* kr = kbase + (size_t)(jj < n_pos ? jj : n_pos - 1) * PD_D;   // clamped, never branched on
#pragma unroll
            for (int i = 0; i < 4; ++i) kpre[q][i] = *reinterpret_cast<const float4*>(kr + (l16 + 16 * i) * 4);
        }
    };
    if (!LOCAL) load_keys(keys + (size_t)mr * Ts * PD_D, 0, Ts);

    PD_STAMP(0)
    pd_wait(cnt, target, status, ctrl);
    PD_STAMP(1)

    if (t512 < 64) *reinterpret_cast<float4*>(qs + 4 * t512) = pd_ld4(pd_rsrc(query), (unsigned)(rr * PD_D + 4 * t512) * 4u);
    __syncthreads();

    // scored positions [w_lo, w_lo + w_n): the whole memory, or the local window
    int w_lo = 0, w_n = Ts;
    float pc = 0.f;   // window centre as the gaussian sees it
    if (LOCAL && lc.d > 0) {
        w_n = 2 * lc.d + 1;
        if (lc.predictive) {
            // (q W_p)[n]: thread (n, half of k); then v_p . tanh(.) over the row's 512 threads
            const int n = t512 & 255, kh = t512 >> 8;
            float a0 = 0.f, a1 = 0.f;
            const float* wpn = lc.wp + (size_t)(128 * kh) * PD_D + n;
            for (int k = 0; k < 128; k += 2) {
                a0 = fmaf(qs[128 * kh + k], wpn[(size_t)k * PD_D], a0);
                a1 = fmaf(qs[128 * kh + k + 1], wpn[(size_t)(k + 1) * PD_D], a1);
            }
            part[kh * PD_D + n] = a0 + a1;
            __syncthreads();
            float v = 0.f;
            if (t512 < 256) v = tanhf_(part[t512] + part[PD_D + t512]) * lc.vp[t512];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) redm[hw] = v;   // waves 4..7 of the row hold zeros
            __syncthreads();
            const float p = (float)Ts * sigmoidf_((redm[0] + redm[1]) + (redm[2] + redm[3]));
            const int c = (int)floorf(p);
            // a window that leaves the memory: the reference's padding arithmetic fails there (decoder.hip)
            if (t512 == 0 && row_ok) {
                lc.p_hist_t[row] = p;
                if (c - lc.d < 0 || c + lc.d + 1 > Ts) *lc.err_flag = 1;
            }
            w_lo = min(max(c - lc.d, 0), Ts - w_n);
            pc = p;
            __syncthreads();   // part / redm are reused below
        } else {
            int c = lc.step > lc.d ? lc.step : lc.d;
            const int hi = Ts - (lc.d + 1);
            c = c < hi ? c : hi;
            w_lo = c - lc.d;
            pc = (float)c;
        }
    }

    const float* kb = keys + ((size_t)mr * Ts + w_lo) * PD_D;
    for (int j0 = 0; j0 < w_n; j0 += 32 * PD_KB) {
        if (LOCAL || j0 > 0) load_keys(kb, j0, w_n);
#pragma unroll
        for (int q = 0; q < PD_KB; ++q) {
            const int jj = j0 + 32 * q + hw * 4 + sub;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 kv = kpre[q][i];
                const float4 qv = *reinterpret_cast<const float4*>(qs + (l16 + 16 * i) * 4);
                s = fmaf(kv.x, qv.x, s);
                s = fmaf(kv.y, qv.y, s);
                s = fmaf(kv.z, qv.z, s);
                s = fmaf(kv.w, qv.w, s);
            }
            s += __shfl_xor(s, 8);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 1);
            if (jj < w_n && l16 == 0) sc[jj] = s;
        }
    }
    __syncthreads();

    PD_STAMP(2)
    float m = -INFINITY;
    for (int jj = t512; jj < w_n; jj += 512) m = fmaxf(m, sc[jj]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) redm[hw] = m;
    __syncthreads();
    m = redm[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) m = fmaxf(m, redm[i]);
    float sum = 0.f;
    for (int jj = t512; jj < w_n; jj += 512) {
        const float e = __expf(sc[jj] - m);
        sc[jj] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    __syncthreads();   // everyone has read the maxima
    if (lane == 0) redm[hw] = sum;
    __syncthreads();
    sum = ((redm[0] + redm[1]) + (redm[2] + redm[3])) + ((redm[4] + redm[5]) + (redm[6] + redm[7]));
    const float inv = 1.0f / sum;

    PD_STAMP(3)
    // context: wave hw takes positions hw, hw + 8, ...; lane d4 owns 4 consecutive depth elements (1 KB rows, coalesced);
    // PD_VB rows requested together (two per trip were ten dependent trips per step at Ts = 150)
    const float* vb = values + ((size_t)mr * Ts + w_lo) * PD_D + 4 * lane;
    float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = hw; j0 < w_n; j0 += 8 * PD_VB) {
        float4 vv[PD_VB];
#pragma unroll
        for (int q = 0; q < PD_VB; ++q) {
            const int jj = j0 + 8 * q;
            vv[q] = *reinterpret_cast<const float4*>(vb + (size_t)(jj < w_n ? jj : w_n - 1) * PD_D);
        }
#pragma unroll
        for (int q = 0; q < PD_VB; ++q) {
            const int jj = j0 + 8 * q;
            const float e = jj < w_n ? sc[jj] : 0.f;
            c0.x = fmaf(e, vv[q].x, c0.x); c0.y = fmaf(e, vv[q].y, c0.y); c0.z = fmaf(e, vv[q].z, c0.z); c0.w = fmaf(e, vv[q].w, c0.w);
        }
    }
    *reinterpret_cast<float4*>(part + hw * PD_D + 4 * lane) = c0;
    __syncthreads();
    if (t512 < 64) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const float4 t4 = *reinterpret_cast<const float4*>(part + w * PD_D + 4 * t512);
            a.x += t4.x; a.y += t4.y; a.z += t4.z; a.w += t4.w;
        }
        a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
        if (row_ok) pd_st4(pd_rsrc(ctx), (unsigned)(row * PD_D + 4 * t512) * 4u, a);
    }
    if (align_t && row_ok) {
        // the reference pads the window back to the memory length (attention.py:85-92) and, with `gaussian`, weights
        // it by exp(-(j - p)^2 / 2 * (D/2)^2) as written at attention.py:73-80 (the context uses the plain softmax)
        const float gk = 0.5f * (0.5f * lc.d) * (0.5f * lc.d);
        for (int k = t512; k < Ts; k += 512) {
            const int jw = k - w_lo;
            float a = 0.f;
            if (jw >= 0 && jw < w_n) {
                a = sc[jw] * inv;
                if (LOCAL && lc.d > 0 && lc.gaussian) {
                    const float dist = (float)k - pc;
                    a *= __expf(-(dist * dist) * gk);
                }
            }
            align_t[(size_t)row * Ts + k] = a;
        }
    }
    PD_STAMP(4)
    pd_publish(cnt);
    PD_STAMP(5)
}
__device__ __attribute__((noinline)) void pd_attention_local(const float* query, const float* keys, const float* values, float* ctx,
                                                             float* align_t, int Ts, float* lds, int j, int b0, int B, unsigned* cnt,
                                                             unsigned target, int* status, const PdLocal lc) {
    pd_attention_body<true>(query, keys, values, ctx, align_t, Ts, lds, j, b0, B, cnt, target, status, lc);
}

__global__ __launch_bounds__(PD_THREADS) void dec_persistent_kernel(PdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int cluster = blockIdx.x / PD_W, j = blockIdx.x - cluster * PD_W;
    const int b0 = cluster * 16;
    unsigned* cnt = p.counters + 64 * cluster;
    int* ctrl = reinterpret_cast<int*>(lds + PD_OFF_CTRL);
    for (int i = tid; i < 3 * 16 * 32 + 2 * 16 * 32; i += PD_THREADS) lds[PD_OFF_H + i] = 0.f;   // zero_state
    if (tid == 0) {
        ctrl[0] = 0;
        // all workgroups resident: the CUs the call pipeline held for this stream are no longer needed
        const unsigned n = __hip_atomic_fetch_add(p.resident, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n + 1 == gridDim.x && p.hold_flag) __hip_atomic_store(p.hold_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();

    const int yld = p.n_steps * PD_D;
    unsigned g = 0;   // phases completed by the cluster
    for (int t = 0; t < p.n_steps; ++t) {
        // The recurrent states are double-buffered by step parity: step t reads h[t & 1] and writes h[(t + 1) & 1].
        // In the CudnnCompatibleGRUCell form the candidate phase continues on the staged tile without a wait, so a
        // workgroup stores its slice of h' while a peer that left the previous wait a little later may still be staging
        // the full previous h: with one buffer that peer would read a mix of h_{t-1} and h_t.
        const float* h_att_old = p.h_att2[t & 1];
        float* h_att_new = p.h_att2[(t + 1) & 1];
        // ONE instance of the phase body in a loop over the step's ten phases (ten inlined copies spill)
#pragma nounroll
        for (int k = 0; k < 10; ++k) {
#ifdef PD_TIMELINE
            if (threadIdx.x == 0) { pd_tl_step = t; pd_tl_phase = k; pd_tl_lds = lds + ((PD_OFF_SC + 2 * ((p.Ts + 3) & ~3) + 1) & ~1); }
#endif
            if (k == 4) {
                PdLocal lc;
                lc.d = p.local_d; lc.gaussian = p.local_gaussian; lc.predictive = p.local_predictive; lc.step = t;
                lc.wp = p.local_wp; lc.vp = p.local_vp; lc.p_hist_t = p.p_hist ? p.p_hist + (size_t)t * p.B : nullptr;
                lc.err_flag = p.err_flag;
                float* align_t = p.align ? p.align + (size_t)t * p.B * p.Ts : nullptr;
                if (p.local_d > 0) pd_attention_local(h_att_new, p.keys, p.memory, p.ctx, align_t, p.Ts, lds, j, b0, p.B, cnt, PD_ARRIVALS * PD_W * g, p.status, lc);
                else pd_attention_body<false>(h_att_new, p.keys, p.memory, p.ctx, align_t, p.Ts, lds, j, b0, p.B, cnt, PD_ARRIVALS * PD_W * g, p.status, lc);
                ++g;
                continue;
            }
            PdPhase ph;
            ph.lda0 = PD_D; ph.k0 = PD_D; ph.lda1 = PD_D; ph.K = 2 * PD_D; ph.ub = 32; ph.act = ACT_NONE; ph.layer = 0;
            ph.ldo = PD_D; ph.yout = nullptr; ph.ldy = PD_D; ph.bias = nullptr; ph.row0 = 0; ph.cont = 0; ph.more = 0;
            ph.delay = p.dbg_delay;
            switch (k) {
                case 0:
                    // PrenetWrapper on concat([x_t, attention_{t-1}]) (wrappers.py:122-124); x_0 = GO frame = zeros
                    // (helpers.py:108), x_t = (y_{t-1} W_o + b_o)[-n_mels:] folded into the pre-net matrix (decoder.hip)
                    ph.a0 = t == 0 ? nullptr : p.yhist + (size_t)(t - 1) * PD_D; ph.lda0 = yld; ph.k0 = t == 0 ? p.n_mels : PD_D;
                    ph.a1 = p.att; ph.K = ph.k0 + PD_D;
                    ph.Wt = t == 0 ? p.w1 : p.w1f; ph.bias = t == 0 ? p.b1 : p.b1f;
                    ph.epi = PD_ACT; ph.act = ACT_RELU; ph.out = p.p1;
                    break;
                case 1:
                    ph.a0 = p.p1; ph.a1 = p.p1; ph.K = PD_D; ph.Wt = p.w2; ph.bias = p.b2; ph.ub = 16;
                    ph.epi = PD_ACT; ph.act = ACT_RELU; ph.out = p.p2; ph.ldo = PD_P2;
                    break;
                case 2:   // attention GRU (model.py:226-229): gates on [p2 ; h_att]
                    ph.a0 = p.p2; ph.lda0 = PD_P2; ph.k0 = PD_P2; ph.a1 = h_att_old; ph.K = PD_P2 + PD_D;
                    ph.Wt = p.ag_w; ph.bias = p.ag_b; ph.epi = p.cudnn ? PD_CUDNN_RU : PD_GATES; ph.out = p.rh; ph.more = p.cudnn;
                    break;
                case 3:   // ... candidate (GRUCell: on [p2 ; r*h_att], after a hop); the new state is the attention query
                    ph.a0 = p.p2; ph.lda0 = PD_P2; ph.k0 = PD_P2; ph.a1 = p.rh; ph.K = PD_P2 + PD_D;
                    ph.Wt = p.cudnn ? p.ag_w : p.ac_w; ph.bias = p.cudnn ? p.ag_b : p.ac_b; ph.out = h_att_new;
                    ph.epi = p.cudnn ? PD_CUDNN_HX : PD_CAND; ph.row0 = p.cudnn ? 2 * PD_D : 0; ph.cont = p.cudnn;
                    break;
                case 5:   // attention_layer(concat([cell_output, context])), no bias
                    ph.a0 = h_att_new; ph.a1 = p.ctx; ph.Wt = p.al_w; ph.epi = PD_ACT; ph.out = p.att;
                    break;
                default: {   // 6..9: two ResidualWrapper(GRU cell) layers (model.py:254-269); the top one writes the y history
                    const int l = (k - 6) >> 1;
                    const bool second = (k - 6) & 1;
                    ph.a0 = l == 0 ? p.att : p.y0; ph.a1 = (second && !p.cudnn) ? p.rh : p.h_dec2[l][t & 1];
                    ph.layer = 1 + l;
                    if (p.cudnn) {
                        ph.Wt = p.g_gw[l]; ph.bias = p.g_gb[l];
                        ph.epi = second ? PD_CUDNN_HX : PD_CUDNN_RU; ph.row0 = second ? 2 * PD_D : 0;
                        ph.cont = second; ph.more = !second;
                    } else {
                        ph.Wt = second ? p.g_cw[l] : p.g_gw[l]; ph.bias = second ? p.g_cb[l] : p.g_gb[l];
                        ph.epi = second ? PD_CAND : PD_GATES;
                    }
                    ph.out = second ? p.h_dec2[l][(t + 1) & 1] : p.rh;
                    if (second) { ph.yout = l == 0 ? p.y0 : p.yhist + (size_t)t * PD_D; ph.ldy = l == 0 ? PD_D : yld; }
                } break;
            }
            pd_phase(ph, lds, j, b0, p.B, cnt, PD_ARRIVALS * PD_W * g, p.status);
            if (!ph.more) ++g;
        }
    }
#ifdef PD_TIMELINE
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 80 && pd_dbg) pd_dbg[threadIdx.x] = reinterpret_cast<unsigned long long*>(pd_tl_lds)[threadIdx.x];
#endif
}

bool decoder_persistent_supports(const DecoderWeights& w, int cudnn, int B, int Ts) {
    (void)cudnn;   // both GRU formulations
    if (w.local_d > 0 && Ts < 2 * w.local_d + 1) return false;
    return w.n_layers == 2 && w.att_units == PD_D && w.dec_units == PD_D && w.mem_units == PD_D &&
           w.prenet1_units == PD_D && w.prenet2_units == PD_P2 && w.n_mels % 16 == 0 && w.n_mels <= PD_D && B >= 1 && Ts >= 1 &&
           pd_lds_bytes(Ts) <= 160 * 1024 && (size_t)B * Ts * PD_D * 4 < 0xFFFFFFF0ull;
}

int decoder_persistent_workgroups(int B) { return PD_W * ((B + 15) / 16); }

hipError_t decoder_persistent_configure() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024 - 64);
}

// Capturable: two memsets and one launch.  `sync` = 64 unsigned per cluster + 1 (resident count) + 1 (status word,
// zeroed by the caller when the buffer is created and after it has been read).
hipError_t decoder_persistent_enqueue(hipStream_t s, const DecoderWeights& w, const DecoderScratch& sc, const float* memory,
                                      const float* keys, int B, int Ts, int n_steps, float* align, unsigned* sync,
                                      int* hold_flag, int cudnn, int dbg_delay) {
    const int clusters = (B + 15) / 16;
    hipError_t e;
    if ((e = hipMemsetAsync(sc.state, 0, sc.state_bytes, s)) != hipSuccess) return e;
    // counters and resident count start at zero for every launch; the status word behind them is sticky (the host
    // clears it when it has read it, api_handle.hip), so a timeout is not lost when several calls are queued before a sync
    if ((e = hipMemsetAsync(sync, 0, ((size_t)64 * clusters + 1) * sizeof(unsigned), s)) != hipSuccess) return e;
    PdParams p;
    p.w1 = w.prenet1_wt; p.b1 = w.prenet1_b; p.w1f = w.prenet1f_wt; p.b1f = w.prenet1f_b; p.w2 = w.prenet2_wt; p.b2 = w.prenet2_b;
    p.ag_w = w.att_gru.gates_wt; p.ag_b = w.att_gru.gates_b; p.ac_w = w.att_gru.cand_wt; p.ac_b = w.att_gru.cand_b;
    p.al_w = w.attn_layer_wt;
    for (int l = 0; l < 2; ++l) {
        p.g_gw[l] = w.gru[l].gates_wt; p.g_gb[l] = w.gru[l].gates_b; p.g_cw[l] = w.gru[l].cand_wt; p.g_cb[l] = w.gru[l].cand_b;
        p.h_dec2[l][0] = sc.h_dec[l]; p.h_dec2[l][1] = sc.h_dec_alt[l];
    }
    p.h_att2[0] = sc.h_att; p.h_att2[1] = sc.h_att_alt;
    p.dbg_delay = dbg_delay;
    p.memory = memory; p.keys = keys;
    p.att = sc.att; p.p1 = sc.p1; p.p2 = sc.p2; p.rh = sc.rh; p.ctx = sc.ctx_parts; p.y0 = sc.y0;
    p.yhist = sc.yhist; p.align = align;
    p.counters = sync; p.resident = sync + 64 * clusters; p.status = reinterpret_cast<int*>(sync + 64 * clusters + 1);
    p.hold_flag = hold_flag;
    p.B = B; p.Ts = Ts; p.n_steps = n_steps; p.n_mels = w.n_mels; p.cudnn = cudnn;
    p.local_d = w.local_d; p.local_gaussian = w.local_gaussian; p.local_predictive = w.local_d > 0 && w.local_predictive;
    p.local_wp = w.local_wp; p.local_vp = w.local_vp; p.p_hist = sc.p_hist; p.err_flag = sc.err_flag;
    if (p.local_predictive && (e = hipMemsetAsync(sc.err_flag, 0, sizeof(int), s)) != hipSuccess) return e;
#ifdef PD_TIMELINE
    {
        static unsigned long long* dbg = nullptr;
        if (dbg) {
            (void)hipStreamSynchronize(s);
            unsigned long long hst[80];
            (void)hipMemcpy(hst, dbg, sizeof(hst), hipMemcpyDeviceToHost);
            for (int k = 0; k < 10; ++k) {
                fprintf(stderr, "phase %d:", k);
                for (int i = 0; i < 6; ++i) fprintf(stderr, " [%d]%.2f", i, (double)(hst[k * 8 + i] - hst[0]) / 100.0);
                fprintf(stderr, "\n");
            }
        } else {
            (void)hipMalloc(&dbg, 80 * sizeof(unsigned long long));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(pd_dbg), &dbg, sizeof(dbg));
        }
    }
#endif
    hipLaunchKernelGGL(dec_persistent_kernel, dim3(PD_W * clusters), dim3(PD_THREADS), pd_lds_bytes(Ts), s, p);
    return hipGetLastError();
}

}  // namespace tts
