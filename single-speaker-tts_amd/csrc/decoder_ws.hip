// Weight-stationary persistent decoder (gfx950, round 5): the 200-step Luong-attention GRU decoder loop as ONE launch
// whose workgroups keep their share of every layer's weights in REGISTERS for the whole loop.
//
// Same arithmetic as decoder.hip / decoder_persistent.hip (reference tacotron/model.py:191-331, wrappers.py:94-124,
// helpers.py:83-110,161-205): the TF-1.8 GRUCell form or the CudnnCompatibleGRUCell form (template CUDNN), global
// LuongAttention or LocalLuongAttention with monotonic / predictive windows (template LOCAL; attention.py:20-265), 32 or 16
// utterances per cluster (template M, same bits).  What changed against decoder_persistent.hip:
// there a cluster of 8 workgroups x 16 utterances re-streamed its share of the 6 MB of weights from L2 in EVERY step
// (128 KB per workgroup and phase: 101 M L2 read requests = 12.6 GB per launch for an algorithmic 44 MB, 20 of the 78 us of
// a step by ablation: profiles/r04_decoder_l2_counters.txt, HISTORY.md part C).  Here
//   * a CLUSTER is 16 workgroups (one per CU, 512 threads = 2 waves per SIMD, 256 registers per lane) x M = 32 or 16 utterances;
//     workgroup j owns units [16 j, 16 j + 16) of every 256-unit layer (for a GRU its r AND u columns: the update gate
//     and the cell state of its units never leave its LDS) and units [8 j, 8 j + 8) of pre-net 2;
//   * its weights -- 5.25 MB / 16 = 336 KB, 172 registers per lane (WS_NREG) -- are loaded ONCE, from an image that
//     tts_finalize_weights lays out in register order (decoder_ws_pack: lane (n, q) of wave w holds W[n][k(w, reg, q)]), and
//     are the B operands of v_mfma_f32_16x16x4_f32 straight from the register file; K is split over the 8 waves (over 4 for
//     the two gate tiles of a GRU), M / 16 16-row blocks per wave;
//   * the hand-off buffers are in BLOCK FORMAT [producer workgroup][row 0..31][its units]: a producer's epilogue writes
//     whole 128-byte lines (two rows x 16 units, eight lanes of one store instruction), a consumer stages the cluster's
//     A tile with a linear copy (sc1 16-byte loads, no address arithmetic, conflict-free ds_write_b128), and a 16-deep k
//     chunk of the tile is one contiguous 1 KB block for the MFMA fragment reads;
//   * hand-off protocol as in decoder_persistent.hip (MI355X_MICROARCH.md, valid forms): every handed-off byte stored
//     sc1, each storing wave waits vmcnt(0) and adds to the cluster's counter for itself, consumers poll the counter with
//     an sc1 load in one lane, then a workgroup barrier, then sc1 loads.  No cache-wide release / acquire, no grid barrier;
//   * attention: workgroup j scores, normalises and contracts rows (M / 16) j ... of its cluster (4 waves each).
// Every wait is bounded; on a timeout the sticky status word is set and the grid drains.  All workgroups must be
// co-resident: 16 * ceil(B / M) compute units (api_stages.hip checks the budget).  Configurations this kernel does not cover
// (other layer sizes or counts) run in decoder_persistent.hip / decoder.hip.
#include "tts_common.h"
#include "decoder.h"
#include <cstring>
#include <cstdio>

namespace tts {

#define WS_W 16
#define WS_NW 8
#define WS_THREADS (WS_NW * 64)
#define WS_D 256
#define WS_P2 128
#define WS_RED_LD 20
#define WS_SPIN_LIMIT 2000000u
#ifndef WS_KB
#define WS_KB 2      // key passes (16 positions each) requested together per row
#endif
#ifndef WS_VB
#define WS_VB 8      // value rows per wave requested together
#endif

// weight registers per lane and phase: 4 * (K / k-slices / 16); the offsets of the phases in the register image
#define WS_R0 0      // pre-net 1 (folded)        K 512, 1 tile : 16
#define WS_R1 16     // pre-net 2                 K 256, 1 tile :  8
#define WS_R2 24     // attention GRU gates       K 384, 2 tiles: 24
#define WS_R3 48     // attention GRU candidate   K 384, 1 tile : 12  (CudnnCompatibleGRUCell: x W_ci over 128 / h W_ch over 256: 16)
#define WS_R5 64     // attention layer           K 512, 1 tile : 16
#define WS_R6 80     // GRU 1 gates               K 512, 2 tiles: 32
#define WS_R7 112    // GRU 1 candidate           K 512, 1 tile : 16
#define WS_R8 128    // GRU 2 gates                             : 32
#define WS_R9 160    // GRU 2 candidate                         : 16
static_assert(WS_R9 + 16 == DEC_WS_NREG, "register image size");

// Rows (utterances) per cluster, M: 32 (two 16-row blocks per wave: the form of round 5, 16 compute units per 32 utterances --
// what fits the call pipeline's 32 reserved units at B = 64) or 16 (round 6: one 16-row block per wave, twice the compute units
// per utterance and about 0.8 of the time per phase -- staging, MFMA and reduction halve, the waits and the drain do not -- for
// calls that have the chip to themselves).  A row's arithmetic does not see M: the same K slices in the same order, the same
// four waves per attention row, MFMA rows are independent -- the two forms give the same bits (tests/test_gpu_persistent.py).
// LDS map (floats) for M rows
#define WS_OFF_AS 0                                               // staged A tile, block format: M rows x <= 512 k
#define WS_OFF_RED(M) ((M) * 512)                                 // [8 waves][M / 16 row blocks][16][WS_RED_LD]
#define WS_OFF_H(M) (WS_OFF_RED(M) + WS_NW * ((M) / 16) * 16 * WS_RED_LD)   // [3 layers][M rows][16 units]: this workgroup's cell states
#define WS_OFF_U(M) (WS_OFF_H(M) + 3 * (M) * 16)                  // [M][16] update gate
#define WS_OFF_R(M) (WS_OFF_U(M) + (M) * 16)                      // [M][16] reset gate (CudnnCompatibleGRUCell form)
#define WS_OFF_BIAS(M) (WS_OFF_R(M) + (M) * 16)                   // [DEC_WS_BIAS_SLOTS][32]
#define WS_OFF_CTRL(M) (WS_OFF_BIAS(M) + DEC_WS_BIAS_SLOTS * 32)
#define WS_OFF_SC(M) (WS_OFF_CTRL(M) + 16)                        // [2][Ts padded] scores

size_t ws_lds_bytes(int Ts, int M) { return ((size_t)WS_OFF_SC(M) + 2 * (size_t)((Ts + 3) & ~3)) * sizeof(float); }

// per-cluster buffers (floats): the zeroed state block, then the plain hand-off buffers
#define WS_BUF(M) ((M) * WS_D)                                    // one 256-unit vector of a cluster
// att | h_att x 2 | h_dec1 x 2 | h_dec2 x 2 | y (top layer output).  The cell states are double-buffered by step parity: in
// the CudnnCompatibleGRUCell form a workgroup stores its slice of h' in the phase that staged h, while a peer that left the
// wait a little later may still be staging it (decoder_persistent.hip, tests/test_gpu_cudnn_variant.py: the late stager)
#define WS_STATE_FLOATS(M) (8 * WS_BUF(M))
#define WS_REST_FLOATS(M) (4 * WS_BUF(M) + (M) * WS_P2)          // p1 | rh | ctx | y0 | p2

typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned ws_u32x4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)0xFFFFFFF0u, 0x00020000);
}
// 16-byte sc1 (write-through / L1-bypassing) accesses: aux bit 4
__device__ __forceinline__ float4 ws_ld4(const __amdgpu_buffer_rsrc_t& rs, unsigned byte_off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16));
}
__device__ __forceinline__ void ws_st4(const __amdgpu_buffer_rsrc_t& rs, unsigned byte_off, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, v), rs, (int)byte_off, 0, 16);
}

enum WsEpi { WS_ACT = 0, WS_GATES = 1, WS_CAND = 2, WS_CUDNN_RU = 3 };

#ifdef WS_TIMELINE   // tools only: s_memrealtime stamps (100 MHz) of workgroup 0, thread 0 in step 100: [phase 0..9][8]
__device__ unsigned long long ws_dbg[10 * 8];
__device__ __shared__ int ws_tl_step, ws_tl_phase;
#define WS_STAMP(I) if (blockIdx.x == 0 && threadIdx.x == 0 && ws_tl_step == 100) ws_dbg[ws_tl_phase * 8 + (I)] = __builtin_amdgcn_s_memrealtime();
#define WS_TL_PHASE(T, K) if (threadIdx.x == 0) { ws_tl_step = (T); ws_tl_phase = (K); }
#else
#define WS_STAMP(I)
#define WS_TL_PHASE(T, K)
#endif

// Start of a phase: wait until `target` arrivals have been counted on the cluster's counter (one lane polls, everybody
// meets at the barrier).
__device__ __forceinline__ void ws_wait(unsigned* cnt, unsigned target, int* status, int* ctrl) {
    if (threadIdx.x == 0 && target > 0) {
        if (ctrl[0] == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023u) == 0 &&
                    (spins > WS_SPIN_LIMIT || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ctrl[0] = 1;   // drain: no further waits in this workgroup
                    break;
                }
            }
        }
    }
    __syncthreads();
}
// End of a phase, per storing wave: its (write-through) stores have drained, then its arrival (agent scope).  A workgroup
// counts WS_ARRIVALS per phase, however many of its waves store.
#define WS_ARRIVALS 2u
__device__ __forceinline__ void ws_publish_wave(unsigned* cnt, unsigned arrivals) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct WsPhase {
    const float* a0; const float* a1;   // the two segments of the A tile: a cluster's buffers, block format
    float* out;                         // WS_ACT: activations; WS_GATES: r*h; WS_CAND: the new state
    float* yout;                        // WS_CAND with residual: y = x + h' (block format), or null
    float* yhist; int yld;              // ... and the same rows in the y history [B][n_steps][256] (+ t * 256), or null
    int bias_slot, layer;
    const float* next1;                 // NEXT1: segment 1 of the phase after this one (a cluster's buffer, 256 units)
    unsigned target;
    int delay;                          // tests only: workgroup 3 stages late
};

// LDS offset (floats) of element (row r of row block 0, k = kk + 4 q) of the staged tile, kk a multiple of 16; the
// distance to row block 1 in *rbs.  Segment 0 holds K0 columns in blocks of UB0 units per producer, segment 1 blocks of 16.
template <int M, int K0, int UB0>
__device__ __forceinline__ int ws_a_off(int kk, int r, int q, int* rbs) {
    if (kk < K0) {
        if (UB0 == 16) { *rbs = 256; return (kk >> 4) * (M * 16) + r * 16 + 4 * q; }
        *rbs = 128;
        return ((kk >> 3) + (q >> 1)) * (M * 8) + r * 8 + 4 * (q & 1);
    }
    *rbs = 256;
    return M * K0 + ((kk - K0) >> 4) * (M * 16) + r * 16 + 4 * q;
}

// One GEMM-shaped phase: out[32 rows][this workgroup's UBO units (x TILES gates)] = epi([a0 | a1] . W^T + bias), the
// weights in registers w[ROFF ...].
// KEEP0: segment 0 of the tile is what the phase before staged there (a GRU's candidate phase behind its gates phase: the
// cell's input x) -- it is not loaded again, and the waves whose K slice lies inside it run their MFMAs BEFORE the wait for
// the cluster: they only need x and the weights.  SROT rotates the wave -> K slice map so that those are waves 4..7: waves
// 0 and 1 come out of the previous phase's epilogue last, and lane 0 of wave 0 polls.
// EARLY1: segment 1 of the tile is a value the cluster finished long ago -- a cell's own previous state under its gates, the
// previous step's attention under pre-net 1 -- and was requested by the phase BEFORE this one (NEXT1 there: four 16-byte loads
// per thread behind that phase's MFMAs, in flight during its epilogue and its publish, handed over in `pre`): it is written to
// LDS at once, and the waves whose K slices lie inside it (4..7) run their MFMAs while lane 0 of wave 0 already polls for the
// cluster; behind the wait only segment 0 is staged and only waves 0..3 multiply -- one wave per SIMD instead of two.
template <int M, int K0, int UB0, int K1, int TILES, int UBO, int EPI, int ACT, int ROFF, bool KEEP0 = false, int SROT = 0, bool EARLY1 = false,
          bool NEXT1 = false>
__device__ __forceinline__ void ws_phase(const float (&w)[DEC_WS_NREG], const WsPhase& ph, float* lds, int j, int b0, int B,
                                         unsigned* cnt, int* status, float4 (&pre)[4]) {
    static_assert(!EARLY1 || (K1 == WS_D && !KEEP0), "EARLY1: a 256-unit second segment");
    static_assert(M == 32 || M == 16, "rows per cluster");
    constexpr int RB = M / 16;                       // 16-row blocks per wave
    constexpr int PRE = WS_D * M / 4 / WS_THREADS;   // 16-byte pieces per thread of a 256-unit buffer (the early segment)
    // 16-byte pieces per thread of the whole staged tile / of its segment 0
    constexpr int K = K0 + K1, KSL = WS_NW / TILES, KW = K / KSL, CH = KW / 16, NLD = K * M / 4 / WS_THREADS, NLD0 = K0 * M / 4 / WS_THREADS;
    static_assert(KW % 16 == 0 && (K * M) % (4 * WS_THREADS) == 0 && (K0 * M) % (4 * WS_THREADS) == 0 && K0 % 16 == 0, "phase shape");
    // (the thread index is made opaque per phase: every address below is a function of it alone, and hoisted out of the
    //  step loop for all nine phases at once those addresses -- not the weights -- were what the register allocator spilled)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    float* As = lds + WS_OFF_AS;
    float* red = lds + WS_OFF_RED(M);
    float* h_loc = lds + WS_OFF_H(M) + ph.layer * (M * 16);
    float* u_loc = lds + WS_OFF_U(M);
    float* r_loc = lds + WS_OFF_R(M);
    const float* bias = lds + WS_OFF_BIAS(M) + ph.bias_slot * 32;
    int* ctrl = reinterpret_cast<int*>(lds + WS_OFF_CTRL(M));

    // ---- this wave's K slice of one 16-column tile, every 16-row block of the cluster (two for M = 32, one for 16)
    const int tile = wave % TILES, slice = (wave / TILES + SROT) % KSL;
    const int kb = slice * KW;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    auto mma_slice = [&]() {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            int rbs;
            const int off = ws_a_off<M, K0, UB0>(kb + 16 * c, r, q, &rbs);
            const float4 a0v = *reinterpret_cast<const float4*>(As + off);
            if (RB == 2) {
                const float4 a1v = *reinterpret_cast<const float4*>(As + off + rbs);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.x, w[ROFF + 4 * c + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v.x, w[ROFF + 4 * c + 0], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.y, w[ROFF + 4 * c + 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v.y, w[ROFF + 4 * c + 1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.z, w[ROFF + 4 * c + 2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v.z, w[ROFF + 4 * c + 2], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.w, w[ROFF + 4 * c + 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v.w, w[ROFF + 4 * c + 3], acc1, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.x, w[ROFF + 4 * c + 0], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.y, w[ROFF + 4 * c + 1], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.z, w[ROFF + 4 * c + 2], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.w, w[ROFF + 4 * c + 3], acc0, 0, 0, 0);
            }
        }
    };
    // KEEP0: the slices inside segment 0 need nothing the cluster is still working on (wave-uniform); EARLY1: those inside
    // segment 1, once it is in LDS
    const bool early = (KEEP0 && kb + KW <= K0) || (EARLY1 && kb >= K0);
    WS_STAMP(0)
    if (EARLY1) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) *reinterpret_cast<float4*>(As + 4 * (tid + WS_THREADS * (u + NLD0))) = pre[u];
        __syncthreads();
    }
    if (early) mma_slice();

    ws_wait(cnt, ph.target, status, ctrl);
    WS_STAMP(1)
    if (ph.delay && j == 3)   // a late stager: what a workgroup that clears its poll late looks like to its peers
        for (int i = 0; i < ph.delay; ++i) __builtin_amdgcn_s_sleep(127);

    // ---- stage the cluster's A tile: a linear copy of the (at most two) block-format buffers, all loads in flight together
    // (KEEP0: segment 0 is in place)
    {
        constexpr int U0 = KEEP0 ? NLD0 : 0, U1 = EARLY1 ? NLD0 : NLD;
        const __amdgpu_buffer_rsrc_t r0 = ws_rsrc(ph.a0), r1 = ws_rsrc((K1 > 0 && !EARLY1) ? ph.a1 : ph.a0);
        float4 sv[NLD];
#pragma unroll
        for (int u = U0; u < U1; ++u)
            sv[u] = u < NLD0 ? ws_ld4(r0, (unsigned)(tid + WS_THREADS * u) * 16u)
                             : ws_ld4(r1, (unsigned)(tid + WS_THREADS * (u - NLD0)) * 16u);
#pragma unroll
        for (int u = U0; u < U1; ++u) *reinterpret_cast<float4*>(As + 4 * (tid + WS_THREADS * u)) = sv[u];
        __syncthreads();
    }
    WS_STAMP(2)
    if (!early) mma_slice();
    WS_STAMP(3)
    // C/D map of 16x16: col = lane & 15, row = (lane >> 4) * 4 + reg.  The partial tiles are indexed by (slice, tile).
    const int rslot = slice * TILES + tile;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[((rslot * RB + 0) * 16 + q * 4 + i) * WS_RED_LD + r] = acc0[i];
        if (RB == 2) red[((rslot * RB + 1) * 16 + q * 4 + i) * WS_RED_LD + r] = acc1[i];
    }
    __syncthreads();
    WS_STAMP(4)
    if (NEXT1) {   // the next phase's early segment: requested now, in flight during this phase's epilogue and publish
        const __amdgpu_buffer_rsrc_t rn = ws_rsrc(ph.next1);
#pragma unroll
        for (int u = 0; u < PRE; ++u) pre[u] = ws_ld4(rn, (unsigned)(tid + WS_THREADS * u) * 16u);
    }

    // ---- epilogue: thread e owns (row, 4 consecutive units) of every gate; the K slices are added in a fixed order.
    // Eight consecutive threads cover one 128-byte line of the output block (two rows x 16 units, four rows x 8).
    constexpr int NT = M * UBO / 4;                // 128 threads (two waves), 64 or (M = 16, pre-net 2) 32
    if (tid < NT) {
        const int row = UBO == 16 ? tid >> 2 : tid >> 1;
        const int c4 = UBO == 16 ? (tid & 3) * 4 : (tid & 1) * 4;
        const int rb = row >> 4, rr = row & 15;
        float4 v[TILES];
#pragma unroll
        for (int g = 0; g < TILES; ++g) {
            v[g] = *reinterpret_cast<const float4*>(bias + g * 16 + c4);
#pragma unroll
            for (int s = 0; s < KSL; ++s) {
                const float4 t4 = *reinterpret_cast<const float4*>(red + (((s * TILES + g) * RB + rb) * 16 + rr) * WS_RED_LD + c4);
                v[g].x += t4.x; v[g].y += t4.y; v[g].z += t4.z; v[g].w += t4.w;
            }
        }
        const unsigned ooff = (unsigned)((j * M + row) * UBO + c4) * 4u;   // byte offset inside a cluster's buffer
        float* hl = h_loc + row * 16 + c4;
        float* ul = u_loc + row * 16 + c4;
        if (EPI == WS_CUDNN_RU) {   // CudnnCompatibleGRUCell: r and u stay in this workgroup, ws_phase_hx continues on the tile
            float4 rr4, uu4;
            rr4.x = sigmoidf_(v[0].x); rr4.y = sigmoidf_(v[0].y); rr4.z = sigmoidf_(v[0].z); rr4.w = sigmoidf_(v[0].w);
            uu4.x = sigmoidf_(v[TILES - 1].x); uu4.y = sigmoidf_(v[TILES - 1].y); uu4.z = sigmoidf_(v[TILES - 1].z); uu4.w = sigmoidf_(v[TILES - 1].w);
            *reinterpret_cast<float4*>(ul) = uu4;
            *reinterpret_cast<float4*>(r_loc + row * 16 + c4) = rr4;
        } else if (EPI == WS_ACT) {
            float4 o = v[0];
            o.x = apply_act(o.x, ACT); o.y = apply_act(o.y, ACT); o.z = apply_act(o.z, ACT); o.w = apply_act(o.w, ACT);
            ws_st4(ws_rsrc(ph.out), ooff, o);
        } else if (EPI == WS_GATES) {
            float4 rr4, uu4;
            rr4.x = sigmoidf_(v[0].x); rr4.y = sigmoidf_(v[0].y); rr4.z = sigmoidf_(v[0].z); rr4.w = sigmoidf_(v[0].w);
            uu4.x = sigmoidf_(v[TILES - 1].x); uu4.y = sigmoidf_(v[TILES - 1].y); uu4.z = sigmoidf_(v[TILES - 1].z); uu4.w = sigmoidf_(v[TILES - 1].w);
            *reinterpret_cast<float4*>(ul) = uu4;               // u stays in this workgroup
            const float4 h4 = *reinterpret_cast<const float4*>(hl);
            rr4.x *= h4.x; rr4.y *= h4.y; rr4.z *= h4.z; rr4.w *= h4.w;   // r*h is the candidate's K operand: hand it over
            ws_st4(ws_rsrc(ph.out), ooff, rr4);
        } else {   // WS_CAND: h' = u h + (1 - u) tanh(.)
            const float4 h4 = *reinterpret_cast<const float4*>(hl);
            const float4 u4 = *reinterpret_cast<const float4*>(ul);
            float4 hn;
            hn.x = u4.x * h4.x + (1.0f - u4.x) * tanhf_(v[0].x);
            hn.y = u4.y * h4.y + (1.0f - u4.y) * tanhf_(v[0].y);
            hn.z = u4.z * h4.z + (1.0f - u4.z) * tanhf_(v[0].z);
            hn.w = u4.w * h4.w + (1.0f - u4.w) * tanhf_(v[0].w);
            *reinterpret_cast<float4*>(hl) = hn;
            ws_st4(ws_rsrc(ph.out), ooff, hn);
            if (ph.yout) {   // ResidualWrapper: y = x + h'; x = this workgroup's units of segment 0 of the staged tile
                const float4 x4 = *reinterpret_cast<const float4*>(As + (j * M + row) * 16 + c4);
                hn.x += x4.x; hn.y += x4.y; hn.z += x4.z; hn.w += x4.w;
                ws_st4(ws_rsrc(ph.yout), ooff, hn);
                if (ph.yhist && b0 + row < B)   // (read by the deferred output projection after the launch: a plain store)
                    *reinterpret_cast<float4*>(ph.yhist + (size_t)(b0 + row) * ph.yld + j * 16 + c4) = hn;
            }
        }
        // (a plain yhist store drains with the others: one wait covers both)
        WS_STAMP(5)
        if (EPI != WS_CUDNN_RU) ws_publish_wave(cnt, WS_ARRIVALS / (unsigned)(NT >= 64 ? NT / 64 : 1));
        WS_STAMP(6)
    }
    if (EPI == WS_CUDNN_RU) __syncthreads();   // r / u are in LDS, the partial tiles may be overwritten (no hand-off here)
}

// CudnnCompatibleGRUCell (reference layers.py:560-577, model.py:226-227,257-259), second half of a cell on the tile its gates
// phase staged -- no wait, no staging, one hand-off per cell: c = tanh(x W_ci + b_ci + r * (h W_ch + b_ch)), h' = u h + (1 - u) c.
// Waves 0..3 multiply segment 0 (x, K0 columns) with W_ci, waves 4..7 segment 1 (h, 256 columns) with W_ch, four K slices each.
template <int M, int K0, int UB0, int ROFF, bool NEXT1 = false>
__device__ __forceinline__ void ws_phase_hx(const float (&w)[DEC_WS_NREG], const WsPhase& ph, float* lds, int j, int b0, int B,
                                            unsigned* cnt, float4 (&pre)[4]) {
    constexpr int RB = M / 16, PRE = WS_D * M / 4 / WS_THREADS;
    constexpr int CH0 = K0 / 64, CH1 = WS_D / 64;   // 16-deep chunks per wave: x part, h part
    static_assert(K0 % 64 == 0 && CH0 <= CH1, "phase shape");
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // (see ws_phase)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    float* As = lds + WS_OFF_AS;
    float* red = lds + WS_OFF_RED(M);
    float* h_loc = lds + WS_OFF_H(M) + ph.layer * (M * 16);
    const float* bias = lds + WS_OFF_BIAS(M) + ph.bias_slot * 32;
    const int tile = wave >> 2, slice = wave & 3;           // tile 0: x W_ci, tile 1: h W_ch
    const int kb = tile ? K0 + slice * (WS_D / 4) : slice * (K0 / 4);
    const int nch = tile ? CH1 : CH0;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CH1; ++c) {
        if (c < nch) {   // wave-uniform
            int rbs;
            const int off = ws_a_off<M, K0, UB0>(kb + 16 * c, r, q, &rbs);
            const float4 a0v = *reinterpret_cast<const float4*>(As + off);
            if (RB == 2) {
                const float4 a1v = *reinterpret_cast<const float4*>(As + off + rbs);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.x, w[ROFF + 4 * c + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v.x, w[ROFF + 4 * c + 0], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.y, w[ROFF + 4 * c + 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v.y, w[ROFF + 4 * c + 1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.z, w[ROFF + 4 * c + 2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v.z, w[ROFF + 4 * c + 2], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.w, w[ROFF + 4 * c + 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v.w, w[ROFF + 4 * c + 3], acc1, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.x, w[ROFF + 4 * c + 0], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.y, w[ROFF + 4 * c + 1], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.z, w[ROFF + 4 * c + 2], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v.w, w[ROFF + 4 * c + 3], acc0, 0, 0, 0);
            }
        }
    }
    const int rslot = slice * 2 + tile;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[((rslot * RB + 0) * 16 + q * 4 + i) * WS_RED_LD + r] = acc0[i];
        if (RB == 2) red[((rslot * RB + 1) * 16 + q * 4 + i) * WS_RED_LD + r] = acc1[i];
    }
    __syncthreads();
    if (NEXT1) {   // (see ws_phase)
        const __amdgpu_buffer_rsrc_t rn = ws_rsrc(ph.next1);
#pragma unroll
        for (int u = 0; u < PRE; ++u) pre[u] = ws_ld4(rn, (unsigned)(tid + WS_THREADS * u) * 16u);
    }
    if (tid < M * 4) {
        const int row = tid >> 2, c4 = (tid & 3) * 4;
        const int rb = row >> 4, rr = row & 15;
        float4 v[2];   // [0] = x W_ci + b_ci, [1] = h W_ch + b_ch
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            v[g] = *reinterpret_cast<const float4*>(bias + g * 16 + c4);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const float4 t4 = *reinterpret_cast<const float4*>(red + (((sl * 2 + g) * RB + rb) * 16 + rr) * WS_RED_LD + c4);
                v[g].x += t4.x; v[g].y += t4.y; v[g].z += t4.z; v[g].w += t4.w;
            }
        }
        const unsigned ooff = (unsigned)((j * M + row) * 16 + c4) * 4u;
        float* hl = h_loc + row * 16 + c4;
        const float4 h4 = *reinterpret_cast<const float4*>(hl);
        const float4 u4 = *reinterpret_cast<const float4*>(lds + WS_OFF_U(M) + row * 16 + c4);
        const float4 r4 = *reinterpret_cast<const float4*>(lds + WS_OFF_R(M) + row * 16 + c4);
        float4 hn;
        hn.x = u4.x * h4.x + (1.0f - u4.x) * tanhf_(v[0].x + r4.x * v[1].x);
        hn.y = u4.y * h4.y + (1.0f - u4.y) * tanhf_(v[0].y + r4.y * v[1].y);
        hn.z = u4.z * h4.z + (1.0f - u4.z) * tanhf_(v[0].z + r4.z * v[1].z);
        hn.w = u4.w * h4.w + (1.0f - u4.w) * tanhf_(v[0].w + r4.w * v[1].w);
        *reinterpret_cast<float4*>(hl) = hn;
        ws_st4(ws_rsrc(ph.out), ooff, hn);
        if (ph.yout) {   // ResidualWrapper: y = x + h'
            const float4 x4 = *reinterpret_cast<const float4*>(As + (j * M + row) * 16 + c4);
            hn.x += x4.x; hn.y += x4.y; hn.z += x4.z; hn.w += x4.w;
            ws_st4(ws_rsrc(ph.yout), ooff, hn);
            if (ph.yhist && b0 + row < B)
                *reinterpret_cast<float4*>(ph.yhist + (size_t)(b0 + row) * ph.yld + j * 16 + c4) = hn;
        }
        ws_publish_wave(cnt, WS_ARRIVALS / (unsigned)(M * 4 / 64));
    }
}

// Luong dot attention for this workgroup's rows of the cluster -- rows 2 j and 2 j + 1 for M = 32, row j for M = 16 (TF-1.8
// _luong_score / _compute_attention; dot form at reference attention.py:396-400): softmax over ALL Ts positions, context =
// alignments . memory.  FOUR waves per row in both forms (the same partial sums in the same order: a row's bits do not depend
// on M); with one row the other four waves only keep the barriers.
// LOCAL: LocalLuongAttention (reference tacotron/attention.py:32-342; decoder.hip / decoder_persistent.hip have the other two
// forms): only the window of 2D+1 positions around the step index (monotonic) or around the predicted centre
// p = Ts sigmoid(v_p . tanh(W_p h)) is scored; the reported alignments are zero outside it and, with `gaussian`, weighted as
// the reference writes it (attention.py:73-80); a predicted window that leaves the memory raises the error flag (the
// reference's padding arithmetic fails there, attention.py:288-304).
struct WsLocal {
    int d, gaussian, predictive, step;
    const float* wp; const float* vp;   // [256][256] (q @ W_p), [256]
    float* p_hist_t;                    // [B] predicted centres of this step
    int* err_flag;
};
template <int M, bool LOCAL>
__device__ __forceinline__ void ws_attention(const float* __restrict__ query, const float* __restrict__ keys,
                                             const float* __restrict__ values, float* ctx, float* align_t, int Ts, float* lds,
                                             int j, int b0, int B, unsigned* cnt, unsigned target, int* status, const WsLocal& lc) {
    constexpr int RPW = M / 16;            // rows per workgroup
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // (see ws_phase)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, hw = wave & 3, t256 = tid & 255;
    const bool active = half < RPW;        // wave-uniform
    float* qs = lds + WS_OFF_AS + half * WS_D;                        // [2][256]
    float* part = lds + WS_OFF_AS + 2 * WS_D + half * (4 * WS_D);     // [2][4][256]
    float* redm = lds + WS_OFF_RED(M) + half * 16;                    // [2][4] maxima, then sums
    float* invs = lds + WS_OFF_RED(M) + 32;                           // [2] 1 / sum
    const int Tsp = (Ts + 3) & ~3;
    float* sc = lds + WS_OFF_SC(M) + half * Tsp;
    int* ctrl = reinterpret_cast<int*>(lds + WS_OFF_CTRL(M));

    const int rl = RPW * j + (active ? half : 0);   // row inside the cluster
    const int row = b0 + rl;
    const bool row_ok = row < B;
    const int mr = row_ok ? row : B - 1;   // (a padding row of the last cluster attends over the last utterance's memory)

    // Scores: 16 lanes per key, 16 keys per pass of the row's 4 waves, WS_KB passes requested together.  The keys do
    // not depend on the query: the first WS_KB passes are requested BEFORE the wait for the cluster.
    const int sub = lane >> 4, l16 = lane & 15;
    float4 kpre[WS_KB][4];
    auto load_keys = [&](const float* kbase, int j0, int n_pos) {
#pragma unroll
        for (int p = 0; p < WS_KB; ++p) {
            const int jj = j0 + 16 * p + hw * 4 + sub;
            const float* kr = kbase + (size_t)(jj < n_pos ? jj : n_pos - 1) * WS_D;   // clamped, never branched on
#pragma unroll
            for (int i = 0; i < 4; ++i) kpre[p][i] = *reinterpret_cast<const float4*>(kr + (l16 + 16 * i) * 4);
        }
    };
    if (active && !LOCAL) load_keys(keys + (size_t)mr * Ts * WS_D, 0, Ts);   // (the window of the local form may depend on the query)
    WS_STAMP(0)

    ws_wait(cnt, target, status, ctrl);
    WS_STAMP(1)

    // the query: row rl of the attention GRU's new state (block format: unit u at ((u / 16) * M + row) * 16 + u % 16)
    if (active && t256 < 64)
        *reinterpret_cast<float4*>(qs + 4 * t256) =
            ws_ld4(ws_rsrc(query), (unsigned)(((t256 >> 2) * M + rl) * 16 + 4 * (t256 & 3)) * 4u);
    __syncthreads();

    // scored positions [w_lo, w_lo + w_n): the whole memory, or the local window
    int w_lo = 0, w_n = Ts;
    float pc = 0.f;   // window centre as the gaussian sees it
    if (LOCAL) {
        w_n = 2 * lc.d + 1;
        if (lc.predictive) {
            // (q W_p)[n] by thread n of the row's 256 (columns coalesced), then v_p . tanh(.) over them
            float v = 0.f;
            if (active) {
                float a0 = 0.f, a1 = 0.f;
                const float* wpn = lc.wp + t256;
                for (int k = 0; k < WS_D; k += 2) {
                    a0 = fmaf(qs[k], wpn[(size_t)k * WS_D], a0);
                    a1 = fmaf(qs[k + 1], wpn[(size_t)(k + 1) * WS_D], a1);
                }
                v = tanhf_(a0 + a1) * lc.vp[t256];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                if (lane == 0) redm[hw] = v;
            }
            __syncthreads();
            const float pp = (float)Ts * sigmoidf_((redm[0] + redm[1]) + (redm[2] + redm[3]));
            const int c = (int)floorf(pp);
            // a window that leaves the memory: the reference's padding arithmetic fails there (decoder.hip)
            if (active && t256 == 0 && row_ok) {
                lc.p_hist_t[row] = pp;
                if (c - lc.d < 0 || c + lc.d + 1 > Ts) *lc.err_flag = 1;
            }
            w_lo = min(max(c - lc.d, 0), Ts - w_n);
            pc = pp;
            __syncthreads();   // redm is reused below
        } else {
            int c = lc.step > lc.d ? lc.step : lc.d;
            const int hi = Ts - (lc.d + 1);
            c = c < hi ? c : hi;
            w_lo = c - lc.d;
            pc = (float)c;
        }
    }
    const float* kb = keys + ((size_t)mr * Ts + w_lo) * WS_D;

    if (active) {
        for (int j0 = 0; j0 < w_n; j0 += 16 * WS_KB) {
            if (LOCAL || j0 > 0) load_keys(kb, j0, w_n);
#pragma unroll
            for (int p = 0; p < WS_KB; ++p) {
                const int jj = j0 + 16 * p + hw * 4 + sub;
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 kv = kpre[p][i];
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
    }
    __syncthreads();
    WS_STAMP(2)

    float m = -INFINITY;
    if (active) {
        for (int jj = t256; jj < w_n; jj += 256) m = fmaxf(m, sc[jj]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) redm[hw] = m;
    }
    __syncthreads();
    float sum = 0.f;
    if (active) {
        m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
        for (int jj = t256; jj < w_n; jj += 256) {
            const float e = __expf(sc[jj] - m);
            sc[jj] = e;
            sum += e;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    }
    __syncthreads();   // everyone has read the maxima
    if (active && lane == 0) redm[hw] = sum;
    __syncthreads();
    float inv = 0.f;
    if (active) {
        sum = (redm[0] + redm[1]) + (redm[2] + redm[3]);
        inv = 1.0f / sum;
        if (hw == 0 && lane == 0) invs[half] = inv;
    }

    // context: wave hw takes positions hw, hw + 4, ...; a lane owns 4 consecutive depth elements (1 KB rows, coalesced);
    // WS_VB rows requested together
    WS_STAMP(3)
    if (active) {
        const float* vb = values + ((size_t)mr * Ts + w_lo) * WS_D + 4 * lane;
        float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j0 = hw; j0 < w_n; j0 += 4 * WS_VB) {
            float4 vv[WS_VB];
#pragma unroll
            for (int p = 0; p < WS_VB; ++p) {
                const int jj = j0 + 4 * p;
                vv[p] = *reinterpret_cast<const float4*>(vb + (size_t)(jj < w_n ? jj : w_n - 1) * WS_D);
            }
#pragma unroll
            for (int p = 0; p < WS_VB; ++p) {
                const int jj = j0 + 4 * p;
                const float e = jj < w_n ? sc[jj] : 0.f;
                c0.x = fmaf(e, vv[p].x, c0.x); c0.y = fmaf(e, vv[p].y, c0.y); c0.z = fmaf(e, vv[p].z, c0.z); c0.w = fmaf(e, vv[p].w, c0.w);
            }
        }
        *reinterpret_cast<float4*>(part + hw * WS_D + 4 * lane) = c0;
    }
    __syncthreads();
    WS_STAMP(4)
    // wave 0 hands the rows over: a lane's 16 bytes are (block j', row RPW j + hf, units 4c..4c+3) -- M = 32: eight lanes one
    // 128-byte line of the context buffer (block format: two rows x 16 units), two store instructions; M = 16: one
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int jb = RPW == 2 ? 8 * i + (lane >> 3) : lane >> 2, hf = RPW == 2 ? (lane >> 2) & 1 : 0, c = lane & 3;
            const float* pp = lds + WS_OFF_AS + 2 * WS_D + hf * (4 * WS_D) + 16 * jb + 4 * c;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) {
                const float4 t4 = *reinterpret_cast<const float4*>(pp + wv * WS_D);
                a.x += t4.x; a.y += t4.y; a.z += t4.z; a.w += t4.w;
            }
            const float iv = invs[hf];
            a.x *= iv; a.y *= iv; a.z *= iv; a.w *= iv;
            ws_st4(ws_rsrc(ctx), (unsigned)((jb * M + RPW * j + hf) * 16 + 4 * c) * 4u, a);
        }
        WS_STAMP(5)
        ws_publish_wave(cnt, WS_ARRIVALS);
        WS_STAMP(6)
    }
    if (active && align_t && row_ok) {
        if (!LOCAL) {
            for (int k = t256; k < Ts; k += 256) align_t[(size_t)row * Ts + k] = sc[k] * inv;
        } else {
            // the reference pads the window back to the memory length (attention.py:85-92) and, with `gaussian`, weights it by
            // exp(-(j - p)^2 / 2 * (D/2)^2) as written at attention.py:73-80 (the context uses the plain softmax)
            const float gk = 0.5f * (0.5f * lc.d) * (0.5f * lc.d);
            for (int k = t256; k < Ts; k += 256) {
                const int jw = k - w_lo;
                float a = 0.f;
                if (jw >= 0 && jw < w_n) {
                    a = sc[jw] * inv;
                    if (lc.gaussian) {
                        const float dist = (float)k - pc;
                        a *= __expf(-(dist * dist) * gk);
                    }
                }
                align_t[(size_t)row * Ts + k] = a;
            }
        }
    }
}

template <bool CUDNN, int M, bool LOCAL = false>
__global__ __launch_bounds__(WS_THREADS) void dec_ws_kernel(WsParams p) {
    constexpr int PRE = WS_D * M / 4 / WS_THREADS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cluster = blockIdx.x / WS_W, j = blockIdx.x - cluster * WS_W;
    const int b0 = cluster * M;
    unsigned* cnt = p.counters + 64 * cluster;
    int* ctrl = reinterpret_cast<int*>(lds + WS_OFF_CTRL(M));

    // ---- this wave's weights: up to 176 registers per lane, once, from the register-order image (decoder_ws_pack)
    float w[DEC_WS_NREG];
    {
        const float4* img = reinterpret_cast<const float4*>(p.wimg) + (size_t)(j * WS_NW + wave) * (DEC_WS_NREG / 4) * 64 + lane;
#pragma unroll
        for (int i = 0; i < DEC_WS_NREG / 4; ++i) {
            const float4 v = img[i * 64];
            w[4 * i + 0] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
        // (opaque from here on: the compiler must keep them in registers, not re-load them inside the loop)
#pragma unroll
        for (int i = 0; i < DEC_WS_NREG; ++i) asm volatile("" : "+v"(w[i]));
    }
    for (int i = tid; i < 4 * M * 16; i += WS_THREADS) lds[WS_OFF_H(M) + i] = 0.f;   // zero_state (h of three cells, u)
    for (int i = tid; i < DEC_WS_BIAS_SLOTS * 32; i += WS_THREADS) lds[WS_OFF_BIAS(M) + i] = p.bimg[j * (DEC_WS_BIAS_SLOTS * 32) + i];
    if (tid == 0) {
        ctrl[0] = 0;
        // all workgroups resident: the CUs the call pipeline held for this stream are no longer needed
        const unsigned n = __hip_atomic_fetch_add(p.resident, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n + 1 == gridDim.x && p.hold_flag) __hip_atomic_store(p.hold_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();

    float* st = p.state + (size_t)cluster * WS_STATE_FLOATS(M);
    float* att = st, *ycur = st + 7 * WS_BUF(M);
    float* rs = p.rest + (size_t)cluster * WS_REST_FLOATS(M);
    float* p1 = rs, *rh = rs + WS_BUF(M), *ctx = rs + 2 * WS_BUF(M), *y0 = rs + 3 * WS_BUF(M), *p2 = rs + 4 * WS_BUF(M);
    const int yld = p.n_steps * WS_D;
    unsigned g = 0;   // hand-offs completed by the cluster
    const unsigned per = WS_ARRIVALS * WS_W;

    // segment 1 of the first phase (the attention of "step -1": zeros), as every later step gets it from the phase before
    float4 pre[4];
    {
        const __amdgpu_buffer_rsrc_t rn = ws_rsrc(att);
#pragma unroll
        for (int u = 0; u < PRE; ++u) pre[u] = ws_ld4(rn, (unsigned)(tid + WS_THREADS * u) * 16u);
    }
    for (int t = 0; t < p.n_steps; ++t) {
        // cell states by step parity: step t reads [t & 1] and writes [(t + 1) & 1]
        const int po = t & 1, pn = po ^ 1;
        float* h_att_o = st + (1 + po) * WS_BUF(M), *h_att = st + (1 + pn) * WS_BUF(M);
        float* h_d1_o = st + (3 + po) * WS_BUF(M), *h_d1 = st + (3 + pn) * WS_BUF(M);
        float* h_d2_o = st + (5 + po) * WS_BUF(M), *h_d2 = st + (5 + pn) * WS_BUF(M);
        WsPhase ph;
        ph.yout = nullptr; ph.yhist = nullptr; ph.yld = yld; ph.layer = 0; ph.delay = p.dbg_delay; ph.next1 = nullptr;
        // PrenetWrapper on concat([x_t, attention_{t-1}]) (wrappers.py:122-124).  x_t = (y_{t-1} W_o + b_o)[-n_mels:] is folded
        // into the pre-net matrix (decoder.hip); x_0 = GO frame = zeros (helpers.py:108): y and attention are zero at step 0,
        // so only the bias differs there (the un-folded one)
        ph.a0 = ycur; ph.a1 = att; ph.out = p1; ph.bias_slot = t == 0 ? 1 : 0; ph.target = per * g++;
        WS_TL_PHASE(t, 0)
        ws_phase<M, WS_D, 16, WS_D, 1, 16, WS_ACT, ACT_RELU, WS_R0, false, 0, true, false>(w, ph, lds, j, b0, p.B, cnt, p.status, pre);
        ph.a0 = p1; ph.a1 = nullptr; ph.out = p2; ph.bias_slot = 2; ph.target = per * g++; ph.next1 = h_att_o;
        WS_TL_PHASE(t, 1)
        ws_phase<M, WS_D, 16, 0, 1, 8, WS_ACT, ACT_RELU, WS_R1, false, 0, false, true>(w, ph, lds, j, b0, p.B, cnt, p.status, pre);
        // attention GRU (model.py:226-229) on [p2 ; h_att]; the new state is the attention query
        if (CUDNN) {   // one hand-off: r, u, then x W_ci and h W_ch on the same staged tile
            ph.a0 = p2; ph.a1 = h_att_o; ph.out = nullptr; ph.bias_slot = 3; ph.target = per * g;
            WS_TL_PHASE(t, 2)
            ws_phase<M, WS_P2, 8, WS_D, 2, 16, WS_CUDNN_RU, ACT_NONE, WS_R2, false, 0, true, false>(w, ph, lds, j, b0, p.B, cnt, p.status, pre);
            ph.out = h_att; ph.bias_slot = 4; ++g;
            WS_TL_PHASE(t, 3)
            ws_phase_hx<M, WS_P2, 8, WS_R3, false>(w, ph, lds, j, b0, p.B, cnt, pre);
        } else {       // TF GRUCell: gates on [p2 ; h_att], a hop, candidate on [p2 ; r*h_att]
            ph.a0 = p2; ph.a1 = h_att_o; ph.out = rh; ph.bias_slot = 3; ph.target = per * g++;
            WS_TL_PHASE(t, 2)
            ws_phase<M, WS_P2, 8, WS_D, 2, 16, WS_GATES, ACT_NONE, WS_R2, false, 0, true, false>(w, ph, lds, j, b0, p.B, cnt, p.status, pre);
            ph.a0 = p2; ph.a1 = rh; ph.out = h_att; ph.bias_slot = 4; ph.target = per * g++;
            WS_TL_PHASE(t, 3)
            ws_phase<M, WS_P2, 8, WS_D, 1, 16, WS_CAND, ACT_NONE, WS_R3, true, 4, false, false>(w, ph, lds, j, b0, p.B, cnt, p.status, pre);
        }
        WS_TL_PHASE(t, 4)
        {
            WsLocal lc;
            lc.d = p.local_d; lc.gaussian = p.local_gaussian; lc.predictive = p.local_predictive; lc.step = t;
            lc.wp = p.local_wp; lc.vp = p.local_vp; lc.p_hist_t = p.p_hist ? p.p_hist + (size_t)t * p.B : nullptr;
            lc.err_flag = p.err_flag;
            ws_attention<M, LOCAL>(h_att, p.keys, p.memory, ctx, p.align ? p.align + (size_t)t * p.B * p.Ts : nullptr, p.Ts, lds, j, b0,
                                   p.B, cnt, per * g++, p.status, lc);
        }
        // attention_layer(concat([cell_output, context])), no bias
        ph.a0 = h_att; ph.a1 = ctx; ph.out = att; ph.bias_slot = 5; ph.target = per * g++; ph.next1 = h_d1_o;
        WS_TL_PHASE(t, 5)
        ws_phase<M, WS_D, 16, WS_D, 1, 16, WS_ACT, ACT_NONE, WS_R5, false, 0, false, true>(w, ph, lds, j, b0, p.B, cnt, p.status, pre);
        // two ResidualWrapper(GRU cell) layers (model.py:254-269); the top one writes the y history
        ph.layer = 1;
        if (CUDNN) {
            ph.a0 = att; ph.a1 = h_d1_o; ph.out = nullptr; ph.bias_slot = 6; ph.target = per * g;
            WS_TL_PHASE(t, 6)
            ws_phase<M, WS_D, 16, WS_D, 2, 16, WS_CUDNN_RU, ACT_NONE, WS_R6, false, 0, true, false>(w, ph, lds, j, b0, p.B, cnt, p.status, pre);
            ph.out = h_d1; ph.yout = y0; ph.bias_slot = 7; ++g; ph.next1 = h_d2_o;
            WS_TL_PHASE(t, 7)
            ws_phase_hx<M, WS_D, 16, WS_R7, true>(w, ph, lds, j, b0, p.B, cnt, pre);
            ph.layer = 2; ph.yout = nullptr;
            ph.a0 = y0; ph.a1 = h_d2_o; ph.out = nullptr; ph.bias_slot = 8; ph.target = per * g;
            WS_TL_PHASE(t, 8)
            ws_phase<M, WS_D, 16, WS_D, 2, 16, WS_CUDNN_RU, ACT_NONE, WS_R8, false, 0, true, false>(w, ph, lds, j, b0, p.B, cnt, p.status, pre);
            ph.out = h_d2; ph.yout = ycur; ph.yhist = p.yhist + (size_t)t * WS_D; ph.bias_slot = 9; ++g; ph.next1 = att;
            WS_TL_PHASE(t, 9)
            ws_phase_hx<M, WS_D, 16, WS_R9, true>(w, ph, lds, j, b0, p.B, cnt, pre);
        } else {
            ph.a0 = att; ph.a1 = h_d1_o; ph.out = rh; ph.bias_slot = 6; ph.target = per * g++;
            WS_TL_PHASE(t, 6)
            ws_phase<M, WS_D, 16, WS_D, 2, 16, WS_GATES, ACT_NONE, WS_R6, false, 0, true, false>(w, ph, lds, j, b0, p.B, cnt, p.status, pre);
            ph.a0 = att; ph.a1 = rh; ph.out = h_d1; ph.yout = y0; ph.bias_slot = 7; ph.target = per * g++; ph.next1 = h_d2_o;
            WS_TL_PHASE(t, 7)
            ws_phase<M, WS_D, 16, WS_D, 1, 16, WS_CAND, ACT_NONE, WS_R7, true, 4, false, true>(w, ph, lds, j, b0, p.B, cnt, p.status, pre);
            ph.layer = 2; ph.yout = nullptr;
            ph.a0 = y0; ph.a1 = h_d2_o; ph.out = rh; ph.bias_slot = 8; ph.target = per * g++;
            WS_TL_PHASE(t, 8)
            ws_phase<M, WS_D, 16, WS_D, 2, 16, WS_GATES, ACT_NONE, WS_R8, false, 0, true, false>(w, ph, lds, j, b0, p.B, cnt, p.status, pre);
            ph.a0 = y0; ph.a1 = rh; ph.out = h_d2; ph.yout = ycur; ph.yhist = p.yhist + (size_t)t * WS_D; ph.bias_slot = 9;
            ph.target = per * g++; ph.next1 = att;
            WS_TL_PHASE(t, 9)
            ws_phase<M, WS_D, 16, WS_D, 1, 16, WS_CAND, ACT_NONE, WS_R9, true, 4, false, true>(w, ph, lds, j, b0, p.B, cnt, p.status, pre);
        }
    }
}

bool decoder_ws_supports(const DecoderWeights& w, int cudnn, int B, int Ts) {
    (void)cudnn;   // both GRU formulations (the register image is packed for the handle's)
    if (w.local_d > 0 && Ts < 2 * w.local_d + 1) return false;
    return w.n_layers == 2 && w.att_units == WS_D && w.dec_units == WS_D && w.mem_units == WS_D &&
           w.prenet1_units == WS_D && w.prenet2_units == WS_P2 && w.n_mels <= WS_D && w.ws_wimg && w.ws_bimg && B >= 1 && Ts >= 1 &&
           ws_lds_bytes(Ts, 32) <= 160 * 1024 - 64 && (size_t)B * Ts * WS_D * 4 < 0xFFFFFFF0ull;
}

// rows: utterances per cluster, 32 or 16 (see the LDS map)
static int ws_rows(int rows) { return rows == 16 ? 16 : 32; }
int decoder_ws_workgroups(int B, int rows) { const int M = ws_rows(rows); return WS_W * ((B + M - 1) / M); }
size_t decoder_ws_scratch_floats(int B, int rows) {
    const int M = ws_rows(rows);
    return (size_t)((B + M - 1) / M) * (WS_STATE_FLOATS(M) + WS_REST_FLOATS(M));
}
int decoder_ws_clusters(int B, int rows) { const int M = ws_rows(rows); return (B + M - 1) / M; }

hipError_t decoder_ws_configure() {
    const void* fns[8] = {reinterpret_cast<const void*>(&dec_ws_kernel<false, 32>), reinterpret_cast<const void*>(&dec_ws_kernel<true, 32>),
                          reinterpret_cast<const void*>(&dec_ws_kernel<false, 16>), reinterpret_cast<const void*>(&dec_ws_kernel<true, 16>),
                          reinterpret_cast<const void*>(&dec_ws_kernel<false, 32, true>), reinterpret_cast<const void*>(&dec_ws_kernel<true, 32, true>),
                          reinterpret_cast<const void*>(&dec_ws_kernel<false, 16, true>), reinterpret_cast<const void*>(&dec_ws_kernel<true, 16, true>)};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// The register-order image of the decoder's weights (host; called by tts_finalize_weights).  Wt arrays are the packed
// [N][K] matrices of decoder.hip (k contiguous, K = the concatenated A operand).  wimg [16 workgroups][8 waves]
// [DEC_WS_NREG / 4 float4][64 lanes][4]: lane (n = lane & 15, q = lane >> 4) of wave w holds, in register ROFF + 4 c + e
// of a phase with TILES column tiles, W[gate * 256 + j * UBO + n][slice * KW + 16 c + 4 q + e] with gate = w % TILES,
// slice = (w / TILES + SROT) % (8 / TILES) -- exactly what ws_phase multiplies the staged element (row, that k) with.  bimg [16][slots][32].
void decoder_ws_pack(const DecWsHostWeights& hw, float* wimg, float* bimg) {
    struct Ph { const float* Wt; int K, tiles, ubo, roff, srot, k0; };
    // srot: ws_phase's SROT (the candidate phases); k0 > 0: a CudnnCompatibleGRUCell's second half (ws_phase_hx) on the
    // [4U][K] block r | u | hh | xi: waves 0..3 hold W_ci (rows 3U..) over the k0 input columns, waves 4..7 W_ch (rows 2U..)
    // over the 256 state columns, four K slices each
    const Ph gru_cell[9] = {
        {hw.w1f, 512, 1, 16, WS_R0, 0, 0}, {hw.w2, 256, 1, 8, WS_R1, 0, 0}, {hw.ag_w, 384, 2, 16, WS_R2, 0, 0}, {hw.ac_w, 384, 1, 16, WS_R3, 4, 0},
        {hw.al_w, 512, 1, 16, WS_R5, 0, 0}, {hw.g_gw[0], 512, 2, 16, WS_R6, 0, 0}, {hw.g_cw[0], 512, 1, 16, WS_R7, 4, 0},
        {hw.g_gw[1], 512, 2, 16, WS_R8, 0, 0}, {hw.g_cw[1], 512, 1, 16, WS_R9, 4, 0}};
    const Ph cudnn_cell[9] = {
        {hw.w1f, 512, 1, 16, WS_R0, 0, 0}, {hw.w2, 256, 1, 8, WS_R1, 0, 0}, {hw.ag_w, 384, 2, 16, WS_R2, 0, 0}, {hw.ag_w, 384, 2, 16, WS_R3, 0, 128},
        {hw.al_w, 512, 1, 16, WS_R5, 0, 0}, {hw.g_gw[0], 512, 2, 16, WS_R6, 0, 0}, {hw.g_gw[0], 512, 2, 16, WS_R7, 0, 256},
        {hw.g_gw[1], 512, 2, 16, WS_R8, 0, 0}, {hw.g_gw[1], 512, 2, 16, WS_R9, 0, 256}};
    const Ph* phs = hw.cudnn ? cudnn_cell : gru_cell;
    std::memset(wimg, 0, sizeof(float) * decoder_ws_wimg_floats());
    for (int j = 0; j < WS_W; ++j)
        for (int wv = 0; wv < WS_NW; ++wv) {
            float* base = wimg + (size_t)(j * WS_NW + wv) * DEC_WS_NREG * 64;
            for (int pi = 0; pi < 9; ++pi) {
                const Ph& ph = phs[pi];
                int ch, kb, row0;
                if (ph.k0 > 0) {
                    const int tile = wv >> 2, slice = wv & 3;
                    ch = tile ? WS_D / 64 : ph.k0 / 64;
                    kb = tile ? ph.k0 + slice * (WS_D / 4) : slice * (ph.k0 / 4);
                    row0 = (tile ? 2 : 3) * WS_D + j * 16;
                } else {
                    const int ksl = WS_NW / ph.tiles, kw = ph.K / ksl;
                    const int gate = wv % ph.tiles, slice = (wv / ph.tiles + ph.srot) % ksl;
                    ch = kw / 16;
                    kb = slice * kw;
                    row0 = gate * WS_D + j * ph.ubo;
                }
                for (int c = 0; c < ch; ++c)
                    for (int e = 0; e < 4; ++e) {
                        const int reg = ph.roff + 4 * c + e;
                        for (int lane = 0; lane < 64; ++lane) {
                            const int n = lane & 15, q = lane >> 4;
                            const int k = kb + 16 * c + 4 * q + e;
                            const float v = n < ph.ubo ? ph.Wt[(size_t)(row0 + n) * ph.K + k] : 0.f;
                            base[((size_t)(reg >> 2) * 64 + lane) * 4 + (reg & 3)] = v;   // float4 i = reg / 4 of the lane
                        }
                    }
            }
        }
    std::memset(bimg, 0, sizeof(float) * decoder_ws_bimg_floats());
    for (int j = 0; j < WS_W; ++j) {
        float* b = bimg + (size_t)j * DEC_WS_BIAS_SLOTS * 32;
        for (int n = 0; n < 16; ++n) {
            b[0 * 32 + n] = hw.b1f[j * 16 + n];
            b[1 * 32 + n] = hw.b1[j * 16 + n];
            if (n < 8) b[2 * 32 + n] = hw.b2[j * 8 + n];
            // a cell's gates slot: [r | u]; its second slot: the candidate's bias (GRUCell) or [b_ci | b_ch] (cudnn: rows 3U.. | 2U..)
            const float* gb[3] = {hw.ag_b, hw.g_gb[0], hw.g_gb[1]};
            const float* cb[3] = {hw.ac_b, hw.g_cb[0], hw.g_cb[1]};
            const int slot[3] = {3, 6, 8};
            for (int l = 0; l < 3; ++l) {
                b[slot[l] * 32 + n] = gb[l][j * 16 + n];
                b[slot[l] * 32 + 16 + n] = gb[l][WS_D + j * 16 + n];
                if (hw.cudnn) {
                    b[(slot[l] + 1) * 32 + n] = gb[l][3 * WS_D + j * 16 + n];
                    b[(slot[l] + 1) * 32 + 16 + n] = gb[l][2 * WS_D + j * 16 + n];
                } else {
                    b[(slot[l] + 1) * 32 + n] = cb[l][j * 16 + n];
                }
            }
        }
    }
}
size_t decoder_ws_wimg_floats() { return (size_t)WS_W * WS_NW * DEC_WS_NREG * 64; }
size_t decoder_ws_bimg_floats() { return (size_t)WS_W * DEC_WS_BIAS_SLOTS * 32; }

// Capturable: two memsets and one launch.  `scratch`: decoder_ws_scratch_floats(B, rows) floats (the state blocks of all clusters
// first: zeroed here); `sync` = 64 unsigned per cluster (of sync_clusters >= the launch's) + 1 (resident count) + 1 (sticky status
// word, see decoder_persistent.hip).
// rows: utterances per cluster (32: 16 compute units per 32 utterances; 16: per 16 -- same bits, see the LDS map)
hipError_t decoder_ws_enqueue(hipStream_t s, const DecoderWeights& w, float* scratch, float* yhist, const float* memory,
                              const float* keys, int B, int Ts, int n_steps, float* align, unsigned* sync, int* hold_flag,
                              int cudnn, int dbg_delay, int rows, int sync_clusters, float* p_hist, int* err_flag) {
    const int M = ws_rows(rows);
    const int clusters = (B + M - 1) / M;
    // (the resident count and the sticky status word sit behind the counters of `sync_clusters` clusters: one place for
    //  both forms of one batch size, so that a status raised by a launch of one form is found after a launch of the other)
    if (sync_clusters < clusters) sync_clusters = clusters;
    const size_t state_floats = M == 16 ? WS_STATE_FLOATS(16) : WS_STATE_FLOATS(32);
    hipError_t e;
    if ((e = hipMemsetAsync(scratch, 0, (size_t)clusters * state_floats * sizeof(float), s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(sync, 0, ((size_t)64 * sync_clusters + 1) * sizeof(unsigned), s)) != hipSuccess) return e;
    WsParams p;
    p.wimg = w.ws_wimg; p.bimg = w.ws_bimg;
    p.memory = memory; p.keys = keys;
    p.state = scratch; p.rest = scratch + (size_t)clusters * state_floats;
    p.yhist = yhist; p.align = align;
    p.counters = sync; p.resident = sync + 64 * sync_clusters; p.status = reinterpret_cast<int*>(sync + 64 * sync_clusters + 1);
    p.hold_flag = hold_flag;
    p.B = B; p.Ts = Ts; p.n_steps = n_steps; p.dbg_delay = dbg_delay; p.cudnn = cudnn;
    p.local_d = w.local_d; p.local_gaussian = w.local_gaussian; p.local_predictive = w.local_d > 0 && w.local_predictive;
    p.local_wp = w.local_wp; p.local_vp = w.local_vp; p.p_hist = p_hist; p.err_flag = err_flag;
    if (p.local_predictive) {
        if (!p_hist || !err_flag) return hipErrorInvalidValue;
        if ((e = hipMemsetAsync(err_flag, 0, sizeof(int), s)) != hipSuccess) return e;
    }
    const dim3 grid(WS_W * clusters), block(WS_THREADS);
    const size_t lds = ws_lds_bytes(Ts, M);
#define WS_LAUNCH(C, MM, L) hipLaunchKernelGGL((dec_ws_kernel<C, MM, L>), grid, block, lds, s, p)
    if (w.local_d > 0) {
        if (M == 16) { if (cudnn) WS_LAUNCH(true, 16, true); else WS_LAUNCH(false, 16, true); }
        else { if (cudnn) WS_LAUNCH(true, 32, true); else WS_LAUNCH(false, 32, true); }
    } else {
        if (M == 16) { if (cudnn) WS_LAUNCH(true, 16, false); else WS_LAUNCH(false, 16, false); }
        else { if (cudnn) WS_LAUNCH(true, 32, false); else WS_LAUNCH(false, 32, false); }
    }
#undef WS_LAUNCH
#ifdef WS_TIMELINE
    {
        (void)hipStreamSynchronize(s);
        unsigned long long hst[80];
        (void)hipMemcpyFromSymbol(hst, HIP_SYMBOL(ws_dbg), sizeof(hst));
        for (int k = 0; k < 10; ++k) {
            fprintf(stderr, "phase %d:", k);
            for (int i = 0; i < 7; ++i) fprintf(stderr, " [%d]%.2f", i, (double)(hst[k * 8 + i] - hst[0]) / 100.0);
            fprintf(stderr, "\n");
        }
    }
#endif
    return hipGetLastError();
}

}  // namespace tts
