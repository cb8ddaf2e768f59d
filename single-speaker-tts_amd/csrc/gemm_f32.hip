// fp32 MFMA GEMM with an implicit-im2col A loader and fused epilogues (gfx950).
//
// Covers every dense contraction of the two CBHG stacks: reference tacotron/layers.py
//   wrapped_dense  :96-111   (pre-net, lifter, highway H|T, final Dense; GRU input projections)
//   conv1d 'SAME'  :361-367, :432-437 (conv bank k=1..K and the two k=3 projections)
//   batch_norm     :380-383, :440-443 (folded to a per-channel affine, applied AFTER the relu)
//   max_pooling1d  :518-521  (fused into the A loader of the first projection: pool=1)
//   highway        :241-258  (H|T share one GEMM; gate mix in the epilogue)
//
// Tile 128x128x32, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 blocks of 32x32 accumulators (f32).
// Operands are staged global -> registers -> LDS with the next tile's global loads in flight during the MFMAs.
//
// Round 4: the products run on the bf16 matrix pipe at f32 accuracy.  Every f32 operand is split EXACTLY into three
// bf16 terms where it is staged (x = hi + mid + lo: truncate to the top 16 bits, subtract, twice; 8 + 8 + 8 significand
// bits, the subtractions are exact), and a 16-deep step of a block is six v_mfma_f32_32x32x16_bf16 into the same f32
// accumulator: hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi.  The three products left out: the splits TRUNCATE, so
// |mid| < 2^-7 |x| and |lo| < 2^-15 |x|, and the two dropped mid*lo terms are bounded by 2^-21 |a||b| in the worst case
// (lo*lo by 2^-30) -- four f32 roundings of one product, not one; the MEASURED error of a whole GEMM against the float64
// oracle is that of the f32-input v_mfma_f32_32x32x2_f32 form the kernel used until round 3 (2.5e-7 rel-L2,
// tests/test_gpu_gemm.py: the truncation errors are one-sided but far below the accumulation's own rounding).  Six bf16 MFMAs of
// 8 passes replace eight f32 MFMAs of 16 passes per 16 k: 2.67x less matrix-pipe time; the splits are ~5.5 VALU
// instructions per element on the otherwise idle vector pipe.  LDS image per operand: [split][row][4 chunks of 8 bf16]
// with the chunk index XOR-swizzled by (row >> 2) & 3 (ds_read_b128 of 16 rows at one chunk hit 16 different bank
// quads without padding: 48 KB per workgroup, three workgroups per CU).
// Non-finite and tiny operands (round 5, tests/test_gpu_round5.py): the split of +-Inf is (Inf, NaN, NaN) -- Inf - Inf in
// the first subtraction -- and a NaN splits into NaNs, so every output that depends on a non-finite operand is NON-FINITE
// (NaN where the f32-input MFMA would give +-Inf for a lone infinity), and every other output is untouched.  Below ~2^-110
// the lower terms are bf16 denormals (and below 2^-126 all three are): whatever the matrix pipe does with them, the result
// differs from the exact one by less than K 2^-126 max|w| in absolute terms -- measured 2.5e-39 on 1e-39 ... 1e-36 operands.
// From 1e-30 to 1e30 the error against float64 is the same 2.5e-7 rel-L2 as at unit scale.
#include "tts_common.h"
#include <cstring>
#include <type_traits>

namespace tts {

// Register cap: amdgpu_num_vgpr(80) makes hipcc allocate 160 unified registers (accumulators included, four values spilled)
// instead of 104 + 64 accumulation registers: still three waves per SIMD, the encoder / post-net 2.5 % faster alone (2.83 ->
// 2.75 ms) and 32 registers per lane left on a CU that holds three GEMM workgroups (small kernels of the other stream fit).
#ifndef GEMM_NUM_VGPR
#define GEMM_NUM_VGPR 80
#endif
#ifndef GEMM_NUM_VGPR_POOL
#define GEMM_NUM_VGPR_POOL 128
#endif
#define BM 128
#define BN 128
#define BK 32
#define LDS_LD (BK + 4)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// x = hi + mid + lo exactly, each a bf16 (the top 16 bits of an f32): bits of x, of x - hi, of x - hi - mid
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x);
    const float r1 = x - __uint_as_float(h & 0xFFFF0000u);
    m = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = __float_as_uint(r2);
}
// the top halves of two words as one: (hi16(b) << 16) | hi16(a)
__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
// four consecutive k of one row -> 8 bytes per split at (split, row, k): byte offset inside an operand's image
__device__ __forceinline__ void store_split4(unsigned char* img, int row, int kq, float4 v) {
    unsigned h[4], m[4], l[4];
    split3(v.x, h[0], m[0], l[0]);
    split3(v.y, h[1], m[1], l[1]);
    split3(v.z, h[2], m[2], l[2]);
    split3(v.w, h[3], m[3], l[3]);
    const int off = (row * 4 + ((kq >> 1) ^ ((row >> 2) & 3))) * 16 + (kq & 1) * 8;
    *reinterpret_cast<uint2*>(img + off) = make_uint2(pack_hi(h[0], h[1]), pack_hi(h[2], h[3]));
    *reinterpret_cast<uint2*>(img + BM * 64 + off) = make_uint2(pack_hi(m[0], m[1]), pack_hi(m[2], m[3]));
    *reinterpret_cast<uint2*>(img + 2 * BM * 64 + off) = make_uint2(pack_hi(l[0], l[1]), pack_hi(l[2], l[3]));
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

// DENORM: instantiation with the de-normalising second output (final Dense layer only; keeps the
// powf out of the register allocation of every other GEMM)
// POOL: instantiation whose A loader takes max(x[t], x[t+1]) (the two k = 3 projections that follow a max-pool): kept
// apart so that every other GEMM carries neither its second load nor its registers.
// PRE and PS are round 5's two measured-and-not-faster variants (HISTORY.md part C, profiles/r05_experiment_gemm_presplit.txt).
// They stay in the body's source as template parameters, but the shipped library instantiates neither: only a tools build
// with -DGEMM_EXPERIMENTS (tools/build_variant.sh) compiles their kernels and accepts the options "gemm_presplit" / "gemm_ps".
// PRE: the weights come PRE-SPLIT (round 5): `g.Wimg` holds, per 128-row block of N and per k tile in the order the k loop
// visits them, the 24 KB LDS image of the B tile itself -- [split][row][4 chunks, swizzled][8 bf16], made once per weight by
// gemm_pack_weights_kernel with the same split3 -- so staging the B tile is six 16-byte loads and six linear ds_write_b128
// per thread and no VALU work: the constants are no longer re-split in every k tile of every workgroup of every call
// (round 4: half of the kernel's 85.5 M VALU instructions per launch, profiles/r04_gemm_mfma_counters.txt).
// PS (round 5): PRODUCER / CONSUMER split inside the workgroup.  512 threads: waves 0..3 only multiply (the 2 x 2 wave tile of
// the 256-thread form), waves 4..7 only stage (global loads, the three-way split, LDS stores) into the OTHER of two LDS image
// pairs; one barrier per k tile.  In the 256-thread form every wave splits, then every wave multiplies, and the three
// workgroups of a compute unit fall into step with each other (the matrix pipe serialises their MFMA phases, so they reach
// their VALU phases together): matrix pipe and vector pipe take turns instead of running side by side -- 41 % matrix-pipe
// busy with neither pipe, nor LDS latency, nor occupancy as the limit (profiles/r05_experiment_gemm_presplit.txt).  Here every
// SIMD holds one multiplying and one staging wave for the whole k loop.
template <bool DENORM, bool POOL, bool PRE, bool PS = false>
__device__ __forceinline__ void gemm_body(const GemmBatch& batch) {
    const GemmGroup& g = batch.g[blockIdx.z];
    const int M = g.M, N = g.N, K = g.K;
    // Workgroup -> tile map.  Workgroups are dealt round-robin over the 8 XCDs in launch order (x fastest),
    // each XCD has its own L2, and all N-blocks of one M-block read the same rows of A: so the N-blocks of an
    // M-block get consecutive slots of ONE XCD (launch ids lin, lin + 8, ...), and the activation rows are
    // fetched from HBM once instead of once per N-block.
    const int nyb = gridDim.y;
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, seq = lin >> 3;
    const int m_blk = (seq / nyb) * 8 + xcd;
    const int m0 = m_blk * BM;
    const int n0 = (seq % nyb) * BN;
    if (n0 >= N || m0 >= M) return;

    // [split][row][32 bf16], chunks swizzled; PS: two such pairs in dynamic LDS (96 KB), As / Bs = the pair being read,
    // As_w / Bs_w = the pair being written
    __shared__ __attribute__((aligned(16))) unsigned char As_static[PS ? 16 : 3 * BM * 64];
    __shared__ __attribute__((aligned(16))) unsigned char Bs_static[PS ? 16 : 3 * BN * 64];
    extern __shared__ __attribute__((aligned(16))) unsigned char ps_smem[];
    unsigned char* As = PS ? ps_smem : As_static;
    unsigned char* Bs = PS ? ps_smem + 3 * BM * 64 : Bs_static;
    unsigned char* As_w = As;
    unsigned char* Bs_w = Bs;

    // PS: both roles index their work with 0..255 (a consumer's wave tile, a producer's staging rows)
    const bool producer = PS && threadIdx.x >= 256;
    const int tid = PS ? (int)(threadIdx.x & 255) : (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // ---- staging assignment: 4 float4 of A and 4 of B per thread per tile
    // idx = tid + 256*i -> row = idx >> 3 (0..127), kq = idx & 7 (float4 column)
    // Both operands are fetched with raw buffer loads: an element that must read as zero (rows past M / N, k past
    // K, conv taps that leave the sequence, ids outside the embedding table) gets the byte offset 0xFFFFFFFF,
    // which the buffer range check turns into a zero result -- the loader has no branches and no selects on data.
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (int)0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rs = PRE ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(g.Wimg), 0, (int)0xFFFFFFF0u, 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Wt), 0, (int)0xFFFFFFF0u, 0x00020000);
    auto buf16 = [](const __amdgpu_buffer_rsrc_t& rs, unsigned byte_off) {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 0));
    };
    auto buf4 = [](const __amdgpu_buffer_rsrc_t& rs, unsigned byte_off) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 0));
    };
    int a_off[4];       // element offset of the row's k = 0 from g.A (may be "negative" for rows the tap mask removes)
    int a_t[4];         // time index of the row inside its sequence (conv masking)
    bool a_ok[4];
    int b_off[4];
    bool b_ok[4];
    const int kq = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        // POOL: a thread stages four CONSECUTIVE rows, so that row i + 1 of the max-pool is a value it has loaded anyway
        // (five loads per tile instead of eight)
        const int m = m0 + (POOL ? 4 * (tid >> 3) + i : row);
        a_ok[i] = m < M;
        const int mm = a_ok[i] ? m : 0;
        a_t[i] = mm % g.T;
        if (g.gather) {
            // ids outside the table read as a zero row (what TF's GPU embedding_lookup returns) instead of
            // whatever follows the table in the weight arena; the Python mirror rejects them up front
            const int id = g.gather[mm];
            const bool in_table = (unsigned)id < (unsigned)g.gather_rows;
            a_ok[i] = a_ok[i] && in_table;
            a_off[i] = (in_table ? id : 0) * g.lda;
        } else {
            a_off[i] = (mm - g.padl) * g.lda;
        }
        const int n = n0 + row;
        b_ok[i] = n < N;
        b_off[i] = (b_ok[i] ? n : 0) * K;
    }

    float4 ra[4], rb[PRE ? 1 : 4];
    uint4 rbi[PRE ? 6 : 1];      // PRE: this thread's six 16-byte pieces of the B tile's image
    // PRE: byte offset of the next tile's image (tiles are requested in the order the image was made in)
    unsigned img_off = PRE ? ((unsigned)(n0 / BN) * (unsigned)((K + BK - 1) / BK) + (unsigned)((g.kt1 > 0 ? g.kt0 : 0) / BK)) * (unsigned)(3 * BN * 64) + (unsigned)tid * 16u
                           : 0u;
    auto load_b_image = [&]() {
#pragma unroll
        for (int i = 0; i < (PRE ? 6 : 0); ++i) rbi[i] = buf16(b_rs, img_off + (unsigned)(4096 * i));
        img_off += 3 * BN * 64;
    };
    float4 ra4 = make_float4(0.f, 0.f, 0.f, 0.f);   // POOL: the raw row behind this thread's four
    unsigned pool_own = 0, pool_nxt = 0;            // POOL (fast path): validity bits of the loaded tile's tap, per row
    bool pool_raw = false;                          // POOL: ra holds raw rows (fast path), to be pooled when stored
    // k order of a convolution whose channel count is a multiple of the tile depth: channel chunk outer, tap
    // inner, so that the k+1 shifted copies of one activation chunk are loaded in consecutive tiles (they hit
    // in L1 / L2) instead of Cin/BK tiles apart.  The weight tile follows the same map.
    const int ktaps = K / g.Cin;
    const bool tap_inner = ktaps > 1 && (g.Cin % BK) == 0;
    // No integer division in the loop.  Whenever all 32 k of a tile share one tap (dense layers: always tap 0; tap_inner
    // convolutions: by construction) the tile's (tap, first k) pair is wave-uniform and advances by a compare per
    // tile.  Otherwise (the post-net bank: 80 channels, tiles straddle taps) every thread carries the tap and the
    // channel of its own four k and advances them by the tile depth.  The tap masks of a row are bits computed once:
    // bit t = tap t of the row lies inside the sequence (up to 16 taps), bit 16 + t = so does the row after it (the
    // max-pool loader, k = 3 only).  Wider kernels than 16 taps take the general path with its division.
    const bool uniform_tap = ktaps == 1 || tap_inner;
    int cur_tap = 0, cur_kb = 0;   // uniform case: of the NEXT tile load_tile is asked for (tiles are requested in order)
    int my_tap = 0, my_ch = 0;     // per-thread case: tap and channel of this thread's four k in the next tile
    {
        const int k_first = g.kt1 > 0 ? g.kt0 : 0;
        if (tap_inner) {
            const int it = k_first / BK;
            cur_tap = it % ktaps;
            cur_kb = cur_tap * g.Cin + (it / ktaps) * BK;
        } else {
            cur_kb = k_first;
        }
        if (!POOL && !uniform_tap) {
            const int kk0 = k_first + 4 * kq;
            my_tap = kk0 / g.Cin;
            my_ch = kk0 - my_tap * g.Cin;
        }
    }
    unsigned tapmask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned m = 0;
        const int nt = POOL ? 3 : (ktaps < 16 ? ktaps : 16);
        for (int t = 0; t < nt; ++t) {
            const int ts = a_t[i] - g.padl + t;
            if (a_ok[i] && ts >= 0 && ts < g.T) m |= 1u << t;
            if (POOL && a_ok[i] && ts >= 0 && ts + 1 < g.T) m |= 1u << (16 + t);
        }
        tapmask[i] = m;
    }
    const bool fast = POOL ? (uniform_tap && ktaps <= 3) : ktaps <= 16;
    auto load_tile = [&](int kt) {
        if (fast) {
            const bool per_thread = !POOL && !uniform_tap;
            const int tap = per_thread ? my_tap : cur_tap;
            const int kk = per_thread ? my_tap * g.Cin + my_ch : cur_kb + 4 * kq;
            const bool kin = kk < K;
            if (POOL) {
                // rows r .. r + 4 of this thread, each loaded once: row j is wanted as itself (own_j: its tap lies inside
                // the sequence) or as the successor of row j - 1 (nxt_{j-1}); at a sequence boundary the two differ, hence
                // the selects: pooled_i = nxt_i ? max(x_i, x_{i+1}) : x_i with x_i = own_i ? raw_i : 0
                bool own[4], nxt[4];
                pool_raw = true;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    own[i] = kin && ((tapmask[i] >> tap) & 1u);
                    nxt[i] = kin && ((tapmask[i] >> (16 + tap)) & 1u);
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    // (a row wanted only as a successor is addressed from its predecessor: past the last row of the
                    // tensor this thread's own offset for it is not a row's)
                    const bool mine = j < 4 && own[j < 4 ? j : 3];
                    const bool succ = j > 0 && nxt[j > 0 ? j - 1 : 0];
                    const int base = mine ? a_off[j < 4 ? j : 3] : a_off[j > 0 ? j - 1 : 0] + g.lda;
                    const float4 v = buf4(a_rs, (mine || succ) ? (unsigned)(base + kk) * 4u : 0xFFFFFFFFu);
                    if (j < 4) ra[j] = v;
                    else ra4 = v;
                }
                // the values are only combined when the tile is stored (store_tile): touched here, the loads would have to
                // land before the MFMAs of the tile in front of them are even issued
                pool_own = (own[0] ? 1u : 0u) | (own[1] ? 2u : 0u) | (own[2] ? 4u : 0u) | (own[3] ? 8u : 0u);
                pool_nxt = (nxt[0] ? 1u : 0u) | (nxt[1] ? 2u : 0u) | (nxt[2] ? 4u : 0u) | (nxt[3] ? 8u : 0u);
                if (PRE) load_b_image();
                else {
#pragma unroll
                    for (int i = 0; i < (PRE ? 0 : 4); ++i)
                        rb[i] = buf4(b_rs, (b_ok[i] && kin) ? (unsigned)(b_off[i] + kk) * 4u : 0xFFFFFFFFu);
                }
            } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = kin && ((tapmask[i] >> tap) & 1u);
                const unsigned off = ok ? (unsigned)(a_off[i] + kk) * 4u : 0xFFFFFFFFu;
                ra[i] = buf4(a_rs, off);
                if (!PRE) rb[PRE ? 0 : i] = buf4(b_rs, (b_ok[i] && kin) ? (unsigned)(b_off[i] + kk) * 4u : 0xFFFFFFFFu);
            }
            if (PRE) load_b_image();
            }
            // advance to the next tile
            if (tap_inner) {          // tap inner, channel chunk outer
                if (++cur_tap == ktaps) { cur_tap = 0; cur_kb += BK - (ktaps - 1) * g.Cin; }
                else cur_kb += g.Cin;
            } else if (uniform_tap) {
                cur_kb += BK;
            } else if (!POOL) {
                my_ch += BK;
                while (my_ch >= g.Cin) { my_ch -= g.Cin; ++my_tap; }
            }
            return;
        }
        const int kk = kt + 4 * kq;
        const bool kin = kk < K;
        const int tap = kk / g.Cin;  // all four floats share the tap (Cin % 4 == 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ts = a_t[i] - g.padl + tap;
            const bool ok = a_ok[i] && kin && ts >= 0 && ts < g.T;
            const unsigned off = ok ? (unsigned)(a_off[i] + kk) * 4u : 0xFFFFFFFFu;
            float4 v = buf4(a_rs, off);
            if (POOL) v = max4(v, buf4(a_rs, (ok && ts + 1 < g.T) ? off + (unsigned)g.lda * 4u : off));
            ra[i] = v;
            if (!PRE) rb[PRE ? 0 : i] = buf4(b_rs, (b_ok[i] && kin) ? (unsigned)(b_off[i] + kk) * 4u : 0xFFFFFFFFu);
        }
        if (PRE) load_b_image();
    };
    auto store_tile = [&]() {
        if (POOL && pool_raw) {   // pooled_i = nxt_i ? max(x_i, x_{i+1}) : x_i with x_i = own_i ? raw_i : 0 (x_{i+1}: the raw value)
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 x = ((pool_own >> i) & 1u) ? ra[i] : z;
                const float4 nx = i < 3 ? ra[i < 3 ? i + 1 : 3] : ra4;
                ra[i] = ((pool_nxt >> i) & 1u) ? max4(x, nx) : x;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + 32 * i;
            const int arow = POOL ? 4 * (tid >> 3) + i : row;
            store_split4(As_w, arow, kq, ra[i]);
            if (!PRE) store_split4(Bs_w, row, kq, rb[PRE ? 0 : i]);
        }
#pragma unroll
        for (int i = 0; i < (PRE ? 6 : 0); ++i) *reinterpret_cast<uint4*>(Bs_w + tid * 16 + 4096 * i) = rbi[i];
    };

    // 32 x 32 blocks of this wave that lie entirely past M or N get no MFMAs (wave-uniform): the final Dense has
    // N = 1025 = 8 tiles + one column, the second post-net projection N = 80 -- a quarter to three quarters of the
    // edge tile's matrix work is on padding.  Their accumulators stay zero and the epilogue never stores them.
    bool blk_live[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
            blk_live[tm][tn] = (m0 + wm * 64 + tm * 32 < M) && (n0 + wn * 64 + tn * 32 < N);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int k_begin = g.kt1 > 0 ? g.kt0 : 0;
    const int k_end = g.kt1 > 0 ? g.kt1 : K;
    // the MFMAs of one k tile on the LDS images As / Bs, instantiated twice: EDGE = false is the form of every full tile (no
    // tests between the MFMAs), EDGE = true the one of a wave that owns a padded block
    auto mma_tile = [&](auto edge_c) {
        constexpr bool EDGE = decltype(edge_c)::value;
        // Two 16-deep steps per tile.  Lane (li, lh) of a 32-row block holds k = 16 q + 8 lh .. + 7 of row li: chunk
        // 2 q + lh of the row, one ds_read_b128 per split.  The A fragments of both row blocks stay for the step, the B
        // fragments of one column block at a time (36 fragment registers).
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ch = 2 * q + lh;
            uint4 fa[2][3];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const int row = wm * 64 + tm * 32 + li;
                const int off = (row * 4 + (ch ^ ((row >> 2) & 3))) * 16;
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) fa[tm][sp] = *reinterpret_cast<const uint4*>(As + sp * BM * 64 + off);
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int row = wn * 64 + tn * 32 + li;
                const int off = (row * 4 + (ch ^ ((row >> 2) & 3))) * 16;
                uint4 fb[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) fb[sp] = *reinterpret_cast<const uint4*>(Bs + sp * BN * 64 + off);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    if (EDGE && !blk_live[tm][tn]) continue;   // wave-uniform
#define GEMM_MMA(SA, SB)                                                                                              \
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[tm][SA]),    \
                                                                          __builtin_bit_cast(bf16x8_t, fb[SB]), acc[tm][tn], 0, 0, 0);
                    GEMM_MMA(0, 2) GEMM_MMA(2, 0) GEMM_MMA(1, 1) GEMM_MMA(0, 1) GEMM_MMA(1, 0) GEMM_MMA(0, 0)
#undef GEMM_MMA
                }
            }
        }
    };
    auto k_loop = [&](auto edge_c) {
        if (!PS) {
            load_tile(k_begin);
            for (int kt = k_begin; kt < k_end; kt += BK) {
                store_tile();
                __syncthreads();
                if (kt + BK < k_end) load_tile(kt + BK);
                mma_tile(edge_c);
                __syncthreads();
            }
            return;
        }
        // PS: the producers are one tile ahead in LDS and one more in registers
        unsigned char* const A0 = ps_smem, * const B0 = ps_smem + 3 * BM * 64;
        unsigned char* const A1 = ps_smem + 3 * (BM + BN) * 64, * const B1 = A1 + 3 * BM * 64;
        if (producer) {
            load_tile(k_begin);
            As_w = A0; Bs_w = B0;
            store_tile();
            if (k_begin + BK < k_end) load_tile(k_begin + BK);
        }
        __syncthreads();
        int cur = 0;
        for (int kt = k_begin; kt < k_end; kt += BK) {
            if (producer) {
                if (kt + BK < k_end) {
                    As_w = cur ? A0 : A1; Bs_w = cur ? B0 : B1;
                    store_tile();
                    if (kt + 2 * BK < k_end) load_tile(kt + 2 * BK);
                }
            } else {
                As = cur ? A1 : A0; Bs = cur ? B1 : B0;
                mma_tile(edge_c);
            }
            __syncthreads();
            cur ^= 1;
        }
    };
    if (blk_live[0][0] && blk_live[0][1] && blk_live[1][0] && blk_live[1][1]) k_loop(std::false_type{});
    else k_loop(std::true_type{});
    if (producer) return;   // (PS: the staging waves have no part in the epilogue)

    // ---- epilogue.  C/D map of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (g.epi == EPI_HIGHWAY) {
        // Packed columns: every 64-column span holds 32 H units then the same 32 T units, so a
        // lane owns H (tn=0) and T (tn=1) of one unit.  out = relu(h)*sig(t) + x*(1-sig(t)).
        const int span = (n0 + wn * 64) >> 6;
        const int unit = span * 32 + li;
        const int nh = n0 + wn * 64 + li;
        // N (= 2 * units) is a multiple of 64 by construction of the packing.
        const float bh = g.bias ? g.bias[nh] : 0.f;
        const float bt = g.bias ? g.bias[nh + 32] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) {
                    const float hh = fmaxf(acc[tm][0][r] + bh, 0.f);
                    const float tt = sigmoidf_(acc[tm][1][r] + bt);
                    const float x = g.A[(size_t)m * g.lda + unit];
                    g.C[(size_t)m * g.ldc + g.coff + unit] = hh * tt + x * (1.0f - tt);
                }
            }
        return;
    }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int n = n0 + wn * 64 + tn * 32 + li;
        if (n >= N) {
            if (DENORM && g.C2 && n < g.N2) {   // zero the row padding of the second output
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (m < M) g.C2[(size_t)m * g.ldc2 + n] = 0.f;
                    }
            }
            continue;
        }
        const float bv = g.bias ? g.bias[n] : 0.f;
        const float sc = g.scale ? g.scale[n] : 1.f;
        const float sh = g.scale ? g.shift[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) {
                    float v = apply_act(acc[tm][tn][r] + bv, g.act);
                    v = v * sc + sh;
                    if (g.R) v += g.R[(size_t)m * g.ldr + n];
                    if (!DENORM || g.C) g.C[(size_t)m * g.ldc + g.coff + n] = v;
                    if (DENORM && g.C2) {
                        const float db = denorm_db(v, g.d_ref, g.d_range);
                        if (g.d_flag && db < -100.0f) *g.d_flag = 1;
                        g.C2[(size_t)m * g.ldc2 + n] = db_pow(db, g.d_pow);
                    }
                }
            }
    }
}

#ifdef GEMM_EXPERIMENTS
#define GEMM_PRE_OK true
#else
#define GEMM_PRE_OK false
#endif
bool gemm_experiments_built() { return GEMM_PRE_OK; }
template <bool DENORM, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(GEMM_NUM_VGPR))) void gemm_f32_kernel(GemmBatch batch) {
    gemm_body<DENORM, false, PRE && GEMM_PRE_OK>(batch);
}
// the max-pool loader keeps a fifth raw row and the validity bits of the tile in flight: a register budget of its own
// (two waves per SIMD) instead of spilling inside the k loop -- scratch accesses queue behind the tile's global loads
template <bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(GEMM_NUM_VGPR_POOL))) void gemm_f32_pool_kernel(GemmBatch batch) {
    gemm_body<false, true, PRE && GEMM_PRE_OK>(batch);
}

#ifdef GEMM_EXPERIMENTS
// the producer / consumer form (gemm_body, PS): 512 threads, two LDS image pairs in dynamic shared memory
#define GEMM_PS_LDS (2 * 3 * (BM + BN) * 64)
template <bool DENORM, bool PRE>
__global__ __launch_bounds__(512) void gemm_ps_kernel(GemmBatch batch) {
    gemm_body<DENORM, false, PRE, true>(batch);
}
template <bool PRE>
__global__ __launch_bounds__(512) void gemm_ps_pool_kernel(GemmBatch batch) {
    gemm_body<false, true, PRE, true>(batch);
}
template <typename K>
static hipError_t gemm_ps_launch(K kernel, dim3 grid, hipStream_t s, const GemmBatch& b) {
    // (more than 64 KB of dynamic LDS needs the attribute; per device and kernel, cheap: set on every launch)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_PS_LDS);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, grid, dim3(512), GEMM_PS_LDS, s, b);
    return hipGetLastError();
}

// The pre-split image of one weight matrix Wt [N][K] (see gemm_body, PRE): grid (k tiles, 128-row blocks of N), the tiles in
// the order the k loop of a GEMM with this (K, Cin) visits them -- linear, or for a convolution whose channel count is a
// multiple of the tile depth: channel chunk outer, tap inner.  Rows past N and k past K are zeros.
__global__ __launch_bounds__(256) void gemm_pack_weights_kernel(const float* __restrict__ Wt, unsigned char* __restrict__ img, int N, int K, int Cin) {
    const int it = blockIdx.x, nb = blockIdx.y, tid = threadIdx.x;
    const int ktaps = K / Cin;
    const bool tap_inner = ktaps > 1 && (Cin % BK) == 0;
    const int kb = tap_inner ? (it % ktaps) * Cin + (it / ktaps) * BK : it * BK;
    unsigned char* tile = img + ((size_t)nb * gridDim.x + it) * (3 * BN * 64);
    const int kq = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const int n = nb * BN + row, kk = kb + 4 * kq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N && kk < K) v = ld4(Wt + (size_t)n * K + kk);
        store_split4(tile, row, kq, v);
    }
}
#endif
size_t gemm_weight_image_bytes(int N, int K) { return (size_t)((N + BN - 1) / BN) * (size_t)((K + BK - 1) / BK) * (3 * BN * 64); }
hipError_t launch_gemm_pack_weights(hipStream_t s, const float* Wt, unsigned char* img, int N, int K, int Cin) {
#ifdef GEMM_EXPERIMENTS
    hipLaunchKernelGGL(gemm_pack_weights_kernel, dim3((K + BK - 1) / BK, (N + BN - 1) / BN), dim3(256), 0, s, Wt, img, N, K, Cin);
    return hipGetLastError();
#else
    (void)s; (void)Wt; (void)img; (void)N; (void)K; (void)Cin;
    return hipErrorNotSupported;
#endif
}

hipError_t launch_gemm(hipStream_t s, const GemmBatch& b, int n_groups) {
    int max_n = 0;
    for (int i = 0; i < n_groups; ++i) {
        const GemmGroup& g = b.g[i];
        max_n = g.N > max_n ? g.N : max_n;
        // the operands are addressed with 32-bit byte offsets (buffer loads): refuse what does not fit
        const size_t a_rows = g.gather ? (size_t)g.gather_rows : (size_t)g.M + (size_t)(g.pool ? 1 : 0);
        if (a_rows * (size_t)g.lda * 4 >= 0xFFFFFFF0ull || (size_t)g.N * (size_t)g.K * 4 >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    }
    const int m_blocks = (b.g[0].M + BM - 1) / BM;
    dim3 grid((m_blocks + 7) / 8 * 8, (max_n + BN - 1) / BN, n_groups);   // M-blocks padded to the XCD count (see the tile map)
    bool denorm = false;
    for (int i = 0; i < n_groups; ++i) denorm = denorm || b.g[i].C2 != nullptr;
    bool pool = false;
    for (int i = 0; i < n_groups; ++i) pool = pool || b.g[i].pool != 0;
    for (int i = 0; i < n_groups; ++i)
        if (pool && (!b.g[i].pool || b.g[i].C2)) return hipErrorInvalidValue;   // a pooled launch is homogeneous, never de-normalising
#ifdef GEMM_EXPERIMENTS
    // pre-split weight images: all groups of a launch or none (api_stages.hip attaches them to every weight it launches with)
    bool pre = true;
    for (int i = 0; i < n_groups; ++i) pre = pre && b.g[i].Wimg != nullptr;
    if (b.ps) {
        if (pool) return pre ? gemm_ps_launch(gemm_ps_pool_kernel<true>, grid, s, b) : gemm_ps_launch(gemm_ps_pool_kernel<false>, grid, s, b);
        if (denorm) return pre ? gemm_ps_launch(gemm_ps_kernel<true, true>, grid, s, b) : gemm_ps_launch(gemm_ps_kernel<true, false>, grid, s, b);
        return pre ? gemm_ps_launch(gemm_ps_kernel<false, true>, grid, s, b) : gemm_ps_launch(gemm_ps_kernel<false, false>, grid, s, b);
    }
    if (pre) {
        if (pool) hipLaunchKernelGGL((gemm_f32_pool_kernel<true>), grid, dim3(256), 0, s, b);
        else if (denorm) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, s, b);
        else hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, s, b);
        return hipGetLastError();
    }
#else
    if (b.ps) return hipErrorNotSupported;
#endif
    if (pool) hipLaunchKernelGGL((gemm_f32_pool_kernel<false>), grid, dim3(256), 0, s, b);
    else if (denorm) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, s, b);
    else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, s, b);
    return hipGetLastError();
}

// out[m][n] = epilogue( sum_s partial[s][m][n] ), slices added in order (fixed summation order)
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int slices, GemmGroup g) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)g.M * g.N;
    if (i >= total) return;
    const int m = (int)(i / g.N), n = (int)(i % g.N);
    float v = 0.f;
    for (int s = 0; s < slices; ++s) v += partial[(size_t)s * total + i];
    v = apply_act(v + (g.bias ? g.bias[n] : 0.f), g.act);
    if (g.scale) v = v * g.scale[n] + g.shift[n];
    if (g.R) v += g.R[(size_t)m * g.ldr + n];
    g.C[(size_t)m * g.ldc + g.coff + n] = v;
}

// The slice count depends on the layer's K only -- never on the batch -- so that an utterance's result does
// not depend on how many others share the batch (the summation order over k is part of the result).
int gemm_splitk_slices(int K) { return K >= 4096 ? 8 : 1; }   // (8: 600 workgroups for the encoder projection at 64 x 150 tokens; 4 left a CU with 1.2)

hipError_t launch_gemm_splitk(hipStream_t s, const GemmGroup& g, int slices, float* partial, int ps) {
    if (slices < 2 || slices > TTS_GEMM_MAX_GROUPS || g.epi != EPI_STD || g.C2) return hipErrorInvalidValue;
    GemmBatch b;
    memset(&b, 0, sizeof(b));
    b.ps = ps;
    const int k_tiles = (g.K + BK - 1) / BK;
    for (int i = 0; i < slices; ++i) {
        GemmGroup p = g;
        p.bias = nullptr; p.scale = nullptr; p.shift = nullptr; p.R = nullptr; p.act = ACT_NONE;
        p.C = partial + (size_t)i * g.M * g.N;
        p.ldc = g.N; p.coff = 0;
        p.kt0 = (int)((long long)k_tiles * i / slices) * BK;
        p.kt1 = (int)((long long)k_tiles * (i + 1) / slices) * BK;
        if (p.kt1 > g.K) p.kt1 = g.K;
        b.g[i] = p;
    }
    hipError_t e = launch_gemm(s, b, slices);
    if (e != hipSuccess) return e;
    const size_t total = (size_t)g.M * g.N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, partial, slices, g);
    return hipGetLastError();
}

}  // namespace tts
