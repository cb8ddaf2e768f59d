// Griffin-Lim phase reconstruction, STFT analysis and the spectrogram de-normalisation (gfx950).
//
// Replaces, batched and on device:
//   griffin_lim_v2 / spectrogram_to_wav   reference audio/synthesis.py:43-125, 5-40
//     (librosa.istft + librosa.stft per iteration, a per-frame Python loop in the reference)
//   inv_normalize_decibel / decibel_to_magnitude / ** power
//                                          reference audio/conversion.py:81-102, 32-53,
//                                          tacotron/inference.py:93-101, 175
//   librosa.output.write_wav(norm=True)    reference audio/io.py:53 (peak normalisation)
//   linear_scale_spectrogram / mel         reference audio/features.py:5-86, 116-145
//
// Internal layout: frame-major.  mag [B][T][FP] float and a ping-pong pair of 32-bit unit-phasor codes
// e^{i phi} [B][T][FP] (gl_pack_phasor; the estimate X = |S| e^{i phi} is rebuilt from mag where it is
// consumed), FP = 1056 (F = 1025 padded so that every row starts on a 128-byte line: -2.5 % per launch).  One frame's spectrum is a
// contiguous row, which is also how the network produces it (B,T,F): the reference's (F,T) transpose
// exists only at the C ABI.
//
// One launch runs one to three Griffin-Lim iterations on PERSISTENT workgroups (512 threads, one per compute unit) that
// draw RUNS of consecutive frames of one utterance from a global counter: gl_stream_kernel, described where it is
// defined.  The time-domain signal never goes to HBM; per bin and iteration the ALGORITHMIC traffic (what the
// roofline is priced on) is 8 B X in + 4 B |S| + 8 B X out; what really moves with three iterations per launch is a
// 4-byte phasor code in and out per launch plus |S|.  The cut of an utterance into runs is planned on the host:
// gl_plan_stream.
//
// FFT: real 2048-point transforms as 1024-point complex FFTs with a split/merge pass.  One wave
// per FFT, 16 points per lane: radix-16 in registers -> 4x4 register/lane transpose (v_permlane16_swap /
// v_permlane32_swap) -> radix-4 -> LDS transpose -> radix-16; all twiddles and this lane's window samples
// (with the iFFT scale folded in) in registers.  Index math validated against numpy in
// tests/test_host_logic.py (test_fft_decomposition_emulation).  Complex numbers are packed values (v_pk_add /
// v_pk_mul / v_pk_fma_f32 with op_sel / neg modifiers, see "complex helpers"): the kernel is bound by VALU issue
// (one instruction per ~4 cycles and SIMD whatever it is), and a packed instruction costs about as much as a
// scalar one.  This file is still compiled with -fno-slp-vectorize: what the SLP vectoriser packs on its own
// (unrelated scalars, with v_mov shuffles to align the pairs) is slower than leaving it scalar.
//
// Compile-time switches used by the tools/ micro-benchmarks only: GL_NO_ALTPRIO, GL_FFT_LDS_STAGE1 (first
// exchange through LDS), GL_NO_STREAMING_HINT, E1S / E2S (exchange strides).
#include "tts_common.h"
#include "griffin_lim.h"
#include <algorithm>
#include <cmath>
#include <mutex>
#include <cstdlib>

namespace tts {

// ------------------------------------------------------------------------------------ complex helpers
// A complex number is ONE packed value (an aligned 64-bit register pair): gfx950 issues a VALU instruction per
// wave every ~4 cycles whether it is v_add_f32 or v_pk_add_f32 (tools/valu_microbench2.hip: 1.72 ns against
// 1.84 ns per instruction and SIMD), so complex add / sub cost one instruction and a complex multiply two
// (v_pk_mul_f32 + v_pk_fma_f32).  Multiplications by +-i, conjugations and the real / imaginary broadcasts of
// the multiply are the op_sel / neg_lo / neg_hi source modifiers of the packed instructions; hipcc does not
// form those from shuffles (it emits v_mov + v_xor), hence the one-line asm statements.  Plain asm, not
// volatile: the compiler still schedules and removes them like any other pure operation.
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf cmk(float a, float b) { return (cf){a, b}; }
__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
__device__ __forceinline__ cf cscale(cf a, float s) { return a * s; }
__device__ __forceinline__ cf cconj(cf a) { return cmk(a.x, -a.y); }
// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ cf cadd_mi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + (+i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ cf cadd_pi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// conj(a + i b) = (a.x - b.y, -a.y - b.x)
__device__ __forceinline__ cf cconj_add_pi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// conj(a - b) = (a.x - b.x, -a.y + b.y)
__device__ __forceinline__ cf cconj_sub(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + conj(b), a - conj(b)
__device__ __forceinline__ cf cadd_conj(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf csub_conj(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a * b
__device__ __forceinline__ cf cmul(cf a, cf b) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// a * conj(b) = (a.x b.x + a.y b.y, a.y b.x - a.x b.y)
__device__ __forceinline__ cf cmul_conj(cf a, cf b) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// a * k for a compile-time constant k, which lives in a scalar register pair
__device__ __forceinline__ cf cmul_k(cf a, cf k) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(k));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "s"(k), "v"(t));
    return r;
}

// Workgroup barrier that orders LDS accesses only.  __syncthreads() also waits for every outstanding global
// load and STORE of the wave (s_waitcnt vmcnt(0)); at the end of phase B that is the full HBM write latency of a
// frame's spectrum row, three times per chunk, for nothing: no wave ever reads what another wave stored to
// global memory inside this kernel.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Progress flags of the overlap-add (LDS words polled / written by single lanes).  The pointer is cast to the LDS
// address space explicitly: a `volatile int*` derived from the dynamic shared array stays a GENERIC pointer (address
// space inference skips volatile accesses), and the access becomes flat_load / flat_store ... sc0 sc1 followed by
// s_waitcnt vmcnt(0) -- every round then waits for the next frame's prefetched spectrum row before its overlap-add
// instead of in the next round's unpacking.  These are ds_read_b32 / ds_write_b32 and touch lgkmcnt only.
typedef __attribute__((address_space(3))) int gl_lds_int;
__device__ __forceinline__ int gl_flag_load(int* p) {
    return __hip_atomic_load((gl_lds_int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void gl_flag_store(int* p, int v) {
    __hip_atomic_store((gl_lds_int*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void wave_lds_sync() {
    // LDS hand-off between lanes of ONE wave.  The LDS unit executes one wave's DS operations in
    // issue order, so a ds_read issued after a ds_write of the same wave observes it for every
    // lane: no s_waitcnt is needed, only a compiler-level ordering point (the compiler still waits
    // on lgkmcnt before it USES a loaded register).
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// forward radix-4 butterfly (W4 = -i): 8 packed adds
__device__ __forceinline__ void r4(cf& a, cf& b, cf& c, cf& d) {
    const cf s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = csub(b, d);
    a = cadd(s0, s2);
    c = csub(s0, s2);
    b = cadd_mi(s1, s3);   // a - i b - c + i d
    d = cadd_pi(s1, s3);   // a + i b - c - i d
}
// the same with c standing for (-i) c: the W16^4 twiddle of the 16-point transform folded into the butterfly
__device__ __forceinline__ void r4_c_mi(cf& a, cf& b, cf& c, cf& d) {
    const cf s0 = cadd_mi(a, c), s1 = cadd_pi(a, c), s2 = cadd(b, d), s3 = csub(b, d);
    a = cadd(s0, s2);
    c = csub(s0, s2);
    b = cadd_mi(s1, s3);
    d = cadd_pi(s1, s3);
}

// -a - i b = (-a.x + b.y, -a.y - b.x) and -a + i b = (-a.x - b.y, -a.y + b.x)
__device__ __forceinline__ cf cneg_add_mi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf cneg_add_pi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,1] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// The radix-4 butterfly with inputs KNOWN to be zero (ZA: a, ZD: d): the frame a forward transform is fed is zero outside
// the window's 128-sample slots, i.e. in the first and last registers of a lane (fft_input), and x + 0 is not something
// the compiler may drop (-0 + 0 = +0), let alone through the asm statements.  6 packed adds with one zero, 4 with two.
template <bool ZA, bool ZD>
__device__ __forceinline__ void r4z(cf& a, cf& b, cf& c, cf& d) {
    if (ZA && ZD) {          // s0 = c, s1 = -c, s2 = s3 = b
        const cf b0 = b, c0 = c;
        a = cadd(c0, b0);
        c = csub(c0, b0);
        b = cneg_add_mi(c0, b0);
        d = cneg_add_pi(c0, b0);
    } else if (ZA) {         // s0 = c, s1 = -c
        const cf c0 = c, s2 = cadd(b, d), s3 = csub(b, d);
        a = cadd(c0, s2);
        c = csub(c0, s2);
        b = cneg_add_mi(c0, s3);
        d = cneg_add_pi(c0, s3);
    } else if (ZD) {         // s2 = s3 = b
        const cf b0 = b, s0 = cadd(a, c), s1 = csub(a, c);
        a = cadd(s0, b0);
        c = csub(s0, b0);
        b = cadd_mi(s1, b0);
        d = cadd_pi(s1, b0);
    } else {
        r4(a, b, c, d);
    }
}

// forward 16-point DFT in registers, natural order in and out: out[k] = sum_j v[j] W16^{jk}
// (64 packed adds + 8 complex multiplies by constants = 80 VALU instructions).  ZLO / ZHI: the inputs v[j], j < ZLO or
// j > ZHI, are known to be zero (their registers are not read): the reference window's [3, 12] saves 12 of the 32 adds of step 1
template <int ZLO = 0, int ZHI = 15>
__device__ __forceinline__ void fft16(cf (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    // step 1: for each j1, radix-4 over j2 (elements j1 + 4 j2) -> t[j1][k2] stored at v[j1 + 4 k2]
    {
        // (only the two patterns r4z knows are used: a zero first and / or last input, the middle two present)
        constexpr bool ok = ZLO >= 0 && ZLO <= 4 && ZHI >= 11 && ZHI <= 15;
        static_assert(ok || (ZLO == 0 && ZHI == 15), "fft16: zero inputs in the first and last four registers only");
        r4z<(0 < ZLO), (12 > ZHI)>(v[0], v[4], v[8], v[12]);
        r4z<(1 < ZLO), (13 > ZHI)>(v[1], v[5], v[9], v[13]);
        r4z<(2 < ZLO), (14 > ZHI)>(v[2], v[6], v[10], v[14]);
        r4z<(3 < ZLO), (15 > ZHI)>(v[3], v[7], v[11], v[15]);
    }
    // twiddle t[j1][k2] *= W16^{j1 k2}; W^4 = -i (t[2][2]) is folded into the second butterfly of k2 = 2
    v[1 + 4] = cmul_k(v[1 + 4], cmk(C1, -S1));    // W^1
    v[1 + 8] = cmul_k(v[1 + 8], cmk(R2, -R2));    // W^2
    v[1 + 12] = cmul_k(v[1 + 12], cmk(S1, -C1));  // W^3
    v[2 + 4] = cmul_k(v[2 + 4], cmk(R2, -R2));    // W^2
    v[2 + 12] = cmul_k(v[2 + 12], cmk(-R2, -R2)); // W^6
    v[3 + 4] = cmul_k(v[3 + 4], cmk(S1, -C1));    // W^3
    v[3 + 8] = cmul_k(v[3 + 8], cmk(-R2, -R2));   // W^6
    v[3 + 12] = cmul_k(v[3 + 12], cmk(-C1, S1));  // W^9
    // step 2: for each k2, radix-4 over j1 -> out[k2 + 4 k1] ; data for k2 sits at v[4 k2 + j1]
    cf o[16];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        cf a = v[4 * k2 + 0], b = v[4 * k2 + 1], c = v[4 * k2 + 2], d = v[4 * k2 + 3];
        if (k2 == 2) r4_c_mi(a, b, c, d);
        else r4(a, b, c, d);
        o[k2 + 0] = a; o[k2 + 4] = b; o[k2 + 8] = c; o[k2 + 12] = d;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = o[i];
}

#ifndef E1S
#define E1S 80   // row stride (complex) of the first exchange image: 2*E1S = 32 (mod 64) banks
#endif
#ifndef E2S
#define E2S 17
#endif
#ifndef EX_CPLX
#ifdef GL_FFT_LDS_STAGE1
#define EX_CPLX 1280   // 16 rows of E1S
#else
#define EX_CPLX 1088   // 64 rows of E2S (the only exchange image) >= the 1024 bins of the merge pass
#endif
#endif

struct FftTw {
    const cf* a;   // LDS table: a[(k2-1)*64] = W1024^{lane*k2}, k2 = 1..15 (already offset by lane)
    cf b[3];       // W64^{(lane&15)*d}, d = 1..3
    __device__ __forceinline__ cf a_at(int k2) const { return a[(k2 - 1) * 64]; }
};
struct FftTwReg {  // the same twiddles held in registers for the whole kernel (no LDS reads inside the FFT)
    cf a[15];
    cf b[3];
    __device__ __forceinline__ cf a_at(int k2) const { return a[k2 - 1]; }
};

// forward 1024-point complex FFT across one wave.  in: v[j] = z[lane + 64 j]; out: v[c] = Z[lane + 64 c].
// Exchange a register-index bit with a lane-index bit, for the pair of complex registers (a, b):
// v_permlane32_swap / v_permlane16_swap transpose the 2 x 2 block {a, b} x {lane bit 5 (or 4) = 0, 1}:
// afterwards a holds [a.lo | b.lo] and b holds [a.hi | b.hi] (halves of 32 lanes, or rows of 16).
__device__ __forceinline__ void swap_bit5(cf& a, cf& b) {
    auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = cmk(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
    b = cmk(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}
__device__ __forceinline__ void swap_bit4(cf& a, cf& b) {
    auto rx = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    auto ry = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = cmk(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
    b = cmk(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}

// forward 1024-point complex FFT across one wave.  in: v[j] = z[lane + 64 j]; out: v[c] = Z[lane + 64 c].
// Index split n = lane + 64 j, k = k2 + 16 k1', ...: radix-16 over j in registers, twiddle, then the
// element (row k2, column lane) has to reach lane (a = lane & 15, kq = k2 & 3) register (k2 >> 2, lane >> 4):
// a 4 x 4 transpose between the two low register-index bits and the two high lane bits, done with
// 32 permlane swaps (no LDS); radix-4; the second exchange (a 16 x 16 transpose inside each row of 16
// lanes) goes through the wave's LDS buffer; radix-16.
template <int ZLO = 0, int ZHI = 15, typename TW>
__device__ __forceinline__ void fft1024(cf (&v)[16], cf* ex, const TW& tw, int lane) {
    fft16<ZLO, ZHI>(v);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) v[k2] = cmul(v[k2], tw.a_at(k2));
#ifdef GL_FFT_LDS_STAGE1
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) ex[k2 * E1S + lane] = v[k2];
    wave_lds_sync();
    {
        const int a = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b) v[4 * i + b] = ex[(kq + 4 * i) * E1S + a + 16 * b];
    }
    wave_lds_sync();
#else
    // new v[4 i + b] at lane (a, kq) = old v[4 i + kq] at lane (a, b)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        swap_bit4(v[4 * i + 0], v[4 * i + 1]);
        swap_bit4(v[4 * i + 2], v[4 * i + 3]);
        swap_bit5(v[4 * i + 0], v[4 * i + 2]);
        swap_bit5(v[4 * i + 1], v[4 * i + 3]);
    }
#endif
    const int a = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
#pragma unroll
        for (int d = 1; d < 4; ++d) v[4 * i + d] = cmul(v[4 * i + d], tw.b[d - 1]);
#pragma unroll
        for (int d = 0; d < 4; ++d) ex[(16 * d + kq + 4 * i) * E2S + a] = v[4 * i + d];
    }
    wave_lds_sync();
#pragma unroll
    for (int x = 0; x < 16; ++x) v[x] = ex[lane * E2S + x];
    wave_lds_sync();
    fft16(v);
}

// Bins MH - k of a spectrum row whose bins k = lane + 64 c this lane holds in z[c]: m[c] = Z[MH - lane - 64 c],
// through the wave's exchange buffer (one pass of 8-byte writes and reads).  Bin MH itself (wanted by lane 0,
// c = 0) is in no register: the caller passes it.
__device__ __forceinline__ void mirror_bins(const cf (&z)[16], cf (&m)[16], cf* ex, int lane, cf nyq) {
    cf* wr = ex + lane;
    const cf* rd = ex + (1024 - lane);
#pragma unroll
    for (int c = 0; c < 16; ++c) wr[64 * c] = z[c];
    if (lane == 0) ex[1024] = nyq;
    wave_lds_sync();
#pragma unroll
    for (int c = 0; c < 16; ++c) m[c] = rd[-64 * c];
    wave_lds_sync();
}

// ------------------------------------------------------------------------------------ GL iteration
// PAIR-OWNER layout of a spectrum row inside the streaming kernel (round 4).  The real-FFT merge and split passes couple bin
// k with bin MH - k: X[k] and X[MH - k] come from the same (Z[k], Z[MH - k]) with ONE twiddle product,
//     E = Z[k] + conj Z[MH-k],  o = W^k (Z[k] - conj Z[MH-k]):   2 X[k] = E - i o,   2 X[MH-k] = conj(E + i o)
// and likewise the two inputs of the inverse transform from (G[k], G[MH-k]).  So a lane OWNS the eight pairs of its bins
// k = lane + 64 c, c < 8 (k < 512): slot c of a row register set holds bin k, slot 8 + c bin MH - k (for lane 0, c = 0 that
// is the Nyquist bin MH; bin 512, its own mirror, is lane 0's extra `mid` value).  Magnitudes and phasor codes are loaded
// and stored straight in this layout (the mirrored half is a descending, still contiguous 256-byte access), the
// normalisation works on it unchanged, and only half of a transform's values cross lanes: the forward FFT's upper
// registers (bins >= 512) go to their owners, the owners send the inverse FFT's upper inputs back.  6 VALU instructions per
// PAIR and 8 + 8 LDS accesses per pass instead of 5 per BIN and 16 + 16 (every lane fetching the mirror of each of its 16
// bins and multiplying by its twiddle): -64 VALU and -32 LDS instructions per frame and iteration.
#define GL_BIN(lane, j) ((j) < 8 ? (lane) + 64 * (j) : MH - (lane) - 64 * ((j) - 8))
// element j of a row through its two lane bases LO = row + lane, HI = row + (MH - lane): the offsets are instruction constants
#define GL_ROW(LO, HI, j) ((j) < 8 ? (LO)[64 * (j)] : (HI)[-64 * ((j) - 8)])
#define GL_NW 8            // waves per workgroup
#define GL_THREADS 512
#define NFFT 2048
#define MH 1024            // NFFT / 2

// The spectra are streamed once per iteration (1.3 GB per launch at the bench size): non-temporal accesses
// keep them from evicting the decoder's weights and attention memory, which the second stream re-reads
// every step while this kernel runs.
#ifdef GL_PLAIN_LOAD
__device__ __forceinline__ float gl_stream_load(const float* p) { return *p; }
#else
__device__ __forceinline__ float gl_stream_load(const float* p) { return __builtin_nontemporal_load(p); }
#endif
__device__ __forceinline__ void gl_stream_store(cf* p, cf v) { __builtin_nontemporal_store(v, p); }
#define GL_STREAM_LOAD(ptr) gl_stream_load(ptr)
#define GL_STREAM_STORE(ptr, val) gl_stream_store((ptr), (val))
// The state between iterations is the UNIT PHASOR of every bin, 32 bits each (round 2; the estimate X = |S| e^{i phi} it
// stands for is rebuilt in phase A from |S|, which phase A reads anyway from then on and phase B no longer does:
// 4 B phasor + 4 B |S| in, 4 B phasor out = 12 instead of 20 bytes per bin and iteration through a memory path that
// gives a compute unit ~21 GB/s however many waves ask, tools/stream_microbench.hip).  Code (round 4, second form): the
// point where the phasor's ray meets the diamond |Re| + |Im| = 1, stored as its imaginary part p = Im / (|Re| + |Im|)
// (|p| <= 1, a float) whose lowest mantissa bit carries the sign of Re.  Decoding: |Re| = 1 - |p| on the diamond, then
// one reciprocal square root of p^2 + (1 - |p|)^2 (between 1/2 and 1: well conditioned everywhere) brings the point
// back to the circle, times |S|.  No normalisation by |x| is needed to encode (the ratio does not see the scale), no
// compare and no select at either end, and the zero bin needs no special path: 0 / max(0, tiny) = 0 decodes to (1, 0),
// numpy's exp(1j * angle(0)) (and -0.0 + 0j to (-1, 0), also numpy's).  6 + 7 VALU instructions per bin and launch (the
// first form -- the smaller component over the larger, two flag bits -- took 11 + 10).  Worst-case error of a decoded
// component 4e-7 (one dropped mantissa bit of p, one v_rcp_f32, one v_rsq_f32; tests/test_host_logic.py::
// test_phasor_code_emulation), the size of the rounding in the x * rsq(|x|^2) * |S| product it replaces.
// Seeded start: the initial phasor e^{2 pi i u} of a bin, u = 24 bits of a 32-bit mix (lowbias32) of the seed and the
// bin's index in the reference's (B, F, T) layout.  The reference draws np.random.rand per call (audio/synthesis.py:91);
// a counter-based draw gives every bin its number wherever it is needed, so the first Griffin-Lim launch of a call makes
// its own initial estimate instead of reading 4 bytes per bin that another kernel wrote.  v_sin_f32 / v_cos_f32 take
// revolutions; their ~1e-6 error is a perturbation of a random angle.
__device__ __forceinline__ cf gl_seed_phasor(unsigned long long seed, unsigned long long idx) {
    unsigned x = (unsigned)idx ^ ((unsigned)(idx >> 32) * 0x9E3779B9u) ^ (unsigned)seed ^ ((unsigned)(seed >> 32) * 0x85EBCA6Bu);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    return cmk(__builtin_amdgcn_cosf(u), __builtin_amdgcn_sinf(u));
}
__device__ __forceinline__ unsigned gl_pack_phasor(cf x) {   // x: any scale
    // p = Im / (|Re| + |Im|), bit 0 = sign of Re (6 VALU instructions; see the comment above gl_seed_phasor)
    const float den = fmaxf(fabsf(x.x) + fabsf(x.y), 1.0e-30f);
    const float p = x.y * __builtin_amdgcn_rcpf(den);
    return (__float_as_uint(p) & ~1u) | (__float_as_uint(x.x) >> 31);
}
// mag >= 0 (every producer of the internal magnitude buffer writes |S|: gl_to_internal_kernel, the de-normalising
// epilogues).  The flag bit is NOT stripped from p before it is used: one ulp of p, exactly what stripping (a
// truncation) costs.  7 VALU instructions.
__device__ __forceinline__ cf gl_unpack_phasor(unsigned c, float mag) {   // -> mag * phasor
    const float p = __uint_as_float(c);
    const float q = 1.0f - fabsf(p);                                       // |Re| on the diamond |Re| + |Im| = 1
    const float xs = __uint_as_float(__float_as_uint(q) | (c << 31));
    const float g = mag * __builtin_amdgcn_rsqf(fmaf(p, p, q * q));       // 1/2 <= p^2 + q^2 <= 1
    return cmk(xs, p) * g;
}

// |S| x / |x|, and (|S|, 0) for a zero bin (numpy's exp(1j * angle(0)) = 1) WITHOUT a compare and two selects per bin: 1e-17 is
// added to the real part (a zero bin becomes (1e-17, 0) and normalises to (|S|, 0) within 5e-5; a bin of 1e-9 -- the spectrum
// X / 1024 of a frame at the -100 dB floor -- moves by 1e-8 of itself, below the rounding of the FFT that made it) and 1e-38 to
// |x|^2 (so that no reciprocal square root of zero is ever multiplied by zero).  6 VALU instructions per bin.
__device__ __forceinline__ cf gl_normalise(cf x, float mag) {
    const float xr = x.x + 1.0e-17f;
    const float s2 = fmaf(xr, xr, fmaf(x.y, x.y, 1.0e-38f));
    const float g = mag * __builtin_amdgcn_rsqf(s2);
    return cmk(xr, x.y) * g;
}
// The state of a bin between launches: the 32-bit phasor code (4 B in, 4 B out per bin and launch, 7 + 6 VALU instructions to
// decode / encode).  (Round 4 measured the raw spectrum value instead -- 8 B each way, decoded by the normalisation, encoded
// by nothing: the bytes cost more than the instructions, HISTORY.md part C; that build switch is in git history.)
typedef unsigned gl_state_t;
__device__ __forceinline__ gl_state_t gl_state_encode(cf x) { return gl_pack_phasor(x); }
__device__ __forceinline__ cf gl_state_decode(gl_state_t s, float mag) { return gl_unpack_phasor(s, mag); }
size_t gl_state_bytes() { return sizeof(gl_state_t); }

// tools-only ablations (garbage results, timing only): -DGL_ABL_NOSTORE drops the spectrum stores, -DGL_ABL_NOLOAD the
// spectrum loads, -DGL_ABL_NOFLAG the waits of the overlap-add chain; -DGL_CLOCK logs the shader clock of every launch
#ifdef GL_ABL_NOLOAD
#define GL_ABL_LD(load, fake) (fake)
#else
#define GL_ABL_LD(load, fake) (load)
#endif

#ifdef GL_CLOCK   // tools only: shader clock held during every launch (workgroup 0), read back by gl_clock_dump()
__device__ unsigned long long gl_clock_log[4096][2];
__device__ unsigned gl_clock_n;
#endif
// ====================================================================================== streaming form
// One Griffin-Lim iteration (MODE 0) or the final iSTFT (MODE 1) WITHOUT phases: a run of consecutive frames of one
// utterance is a stream.  Frame index i of the run (frame t = run_t0 - halo + i, halo = ncol - 1) belongs to wave
// i mod 8, and a wave's iteration for index i is
//     decode + split + inverse FFT + synthesis window of frame i            (registers, the wave's exchange buffer)
//     overlap-add into a RING of R frames in LDS                            (in index order: a chain, see below)
//     forward FFT + merge + phasor code of frame i - halo                   (final once index i has been added)
// so loads of spectrum rows, arithmetic and stores of new rows interleave all the time instead of taking turns, no
// wave waits at a workgroup barrier inside a run (there is one per run), nothing is copied when the window moves on,
// and both windows stay in registers.  The eight waves drift apart by themselves (the chain staggers them), so the
// two waves of a SIMD are usually in different parts of the iteration.
//
// Ring.  A frame's SPAN is the 128-sample slots of its 2048 padded samples that the window touches (slots c_lo ..
// c_hi of the lane layout f = 2 (lane + 64 c) + e; 10 slots = 1280 samples for 1102): span sample q of index k lives
// at ring position hop (k mod R) + q.  The window is zero on the rest of the two outer slots, so whole slots are
// added (adding zeros) and no lane predicate is needed except in ONE slot: index k ACCUMULATES into span samples
// q < S - hop (what indices < k have written) and STORES the last hop samples (nobody has), so the ring is never
// cleared.  Spans that run past the end of the ring continue linearly into a GUARD of S - hop samples; the first
// index of the next lap (k mod R == 0) reads its accumulate part from the guard and writes it to the start of the
// ring (the guard is zeroed when a run starts: index 0 is such a fold, too).  Overlap-adds happen in index order --
// index i waits for the LDS word `ola_done` to reach i and sets it to i + 1 afterwards (LDS executes a wave's
// operations in order: the flag store follows the data stores) -- which fixes the summation order of every sample
// (bit-reproducible) and is the only synchronisation inside a run: once index i has been added, every frame <= i - halo
// has its final signal, and what the forward FFT of frame i - halo reads is not written again before the ring comes
// round (R >= 9 + halo + ceil(S / hop) + 1: by the time an index may overwrite a position, every wave has finished the
// iteration that read it; gl_stream_ring_frames).  Frames whose window leaves the signal (reflect padding) or whose
// span crosses the lap end (the final values of the wrapped part are at the ring's start, not in the guard) take an
// index-mapped read path: 4 + (halo + ceil(S / hop)) / R of the frames.
// MODE 1 writes out, straight from the overlap-add's registers, the hop samples that index i makes final.
#define GL_NO_ITEM 0x7FFFFFFEu   // "not drawn yet" in the control word of the next item (item ids are < 2^31 - 2)
enum { CT_OLA = 0, CT_SNEXT = 1, CT_OLB = 2 /* chain words of stages 1, 2 */, CT_SWORDS = 16 };

// NST = 2: TWO iterations per launch.  The kernel draws a constant amount of power per instruction and per byte, and
// with the spectra streaming the chip holds a shader clock of ~1.9 GHz against ~2.3 GHz for the same arithmetic without
// memory traffic (tools: -DGL_CLOCK): what shortens the launch is energy, not overlap.  So the second iteration is fed
// from registers: stage A is the iteration above on ring A; its merged spectrum of frame t - lag is normalised to
// |S| e^{i phi} where it stands (no phasor code written, none read back: 8 instead of 12 bytes per bin and iteration,
// ~90 VALU instructions per frame and iteration less) and goes straight into stage B -- inverse FFT, overlap-add into
// ring B in the order of ITS chain, forward FFT of frame t - 2 lag, phasor code, store.  A run then needs halo + lag
// more frames at either end in stage A (2.6 % more transforms for runs of 144 frames).
// SEEDED: stage 0 makes its input from the seed (the first launch of a call without an initial-phase array) -- an
// instantiation of its own: as a run-time branch it cost every launch 38 more spilled registers.
template <int MODE, int WIN_CT, int HOP_CT, bool MSE, int NST = 1, bool SEEDED = false>
__global__ __launch_bounds__(GL_THREADS) void gl_stream_kernel(GlParams p) {
    static_assert(!SEEDED || MODE == 0, "the seeded start is an iteration's");
    static_assert(NST == 1 || (NST >= 2 && NST <= 3 && MODE == 0 && !MSE), "several iterations per launch: plain iterations only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int win = WIN_CT ? WIN_CT : p.win;
    const int hop = HOP_CT ? HOP_CT : p.hop;
    const int ncol = (WIN_CT && HOP_CT) ? (WIN_CT + HOP_CT - 1) / (HOP_CT ? HOP_CT : 1) : p.ncol;
    const int halo = ncol - 1;
    const int wpad = (NFFT - win) >> 1;
    const int c_lo = wpad >> 7;                        // first / last 128-sample slot the window touches
    const int c_hi = (wpad + win - 1) >> 7;
    // the same two as constants where the window is one: registers c < FZ_LO and c > FZ_HI of a forward transform's input
    // are zero (fft_input) and its first radix-16 leaves them out (fft16<ZLO, ZHI>); patterns it does not know: no pruning
    constexpr int fz_lo_ = WIN_CT ? ((NFFT - WIN_CT) >> 1) >> 7 : 0;
    constexpr int fz_hi_ = WIN_CT ? (((NFFT - WIN_CT) >> 1) + WIN_CT - 1) >> 7 : 15;
    constexpr bool fz_ok_ = fz_lo_ <= 4 && fz_hi_ >= 11 && fz_hi_ <= 15;
    constexpr int FZ_LO = fz_ok_ ? fz_lo_ : 0, FZ_HI = fz_ok_ ? fz_hi_ : 15;
    const int n_sl = c_hi - c_lo + 1;
    const int S = 128 * n_sl;                          // span of a frame in the ring
    const int fs = 128 * c_lo;                         // padded sample of span sample 0
    const int acc_len = S - hop;                       // span samples that earlier indices have written
    // A frame's forward FFT runs `lag` indices behind its overlap-add: halo, or one more when the reflect padding of the
    // signal's first frame reaches exactly as far as halo frames make final (windows with ncol * hop == win)
    const int lag = (halo + 1) * hop > 2 * (MH - wpad) ? halo : halo + 1;
    const int R = p.ring_frames;
    const int ring_len = hop * R;
    const int ring_floats = (ring_len + acc_len + 128 + 3) & ~3;
    const int L = hop * (p.T - 1);                     // samples of the (trimmed) signal
    // carve: [exchange: GL_NW * EX_CPLX cf][control][ring A: ring_len + acc_len + 128 floats][ring B]
    cf* ex_all = reinterpret_cast<cf*>(smem_raw);
    int* ctrl = reinterpret_cast<int*>(ex_all + GL_NW * EX_CPLX);
    float* ringA = reinterpret_cast<float*>(ctrl + CT_SWORDS);   // the ring of stage k follows at k * ring_floats

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // the wave index as a SCALAR: everything derived from it (frame index, ring slot, the branches on them) then lives
    // in SGPRs and branches without exec masks -- as `tid >> 6` it is a vector value to the compiler
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf* ex = ex_all + wave * EX_CPLX;

#ifdef GL_CLOCK   // tools only: shader clock held during the launch (s_memtime ticks per 100 MHz s_memrealtime tick)
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef GL_TIMELINE
    if (p.dbg && tid == 0) p.dbg[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    // ---------------- one-time setup: twiddles and BOTH windows in registers
    if (tid == 0) ctrl[CT_SNEXT] = (int)atomicAdd(p.work_counter, 1u);
    // (the counter of the launch before this one on the stream is drained: every workgroup that drew from it has ended)
    if (tid == 0 && blockIdx.x == 0 && p.clear_counter) *p.clear_counter = 0u;
    // twr[j] = W2048^{lane + 64 j} for j < 8; W2048^{512} = -i, so slot j + 8 uses -i twr[j] (folded into the adds)
    cf twr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) twr[j] = reinterpret_cast<const cf*>(p.tw2048)[lane + 64 * j];
    FftTwReg tw;
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) tw.a[k2 - 1] = reinterpret_cast<const cf*>(p.tables)[1024 + (k2 - 1) * 64 + lane];
#pragma unroll
    for (int d = 1; d < 4; ++d) tw.b[d - 1] = reinterpret_cast<const cf*>(p.tw1024)[16 * (lane & 15) * d];
    // p.wlane: [set][lane][c][e], set 0 = analysis window w[n] / (2 MH), set 1 = set 0 / window-sum-square at an
    // interior frame (gl_build_wlane); zero outside the window, statically so for the slots outside the span
    float wana[16][2], wsyn[16][2];
    {
        const float4* wa = reinterpret_cast<const float4*>(p.wlane + (0 * 64 + lane) * 32);
        const float4* ws = reinterpret_cast<const float4*>(p.wlane + (1 * 64 + lane) * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 a4 = wa[q], s4 = ws[q];
            wana[2 * q][0] = a4.x; wana[2 * q][1] = a4.y; wana[2 * q + 1][0] = a4.z; wana[2 * q + 1][1] = a4.w;
            wsyn[2 * q][0] = s4.x; wsyn[2 * q][1] = s4.y; wsyn[2 * q + 1][0] = s4.z; wsyn[2 * q + 1][1] = s4.w;
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (WIN_CT && (c < c_lo || c > c_hi)) { wana[c][0] = wana[c][1] = wsyn[c][0] = wsyn[c][1] = 0.f; }
        // the inverse transform's output is conj(z): the sign of the odd sample is in the window, so that the synthesis window
        // is one packed multiply per pair (as `v * cmk(w0, -w1)` it was two scalar ones: the negation is not a source modifier
        // the compiler forms on a packed operand)
#pragma unroll
        for (int c = 0; c < 16; ++c) wsyn[c][1] = -wsyn[c][1];
    }
    __syncthreads();
    int item = __builtin_amdgcn_readfirstlane(ctrl[CT_SNEXT]);
#ifdef GL_TIMELINE
    int tl_runs = 0;
    const int tl_first = item;
#endif

    // ---------------- work item -> (utterance, first frame, frames, slot of its partial results); wave-uniform
    auto decode_item = [&](int it, int& b, int& t0, int& len, int& slot) {
        const int4 q = p.items[it];   // (wave-uniform index: a scalar load)
        b = q.x; t0 = q.y; len = q.z; slot = q.w;
    };

    // prefetch registers of one spectrum row in the pair-owner layout: phasor codes and magnitudes of bins lane + 64 j
    // (j < 8) and MH - lane - 64 (j - 8) (j >= 8), and of bin 512 (`nyq_*`: the name is older than the layout; only lane 0's is used)
    gl_state_t gc[16];
    float gs[16];
    gl_state_t nyq_c;
    float nyq_s;
    constexpr bool seeded = SEEDED;
#define GLS_LOAD_ROW(BASE_C, BASE_M, TF)                                                        \
    {                                                                                           \
        int tf_ = (TF);                                                                         \
        tf_ = tf_ < 0 ? 0 : (tf_ >= p.T ? p.T - 1 : tf_);                                       \
        const gl_state_t* prow_ = (BASE_C) + (size_t)tf_ * p.FP;                                \
        const float* srow_ = (BASE_M) + (size_t)tf_ * p.FP;                                     \
        const gl_state_t* plo_ = prow_ + lane; const gl_state_t* phi_ = prow_ + (MH - lane);    \
        const float* slo_ = srow_ + lane; const float* shi_ = srow_ + (MH - lane);              \
        if (!seeded) {                                                                          \
            _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) gc[j_] = GL_ROW(plo_, phi_, j_); \
            nyq_c = prow_[MH / 2];                                                              \
        }                                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) gs[j_] = GL_ABL_LD(GL_STREAM_LOAD(j_ < 8 ? slo_ + 64 * j_ : shi_ - 64 * (j_ - 8)), (float)(tf_ + j_ + lane)); \
        nyq_s = srow_[MH / 2];                                                                  \
    }
    // vmcnt counts loads and stores together, in issue order.  The row for the next iteration is requested early in
    // this one and the new row is stored at its end, so at the top of the loop the loads are OLDER than 17 stores: the
    // compiler's wait for "the last load" there (it does not carry the stores round the loop) is a wait for every store
    // of the previous iteration to reach memory -- a full write latency per iteration.  Touching the row registers just
    // before the stores makes the compiler wait for the loads there, where they are the youngest operations and some
    // microseconds old; at the top of the loop nothing is left to wait for (every path into the loop settles them).
#define GLS_TOUCH_ROW()                                                                                              \
    asm volatile("" : "+v"(gc[0]), "+v"(gc[1]), "+v"(gc[2]), "+v"(gc[3]), "+v"(gc[4]), "+v"(gc[5]), "+v"(gc[6]),       \
                      "+v"(gc[7]), "+v"(gc[8]), "+v"(gc[9]), "+v"(gc[10]), "+v"(gc[11]), "+v"(gc[12]), "+v"(gc[13]), \
                      "+v"(gc[14]), "+v"(gc[15]), "+v"(nyq_c));                                                      \
    asm volatile("" : "+v"(gs[0]), "+v"(gs[1]), "+v"(gs[2]), "+v"(gs[3]), "+v"(gs[4]), "+v"(gs[5]), "+v"(gs[6]),       \
                      "+v"(gs[7]), "+v"(gs[8]), "+v"(gs[9]), "+v"(gs[10]), "+v"(gs[11]), "+v"(gs[12]), "+v"(gs[13]), \
                      "+v"(gs[14]), "+v"(gs[15]), "+v"(nyq_s));
    const gl_state_t* x_in = reinterpret_cast<const gl_state_t*>(p.phase_in);
    gl_state_t* x_out = reinterpret_cast<gl_state_t*>(p.phase_out);

#ifdef GL_TIMELINE   // tools only: 100 MHz stamps of workgroup 0's waves, [wave][64], from the tenth iteration of a run on
    int stamp_n = 0;
#define GLS_STAMP()                                                                                       \
    if (p.dbg && blockIdx.x == 0 && lane == 0 && stamp_n < 64 && i >= 10 * GL_NW)                         \
        p.dbg[1024 + wave * 64 + stamp_n++] = __builtin_amdgcn_s_memrealtime();
#else
#define GLS_STAMP()
#endif
    // Issue priority by lateness.  The overlap-add chain makes every wave do one index per round; the hardware
    // arbitrates the two waves of a SIMD by age, so waves 0-3 run ahead, then sleep at the chain while their partners
    // (4-7) run alone -- and a wave alone on a SIMD fills fewer issue slots than two (the timeline showed 2.4 us of
    // waiting per 7.5 us iteration for the older half, none for the younger).  A wave whose next overlap-add is what the
    // chain will ask for next raises its priority, one that is far ahead lowers it: the waves then arrive at the chain
    // about when it is their turn.
#ifndef GL_NO_LATEPRIO
#define GLS_URGENCY(NEXT_INDEX)                                                                    \
    {                                                                                              \
        const int d_ = __builtin_amdgcn_readfirstlane((NEXT_INDEX) - gl_flag_load(ctrl + CT_OLA)); \
        if (d_ <= 1) __builtin_amdgcn_s_setprio(3);                                                \
        else if (d_ <= 3) __builtin_amdgcn_s_setprio(2);                                           \
        else if (d_ <= 5) __builtin_amdgcn_s_setprio(1);                                           \
        else __builtin_amdgcn_s_setprio(0);                                                        \
    }
#else
#define GLS_URGENCY(NEXT_INDEX)
#endif

    // ---------------- the pieces of an iteration (all inlined; `v` is the wave's FFT register set)
    // G in the pair-owner layout (gk[c] = G[k], gk[8 + c] = G[MH - k], k = lane + 64 c; mid = G[512], lane 0's) -> input of
    // the inverse transform, v[j] = Zin[lane + 64 j] (real-FFT split pass): Zin[k] = conj(E + i O), Zin[MH-k] = E - i O with
    // E = G[k] + conj G[MH-k], O = conj(W^k) (G[k] - conj G[MH-k]) (the transform is fed conj(Zin), the two 1/2 are in the
    // window); the owner keeps Zin[k] and sends Zin[MH-k] to the lane that transforms it
    auto split_pass = [&](cf (&gk)[16], cf mid, cf (&v)[16]) __attribute__((always_inline)) {
        cf* wr = ex + (MH - lane);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            cf xk = gk[c];
            cf xm = gk[8 + c];
            if (c == 0 && lane == 0) { xk.y = 0.f; xm.y = 0.f; }   // DC and Nyquist bins are real
            const cf e = cadd_conj(xk, xm);
            const cf o = cmul_conj(csub_conj(xk, xm), twr[c]);
            v[c] = cconj_add_pi(e, o);
            wr[-64 * c] = cadd_mi(e, o);          // (lane 0, c = 0: slot MH, read by nobody)
        }
        if (lane == 0) ex[MH / 2] = cscale(mid, 2.0f);   // bin 512 is its own mirror: Zin = 2 G
        wave_lds_sync();
#pragma unroll
        for (int j = 8; j < 16; ++j) v[j] = ex[lane + 64 * j];
        wave_lds_sync();
    };
    // z[m] = conj(v) / MH, m = lane + 64 c: x[2m] = Re, x[2m+1] = Im; synthesis window (with 1 / window-sum-square)
    auto synth_window = [&](int t, cf (&v)[16]) __attribute__((always_inline)) {
        if (t >= halo && t + halo < p.T) {
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = v[c] * cmk(wsyn[c][0], wsyn[c][1]);   // (wsyn[c][1] carries the conjugation's sign)
        } else {
            // frame near an utterance end: fewer overlapping neighbours, 1 / wss per sample (rare)
            const float* rwp = p.rwss + (size_t)t * hop + wpad;
#pragma unroll
            for (int c0 = 0; c0 < 16; c0 += 4) {
#pragma unroll
                for (int c = c0; c < c0 + 4; ++c) {
                    const int nw0 = 2 * (lane + 64 * c) - wpad;
                    const bool i0 = nw0 >= 0 && nw0 < win, i1 = nw0 + 1 >= 0 && nw0 + 1 < win;
                    const float r0 = rwp[i0 ? nw0 : 0], r1 = rwp[i1 ? nw0 + 1 : 0];
                    v[c] = cmk(i0 ? v[c].x * wana[c][0] * r0 : 0.f, i1 ? -v[c].y * wana[c][1] * r1 : 0.f);
                }
                asm volatile("" ::: "memory");
            }
        }
    };
    // overlap-add of index idx (ring slot s, frame t) into `ring`, in the order of the chain word `chain`
    float pk = 0.f;
    auto overlap_add = [&](float* ring, int chain, int idx, int s, int t, int b, int run_t0, int run_len, const cf (&v)[16])
                           __attribute__((always_inline)) {
        // Everything that does not depend on the ring comes BEFORE the wait for the chain: a chain step -- flag seen, reads,
        // adds, writes, flag passed on -- is what the eight waves of a workgroup do one after another (0.7 us per step in the
        // per-wave timeline; the final iSTFT, one stage per index, is bound by it), so the addresses and the masks of the
        // boundary slot are made while the wave would wait anyway, and what MODE 1 does with the finished samples (stores to
        // the waveform, the running peak) comes AFTER the flag is passed on.
        // Element (float) offsets of this lane's pair in span slot 0, for writing and for reading (the first index of a
        // lap folds the guard in).  One opaque base per PAIR of slots: slot 2k + 1 is 512 bytes behind slot 2k, which the
        // two 8-bit dword offsets of ds_read2 / ds_write2 reach (left to itself the compiler makes an address per slot and
        // direction: 18 VALU instructions per overlap-add).
        // (offsets from the start of the workgroup's LDS, not from `ring`: a stage's ring is a run-time base, and base +
        // constant + 512 is not folded into the instruction's offset fields)
        float* const lds_f = reinterpret_cast<float*>(smem_raw);
        const int e_wr = (int)(ring - lds_f) + hop * s + 2 * lane;
        const int e_rd = e_wr + (s == 0 ? ring_len : 0);
        int wo[8], ro[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            wo[k] = e_wr + 256 * k;
            ro[k] = e_rd + 256 * k;
            if (2 * k < n_sl) asm volatile("" : "+v"(wo[k]), "+v"(ro[k]));
        }
        // the slot where accumulate turns into store: which of this lane's two samples earlier indices have written
        const int qb_edge = 128 * ((acc_len - 1) >> 7);
        unsigned keep0 = qb_edge + 2 * lane < acc_len ? ~0u : 0u, keep1 = qb_edge + 2 * lane + 1 < acc_len ? ~0u : 0u;   // (bit masks)
        asm volatile("" : "+v"(keep0), "+v"(keep1));
#ifndef GL_ABL_NOFLAG
        while (gl_flag_load(ctrl + chain) < idx) __builtin_amdgcn_s_sleep(1);
#endif
        asm volatile("" ::: "memory");
        cf fin[16];   // MODE 1: what the slots hold after this index (the finished samples are among them)
        {
            // All reads first, then the adds, then the writes: written slot by slot, every read waits for the
            // previous slot's write (the compiler cannot tell that they do not alias) and the critical section of
            // the chain is eight LDS round trips instead of one.
            cf o[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int j = c - c_lo;
                o[c] = cmk(0.f, 0.f);
                if (j < 0 || j >= n_sl) continue;                  // wave-uniform (static for the reference window)
                const int qb = 128 * j;
                const float* rp = lds_f + ro[j >> 1] + 128 * (j & 1);
                if (qb < acc_len) o[c] = cmk(rp[0], rp[1]);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int j = c - c_lo;
                if (j < 0 || j >= n_sl) continue;
                const int qb = 128 * j;
                cf a = v[c];                                       // a slot no earlier index has written: a plain store
                if (qb < acc_len) {
                    cf acc = o[c];
                    if (qb + 127 >= acc_len) {                     // (qb == qb_edge)
                        acc.x = __uint_as_float(__float_as_uint(acc.x) & keep0);
                        acc.y = __uint_as_float(__float_as_uint(acc.y) & keep1);
                    }
                    a = cadd(acc, v[c]);
                }
                float* wp = lds_f + wo[j >> 1] + 128 * (j & 1);
                wp[0] = a.x;
                wp[1] = a.y;
                if (MODE == 1) fin[c] = a;
            }
        }
        asm volatile("" ::: "memory");
        if (lane == 0) gl_flag_store(ctrl + chain, idx + 1);
        if (MODE == 1) {
            // span samples [q_fin, q_fin + hop) of this index are final now; y of span sample 0
            asm volatile("" ::: "memory");
            const int q_fin = wpad - fs;
            const int y0 = t * hop + fs - MH;
            const bool emit = (t >= run_t0 || run_t0 == 0) && (t < run_t0 + run_len || run_t0 + run_len == p.T);
            float* wb = p.wav + (size_t)b * L;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int j = c - c_lo;
                if (j < 0 || j >= n_sl) continue;
                const int qb = 128 * j;
                if (qb + 127 >= q_fin && qb < q_fin + hop) {       // slot holds final samples (wave-uniform)
                    const int q = qb + 2 * lane;
                    const int y = y0 + q;
                    const cf a = fin[c];
                    if (emit && q >= q_fin && q < q_fin + hop && y >= 0 && y < L) { wb[y] = a.x; pk = fmaxf(pk, fabsf(a.x)); }
                    if (emit && q + 1 >= q_fin && q + 1 < q_fin + hop && y + 1 >= 0 && y + 1 < L) { wb[y + 1] = a.y; pk = fmaxf(pk, fabsf(a.y)); }
                }
            }
        }
    };
    // windowed input of the forward transform of frame tm, whose overlap-add index was idx - lag (ring slot sm), read in
    // the iteration of index idx of its stage (y_base: trimmed-signal index of that stage's ring coordinate 0)
    auto fft_input = [&](const float* ring, int idx, int sm, int tm, int y_base, cf (&v)[16]) __attribute__((always_inline)) {
        const int ylo = tm * hop + wpad - MH;                    // y index of window sample 0
        const bool edge = ylo < 0 || ylo + win > L;              // reflect padding needed
        // The span is read linearly when it does not cross the end of the lap, or when the next lap's fold (index
        // m + R - sm) has not happened yet: what it will fold is still in the guard, behind the ring.
        if (!edge && (hop * sm + S <= ring_len || R - sm > lag)) {
            // (one opaque LDS base per pair of slots, as in overlap_add; a packed multiply per slot)
            const float* const lds_f = reinterpret_cast<const float*>(smem_raw);
            const int e_rd = (int)(ring - lds_f) + hop * sm + 2 * lane;
            int ro[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                ro[k] = e_rd + 256 * k;
                if (2 * k < n_sl) asm("" : "+v"(ro[k]));
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int j = c - c_lo;
                if (j < 0 || j >= n_sl) { v[c] = cmk(0.f, 0.f); continue; }
                const float* sf = lds_f + ro[j >> 1] + 128 * (j & 1);
                v[c] = cmk(wana[c][0], wana[c][1]) * cmk(sf[0], sf[1]);
            }
        } else {
            // index-mapped reads: reflect at the signal's ends; run coordinate u = lap * ring_len + off lives at
            // ring position off -- unless off < acc_len and that lap's fold (index lap * R) is still to come:
            // then it is in the guard
            const int lap0 = (idx - lag) / R;
            const int ub = y_base + lap0 * ring_len;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float x[2] = {0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int f = 2 * (lane + 64 * c) + e;
                    const int nw = f - wpad;
                    if (nw >= 0 && nw < win) {
                        int y = ylo + nw;
                        y = y < 0 ? -y : y;
                        y = y >= L ? 2 * (L - 1) - y : y;
                        int off = y - ub, lap = lap0;
                        if (off >= ring_len) { off -= ring_len; ++lap; }
                        else if (off < 0) { off += ring_len; --lap; }
                        const int pos = (off < acc_len && lap * R > idx) ? ring_len + off : off;
                        x[e] = wana[c][e] * ring[pos];
                    }
                }
                v[c] = cmk(x[0], x[1]);
            }
        }
    };
    // forward transform output (v[j] = Z[lane + 64 j], after fft1024) -> X / MH in the pair-owner layout (real-FFT merge pass):
    // sink(c, X[k]) and sink(8 + c, X[MH - k]) for k = lane + 64 c, c < 8; returns X[512] (meaningful in lane 0).  The upper
    // registers travel to their owners through the wave's exchange buffer; Z[MH] := Z[0] gives lane 0 its Nyquist bin.
    auto merge_pass = [&](const cf (&v)[16], auto&& sink) __attribute__((always_inline)) -> cf {
#pragma unroll
        for (int j = 8; j < 16; ++j) ex[lane + 64 * j] = v[j];
        if (lane == 0) ex[MH] = v[0];
        wave_lds_sync();
        cf zm[8];   // all mirrored values first: one LDS latency for the pass instead of one per pair
        const cf* rd = ex + (MH - lane);
#pragma unroll
        for (int c = 0; c < 8; ++c) zm[c] = rd[-64 * c];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const cf e = cadd_conj(v[c], zm[c]);
            const cf o = cmul(csub_conj(v[c], zm[c]), twr[c]);
            sink(c, cadd_mi(e, o));               // 2 X[k] = E - i o
            sink(8 + c, cconj_add_pi(e, o));      // 2 X[MH - k] = conj(E + i o)
        }
        wave_lds_sync();
        return cscale(cconj(v[8]), 2.0f);         // 2 X[512] = 2 conj(Z[512])
    };

    const int lead = NST * halo;   // frames the first index (of stage 0) lies before the run
    bool have_row = false;   // (per wave) the first row of this run was requested in the last iteration of the previous one
    while (item < p.n_items) {
        int b, run_t0, run_len, slot;
        decode_item(item, b, run_t0, run_len, slot);
        const float* magb = p.mag + (size_t)b * p.T * p.FP;
        const gl_state_t* phb = x_in + (size_t)b * p.T * p.FP;
        // The next item is drawn LATE in a run (by the wave of index n_idx - 24, read by every wave in its last iteration):
        // drawn at the start, a workgroup committed itself to a second run before it knew how long the first would take, and
        // a cut with runs of two lengths (gl_plan_stream) paired long runs with short ones at random.  Runs too short for
        // that (fewer than four rounds of the waves) still draw it at the start.
        const int n_idx = run_len + NST * (halo + lag);
        const bool late = n_idx >= 4 * GL_NW;
        const int i_res = n_idx - 3 * GL_NW;
        unsigned next_item_reg = GL_NO_ITEM;
        if (tid == 0 && !late) next_item_reg = atomicAdd(p.work_counter, 1u);
        if (!have_row) GLS_LOAD_ROW(phb, magb, run_t0 - lead + wave)
        have_row = false;
        // run start: the guards read as zero for index 0, the chains start at 0
        for (int q = tid; q < acc_len + 128; q += GL_THREADS) {
#pragma unroll
            for (int k = 0; k < NST; ++k) ringA[k * ring_floats + ring_len + q] = 0.f;
        }
        if (tid == 0) { ctrl[CT_OLA] = 0; ctrl[CT_OLB] = 0; ctrl[CT_OLB + 1] = 0; ctrl[CT_SNEXT] = (int)next_item_reg; }
        __syncthreads();
        int next_item = __builtin_amdgcn_readfirstlane(ctrl[CT_SNEXT]);   // (late: GL_NO_ITEM until the wave's last iteration)
        int nb = b, nt0 = 0, nlen = 0, nslot = 0;
        if (!late && next_item < p.n_items) decode_item(next_item, nb, nt0, nlen, nslot);

        // the LAST stage (the only one of NST == 1) has the indices of one iteration, j <-> frame run_t0 - halo + j; every
        // stage before it runs halo + lag indices ahead of the next: stage 0 index i <-> frame run_t0 - NST halo + i
        const int y_base_a = (run_t0 - lead) * hop - MH + fs;   // trimmed-signal index of stage 0's ring coordinate 0 (lap 0)
        float mse_acc = 0.f;
        pk = 0.f;
        int s = wave % R;                                     // ring slot of this wave's index (wave < 8 <= R)
        GLS_TOUCH_ROW()
        for (int i = wave; i < n_idx; i += GL_NW) {
            const int t = run_t0 - lead + i;
            const bool valid = t >= 0 && t < p.T;             // wave-uniform
            unsigned drawn = 0;
            if (late && i == i_res && lane == 0) drawn = atomicAdd(p.work_counter, 1u);   // (consumed before this index's overlap-add)
            if (late && i + GL_NW >= n_idx) {
                // this wave's last iteration: its previous one waited for the chain to pass index i - 8 > i_res, whose wave
                // stored the item before it passed the chain on (LDS executes a wave's operations in order)
                int nv;
                while ((nv = gl_flag_load(ctrl + CT_SNEXT)) == (int)GL_NO_ITEM) __builtin_amdgcn_s_sleep(1);
                next_item = __builtin_amdgcn_readfirstlane(nv);
                if (next_item < p.n_items) decode_item(next_item, nb, nt0, nlen, nslot);
            }
            GLS_STAMP()   // 0: iteration start
            GLS_URGENCY(i)
            cf v[16];
            if (valid) {
                cf gk[16];   // X[k] = |S[k]| * phasor[k]
                if (seeded) {
                    const unsigned long long row0 = (unsigned long long)b * p.F * p.T + (unsigned long long)t;   // bin f: + f T
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const cf e = gl_seed_phasor(p.seed, row0 + (unsigned long long)GL_BIN(lane, j) * p.T);
                        gk[j] = cmk(gs[j] * e.x, gs[j] * e.y);
                    }
                    const cf en = gl_seed_phasor(p.seed, row0 + (unsigned long long)(MH / 2) * p.T);
                    split_pass(gk, cmk(nyq_s * en.x, nyq_s * en.y), v);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) gk[j] = gl_state_decode(gc[j], gs[j]);
                    split_pass(gk, gl_state_decode(nyq_c, nyq_s), v);
                }
            } else {
                // (zeros from asm statements: as constants the sixteen pairs are materialised in FRONT of the branch on
                // `valid` -- 32 v_mov in every iteration for the few indices that lie outside the utterance)
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_mov_b64 %0, 0" : "=v"(v[j]));
            }
            GLS_STAMP()   // 1: decoded, split
            GLS_URGENCY(i)
            // the row is consumed.  NST == 1: request this wave's next one now (of this run, or the first of the next item).
            // NST > 1: the LAST stage requests it; the magnitude registers first take |S| of the frames that go from stage to
            // stage, each requested where its stage starts (below)
#define GLS_NEXT_ROW()                                                                                                   \
            if (i + GL_NW < n_idx) {                                                                                     \
                GLS_LOAD_ROW(phb, magb, t + GL_NW)                                                                       \
            } else if (next_item < p.n_items) {                                                                          \
                GLS_LOAD_ROW(x_in + (size_t)nb * p.T * p.FP, p.mag + (size_t)nb * p.T * p.FP, nt0 - lead + wave)         \
                have_row = true;                                                                                         \
            }
            if (NST == 1) {
                GLS_NEXT_ROW()
            }
            if (valid) {
                fft1024(v, ex, tw, lane);
                synth_window(t, v);
            }
            GLS_STAMP()   // 2: inverse FFT + window done
            if (late && i == i_res && lane == 0) gl_flag_store(ctrl + CT_SNEXT, (int)drawn);
            overlap_add(ringA, CT_OLA, i, s, t, b, run_t0, run_len, v);
            GLS_STAMP()   // 3: overlap-add issued, flag passed on
            if (NST == 1) {
                // the next row has arrived (requested before the inverse FFT) -- settled on every path round the loop, at
                // the point of the iteration where the fewest registers are live, before this iteration's stores are issued
                GLS_TOUCH_ROW()
            }
            GLS_URGENCY(i + GL_NW)

            // ---------------- further stages: stage k runs halo + lag indices behind stage k - 1 and takes the forward
            // transform of stage k - 1's frame t_k = t_{k-1} - lag, normalised to |S| e^{i phi} in registers, as its input.
            // A loop, not NST copies of the code: the instruction cache is shared by two compute units.
            int sk = s, ik = i, tk = t;               // ring slot, index and frame of the current stage
            float* ring_k = ringA;
            int yb_k = y_base_a;
#pragma nounroll
            for (int k = 1; k < NST; ++k) {
                const int i_prev = ik, s_prev = sk;
                const float* ring_prev = ring_k;
                const int yb_prev = yb_k;
                ik -= halo + lag;
                tk -= lag;
                sk -= halo + lag;
                sk += sk < 0 ? R : 0;
                ring_k += ring_floats;
                yb_k += halo * hop;
                const bool valid_k = ik >= 0 && tk >= 0 && tk < p.T;
                {
                    // |S| of this stage's frame, requested where the stage starts (the forward transform covers the loads;
                    // unconditional: a row that is not needed is one that exists).  ONE load site for every further stage --
                    // round 6: requested at the end of the stage before, beside the last stage's request for the next row,
                    // the two sites' registers met in a phi, the allocator gave this site temporaries and copied them into
                    // place behind an `s_waitcnt vmcnt(1)` right after the loads (a whole memory round trip exposed per
                    // iteration), and because those temporaries were transform registers as well, the compiler's waits for
                    // "loads that may still write them" stood in the middle of the transforms and inside the overlap-add's
                    // critical section: 190.5 -> 185.9 us per iteration alone (profiles/r06_experiment_gl_issue.txt)
                    const int tq = tk < 0 ? 0 : (tk >= p.T ? p.T - 1 : tk);
                    const float* mrow_ = magb + (size_t)tq * p.FP;
                    const float* mlo_ = mrow_ + lane; const float* mhi_ = mrow_ + (MH - lane);
#pragma unroll
                    for (int j = 0; j < 16; ++j) gs[j] = GL_ROW(mlo_, mhi_, j);
                    nyq_s = mrow_[MH / 2];
                }
                if (valid_k) {
                    int sm = s_prev - lag;
                    sm += sm < 0 ? R : 0;
                    fft_input(ring_prev, i_prev, sm, tk, yb_prev, v);
                    fft1024<FZ_LO, FZ_HI>(v, ex, tw, lane);
                    GLS_STAMP()   // (a) a further stage's forward transform done
                    cf gk[16];
                    // |S| e^{i phi}: x * (|S| / |x|), and (|S|, 0) for a zero bin (numpy's exp(1j * angle(0)) = 1); the
                    // bins are X / MH of a windowed signal, far from both ends of the float range
                    const cf xmid = merge_pass(v, [&](int c, cf x) { gk[c] = gl_normalise(x, gs[c]); });   // |S| e^{i phi}
                    split_pass(gk, gl_normalise(xmid, nyq_s), v);
                    GLS_STAMP()   // (b) merged, normalised, split
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) asm volatile("v_mov_b64 %0, 0" : "=v"(v[j]));   // (as in stage 0)
                }
                if (k + 1 == NST) {
                    GLS_NEXT_ROW()
                }
                if (ik >= 0) {
                    if (valid_k) {
                        fft1024(v, ex, tw, lane);
                        synth_window(tk, v);
                    }
                    GLS_STAMP()   // (c) inverse transform and window done
                    overlap_add(ring_k, CT_OLB + k - 1, ik, sk, tk, b, run_t0, run_len, v);
                    GLS_STAMP()   // (d) overlap-add done, flag passed on
                }
            }
            if (NST > 1) GLS_TOUCH_ROW()
            GLS_STAMP()   // 4
            // ---------------- forward FFT of the frame `lag` behind in the last stage: its signal is final
            const int jj = ik, sb = sk;
            if (MODE == 0 && jj >= halo + lag && jj < halo + lag + run_len) {
                const int tm = tk - lag;                                 // in [run_t0, run_t0 + run_len), < T
                int sm = sb - lag;
                sm += sm < 0 ? R : 0;
                const float* mrow = magb + (size_t)tm * p.FP;
                float mg[16];
                if (MSE) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) mg[c] = fabsf(GL_STREAM_LOAD(c < 8 ? mrow + lane + 64 * c : mrow + (MH - lane) - 64 * (c - 8)));
                }
                fft_input(ring_k, jj, sm, tm, yb_k, v);
                fft1024<FZ_LO, FZ_HI>(v, ex, tw, lane);
                GLS_STAMP()   // 5: forward FFT done
                GLS_URGENCY(i + GL_NW)
                gl_state_t* orow = x_out + ((size_t)b * p.T + tm) * p.FP;
                gl_state_t* olo = orow + lane;
                gl_state_t* ohi = orow + (MH - lane);
                const cf xmid = merge_pass(v, [&](int c, cf x) {
#ifdef GL_ABL_NOSTORE
                    if (__float_as_uint(x.x) == 0x12345678u) __builtin_nontemporal_store(gl_state_encode(x), c < 8 ? olo + 64 * c : ohi - 64 * (c - 8));
#else
                    __builtin_nontemporal_store(gl_state_encode(x), c < 8 ? olo + 64 * c : ohi - 64 * (c - 8));
#endif
                    if (MSE) {
                        const float d = mg[c] - (float)MH * sqrtf(fmaf(x.x, x.x, x.y * x.y));   // x = X / MH
                        mse_acc += d * d;
                    }
                });
                if (lane == 0) {
                    __builtin_nontemporal_store(gl_state_encode(xmid), orow + MH / 2);   // bin 512
                    if (MSE) {
                        const float d = fabsf(mrow[MH / 2]) - (float)MH * sqrtf(fmaf(xmid.x, xmid.x, xmid.y * xmid.y));
                        mse_acc += d * d;
                    }
                }
                GLS_STAMP()   // 6: merged, encoded, stores issued
            }
            s += GL_NW;
            s -= s >= R ? R : 0;
        }
        __builtin_amdgcn_s_setprio(0);
        // ---------------- run end: per-run partial results (fixed order), then everyone is done with the rings
        if ((MODE == 0 && MSE) || (MODE == 1 && p.peak_partial)) {
            float r = MODE == 0 ? mse_acc : pk;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) r = MODE == 0 ? r + __shfl_xor(r, o) : fmaxf(r, __shfl_xor(r, o));
            __syncthreads();   // all waves are done with their exchange buffers
            float* red = reinterpret_cast<float*>(ex_all);
            if (lane == 0) red[wave] = r;
            __syncthreads();
            if (tid == 0) {
                float a = 0.f;
                for (int w = 0; w < GL_NW; ++w) a = MODE == 0 ? a + red[w] : fmaxf(a, red[w]);
                float* part = (MODE == 0 ? p.mse_partial : p.peak_partial) + (size_t)b * p.slots_per_utt + (slot & 0xffff);
                part[0] = a;
                for (int q = 1; q <= (slot >> 16); ++q) part[q] = 0.f;   // (the utterance's last run: slots other utterances have)
            }
        }
        __syncthreads();
        item = next_item;
#ifdef GL_TIMELINE
        ++tl_runs;
#endif
    }
#ifdef GL_TIMELINE
    if (p.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.dbg[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (blockIdx.x < 512) p.dbg[1536 + blockIdx.x] = (unsigned long long)(unsigned)tl_first | ((unsigned long long)tl_runs << 32) | ((unsigned long long)(xcc & 0xf) << 48);
    }
#endif
#ifdef GL_CLOCK
    if (tid == 0 && blockIdx.x == 0 && MODE == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - clk_t0, dr = __builtin_amdgcn_s_memrealtime() - clk_r0;
        const unsigned n = atomicAdd(&gl_clock_n, 1u);
        if (n < 4096) { gl_clock_log[n][0] = dt; gl_clock_log[n][1] = dr; }
    }
#endif
#undef GLS_LOAD_ROW
#undef GLS_NEXT_ROW
#undef GLS_TOUCH_ROW
#undef GLS_STAMP
#undef GLS_URGENCY
}


// ---- streaming form: geometry, plan, launch
namespace {
struct GlStreamGeom { int c_lo, n_sl, S, acc_len, halo; };
GlStreamGeom gl_stream_geom(int win, int hop) {
    GlStreamGeom g;
    const int wpad = (NFFT - win) >> 1;
    g.c_lo = wpad >> 7;
    g.n_sl = ((wpad + win - 1) >> 7) - g.c_lo + 1;
    g.S = 128 * g.n_sl;
    g.acc_len = g.S - hop;
    g.halo = (win + hop - 1) / hop - 1;
    return g;
}
}  // namespace

// Frames the ring holds (0: the window / hop pair does not fit).  Lower bound: what keeps an index from overwriting ring
// positions that a slower wave may still read (see gl_stream_kernel), and the reflect-padded frames' reach; upper
// bound: LDS.  More frames only make the lap-end read path rarer.
int gl_stream_ring_frames(int win, int hop, int n_stage) {
    const GlStreamGeom g = gl_stream_geom(win, hop);
    if (g.acc_len < 0) return 0;   // hop > span: frames do not even touch (ncol = 1 with a hop beyond the padded slots)
    const int wpad = (NFFT - win) >> 1;
    const int lag = (g.halo + 1) * hop > 2 * (MH - wpad) ? g.halo : g.halo + 1;
    // (n_stage rings share what the exchange buffers leave of the 160 KB)
    const int budget = (160 * 1024 - (int)(GL_NW * EX_CPLX * sizeof(cf)) - CT_SWORDS * (int)sizeof(int)) / (int)sizeof(float) / n_stage - g.acc_len - 132;
    int need = 9 + lag + (g.S + hop - 1) / hop + 1;
    const int reach = (g.S + win + 2 * hop + hop - 1) / hop;   // what a reflect-padded frame reads is still in the ring, within one lap
    need = std::max(need, std::max(reach, GL_NW));
    const int R = std::min(budget / hop, std::max(64, need));
    return R >= need ? R : 0;
}

size_t gl_stream_lds_bytes(const GlParams& p) {
    const GlStreamGeom g = gl_stream_geom(p.win, p.hop);
    return (size_t)(GL_NW * EX_CPLX) * sizeof(cf) + CT_SWORDS * sizeof(int) +
           (size_t)(p.n_stage < 1 ? 1 : p.n_stage) * (size_t)((p.hop * p.ring_frames + g.acc_len + 128 + 3) & ~3) * sizeof(float);
}

// Work items of the streaming form.  The frames of all utterances, one after another, are dealt to the workgroups in
// contiguous pieces of equal COST; a piece that crosses the end of an utterance is two runs (the tail of one utterance
// and the head of the next).  What a run costs beyond its frames was measured per workgroup (tools: -DGL_TIMELINE,
// profiles/r06_experiment_gl_cut.txt): at three iterations per launch every stage starts halo + lag indices before the
// next one's first frame -- 24 indices per run that carry 72 of a frame's 6 transforms, 12 frames' worth with start and
// drain, a little less at an utterance's end where the frames outside are skipped but the reflect-padded ones take the
// index-mapped path.  Until round 6 every utterance was cut alike into runs of one length (a multiple of the eight waves)
// and a rest: at T = 1000, B = 64 on 224 workgroups three runs of 296 frames and one of 112, so that 192 workgroups took
// one long run (640 us) and 32 two short ones (515 us) -- 4.7 % of the chip idle in every launch; on 256 workgroups three of
// 256 and one of 232 (572 / 515 us, 4.5 %).  The waveform's bits do not depend on the cut (every sample is summed over the
// frames that cover it in ascending order whatever run they are in: tests/test_gpu_audio.py).
namespace {
struct GlRun { int b, t0, len; };
struct GlCutCost { double interior, edge; };   // per END of a run, in frames
GlCutCost gl_cut_cost(int halo, int lag, int n_stage) {
    // interior end: (halo + lag) / 2 * n_stage^2 transforms of the 2 n_stage a frame takes = (halo + lag) n_stage / 4
    // frames (6 at 4 / 4 / 3), measured 5.5 with the start and drain of the stream; an utterance's end: about half
    const double c = (halo + lag) * n_stage / 4.0;
    return GlCutCost{c * (5.5 / 6.0), c * 0.5};
}
// deals the frames to `W` workers with at most `M` cost each; returns false if they do not fit.  workers[w] = its runs
bool gl_deal(int T, int B, int W, double M, const GlCutCost& cc, int min_len, std::vector<std::vector<GlRun>>& workers, double* makespan) {
    workers.assign((size_t)W, {});
    int b = 0, t = 0, w = 0;
    double load = 0.0, worst = 0.0;
    while (b < B) {
        if (w >= W) return false;
        const int rest = T - t;
        const double left = t > 0 ? cc.interior : cc.edge;
        const double whole = rest + left + cc.edge;                    // the rest of the utterance as one run
        if (load + whole <= M + 1e-9) {
            workers[w].push_back(GlRun{b, t, rest});
            load += whole;
            ++b; t = 0;
            continue;
        }
        int len = (int)std::floor(M - load - left - cc.interior + 1e-9);   // a run that ends inside the utterance
        if (rest - len < min_len) len = rest - min_len;                     // (never leave a sliver to the next worker)
        if (len >= min_len) {
            workers[w].push_back(GlRun{b, t, len});
            load += len + left + cc.interior;
            t += len;
        } else if (workers[w].empty()) {
            return false;                                                   // M is smaller than the smallest run
        }
        worst = std::max(worst, load);
        ++w; load = 0.0;
    }
    worst = std::max(worst, load);
    if (makespan) *makespan = worst;
    return true;
}
}  // namespace

int gl_plan_items(int T, int B, int win, int hop, int n_workers, int n_stage, int force_runs, int force_run_len,
                  std::vector<int4>* items, int* slots_per_utt, int* workers_out) {
    const int ncol = (win + hop - 1) / hop, halo = ncol - 1;
    n_stage = n_stage < 1 ? 1 : (n_stage > 3 ? 3 : n_stage);
    n_workers = n_workers < 1 ? 1 : n_workers;
    const int wpad = (TTS_GL_NFFT - win) >> 1;
    const int lag = (halo + 1) * hop > 2 * (TTS_GL_NFFT / 2 - wpad) ? halo : halo + 1;
    std::vector<std::vector<GlRun>> workers;
    // tests / experiments only (per-handle options "gl_runs" / "gl_run_len" behind "debug_hooks", api_handle.hip): every
    // utterance cut alike into runs of one length and a rest, one run per list entry
    int forced_len = 0;
    if (force_runs >= 1 && force_runs <= T) forced_len = ((T + force_runs - 1) / force_runs + GL_NW - 1) / GL_NW * GL_NW;
    if (force_run_len >= GL_NW) forced_len = force_run_len / GL_NW * GL_NW;
    if (forced_len > 0) {
        for (int t0 = 0; t0 < T; t0 += forced_len)
            for (int b = 0; b < B; ++b) workers.push_back({GlRun{b, t0, std::min(forced_len, T - t0)}});
    } else {
        const GlCutCost cc = gl_cut_cost(halo, lag, n_stage);
        // the shortest run: a round of the eight waves -- down to half a round where the workgroups outnumber the rounds (one
        // utterance on a whole chip: 250 runs of 4 frames instead of 125 of 8, Griffin-Lim 1.06 -> 0.92 ms per call at B = 1)
        const long long per_worker = (long long)B * T / n_workers;
        const int min_len = std::min(T, (int)std::max<long long>(GL_NW / 2, std::min<long long>(GL_NW, per_worker)));
        // the smallest makespan over a scan of the bound (the deal is greedy: a lower bound does not always give a lower result)
        const double total = (double)B * (T + 2 * cc.edge);
        double lo = std::max(total / n_workers, (double)min_len + 2 * cc.edge), best_t = 1e300;
        std::vector<std::vector<GlRun>> cand;
        for (int step = 0; step < 400; ++step) {
            const double M = lo * (1.0 + 0.0025 * step);
            double t = 0.0;
            if (!gl_deal(T, B, n_workers, M, cc, min_len, cand, &t)) continue;
            if (t < best_t - 1e-9) { best_t = t; workers = cand; }
        }
        if (workers.empty()) {   // (cannot happen: at twice the average every deal fits) one run per utterance
            for (int b = 0; b < B; ++b) workers.push_back({GlRun{b, 0, T}});
        }
    }
    // table order = the order the persistent workgroups draw in: every worker's first run, then the runs that follow in
    // the order their workers come free (the shortest first runs first)
    std::vector<int> slot_of((size_t)B, 0), runs_of((size_t)B, 0);
    for (const auto& w : workers) for (const GlRun& r : w) ++runs_of[r.b];
    int spu = 1;
    for (int b = 0; b < B; ++b) spu = std::max(spu, runs_of[b]);
    std::vector<int> slot_at;   // slot of a run = its ordinal inside the utterance (by first frame)
    auto slot_word = [&](const GlRun& r) {
        int ord = 0;
        for (const auto& w : workers) for (const GlRun& q : w) if (q.b == r.b && q.t0 < r.t0) ++ord;
        const int pad = ord == runs_of[r.b] - 1 ? spu - runs_of[r.b] : 0;
        return ord | (pad << 16);
    };
    // table order = the order the persistent workgroups draw in: every worker's first run, then the runs that follow in
    // the order their workers come free (the shortest first runs first)
    items->clear();
    size_t depth = 0;
    for (const auto& w : workers) depth = std::max(depth, w.size());
    for (size_t d = 0; d < depth; ++d) {
        std::vector<std::pair<double, const GlRun*>> level;
        for (const auto& w : workers) {
            if (w.size() <= d) continue;
            double before = 0.0;
            for (size_t q = 0; q < d; ++q) before += w[q].len;
            level.push_back({before, &w[d]});
        }
        std::stable_sort(level.begin(), level.end(), [](const std::pair<double, const GlRun*>& a, const std::pair<double, const GlRun*>& b) { return a.first < b.first; });
        for (const auto& e : level) items->push_back(make_int4(e.second->b, e.second->t0, e.second->len, slot_word(*e.second)));
    }
    if (workers_out) {
        int nw = 0;
        for (const auto& w : workers) nw += !w.empty();
        *workers_out = nw;
    }
    if (slots_per_utt) *slots_per_utt = spu;
    return (int)items->size();
}

// The cut of a shape is made once per process and uploaded once per device (a table of 16 bytes per run).
hipError_t gl_plan_stream(GlParams& p, int n_workers, int n_stage, int force_runs, int force_run_len, hipStream_t stream) {
    // A table on a device: uploaded by an asynchronous copy on the stream of the call that first needs it (the host copy lives
    // in the cache, at a stable address) -- a cut made in the middle of a run of pipelined calls must not wait for the streams to
    // drain, as a synchronous copy does.  Calls on other streams wait for the `ready` event until it has been seen complete.
    struct Table { int4* ptr = nullptr; hipEvent_t ready = nullptr; hipStream_t owner = nullptr; bool seen = false; };
    struct Plan { std::vector<int4> items; int slots = 1, workers = 1; std::map<int, Table> dev; };
    struct Pool { char* base = nullptr; size_t used = 0, size = 0; };   // tables are carved from 1 MB blocks and never freed
    static std::map<std::vector<int>, Plan> cache;
    static std::map<int, Pool> pools;
    static std::mutex cache_mutex;
    std::lock_guard<std::mutex> lock(cache_mutex);
    // ONE cut for all launches of a call; ring_frames is set per launch (launch_gl_stream)
    p.ring_frames = gl_stream_ring_frames(p.win, p.hop, 1);
    const std::vector<int> key = {p.T, p.B, p.win, p.hop, n_workers, n_stage, force_runs, force_run_len};
    Plan& plan = cache[key];
    if (plan.items.empty())
        gl_plan_items(p.T, p.B, p.win, p.hop, n_workers, n_stage, force_runs, force_run_len, &plan.items, &plan.slots, &plan.workers);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    Table& t = plan.dev[dev];
    if (!t.ptr) {
        const size_t bytes = (plan.items.size() * sizeof(int4) + 255) & ~(size_t)255;
        Pool& pool = pools[dev];
        if (pool.used + bytes > pool.size) {
            const size_t block = std::max<size_t>(bytes, 1u << 20);
            void* m = nullptr;
            if ((e = hipMalloc(&m, block)) != hipSuccess) return e;
            pool.base = static_cast<char*>(m); pool.used = 0; pool.size = block;
        }
        int4* d = reinterpret_cast<int4*>(pool.base + pool.used);
        hipEvent_t ev = nullptr;
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(d, plan.items.data(), plan.items.size() * sizeof(int4), hipMemcpyHostToDevice, stream)) != hipSuccess ||
            (e = hipEventRecord(ev, stream)) != hipSuccess) {
            (void)hipEventDestroy(ev);
            return e;
        }
        pool.used += bytes;
        t.ptr = d; t.ready = ev; t.owner = stream; t.seen = false;
    } else if (!t.seen) {
        if (hipEventQuery(t.ready) == hipSuccess) t.seen = true;
        else if (stream != t.owner && (e = hipStreamWaitEvent(stream, t.ready, 0)) != hipSuccess) return e;
    }
    p.items = t.ptr;
    p.n_items = (int)plan.items.size();
    p.slots_per_utt = plan.slots;
    p.n_workers = plan.workers;
    return hipSuccess;
}

template <int MODE, int W, int H, bool MSE, int NST = 1>
static hipError_t gl_stream_set_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gl_stream_kernel<MODE, W, H, MSE, NST, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess || MODE != 0) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gl_stream_kernel<0, W, H, MSE, NST, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// The (window, hop) pairs the streaming kernel is instantiated for (both windows in registers, every span bound static): the
// model's 50 ms / 12.5 ms at the reference's 22.05 kHz (1102 / 275, dataset_params.sampling_rate; audio/conversion.py:122-136)
// and, round 6, the same 50 / 12.5 ms at 16 kHz (800 / 200).  Any other pair takes the general kernels of
// griffin_lim_generic.hip.  The run-time (WIN_CT = HOP_CT = 0) form of the kernel body stays in the source for the tools; it
// needed ~100 spilled registers per lane.
bool gl_stream_instantiated(int win, int hop) { return (win == 1102 && hop == 275) || (win == 800 && hop == 200); }

template <int W, int H>
static hipError_t gl_stream_launch_wh(hipStream_t s, const GlParams& p, dim3 grid, size_t lds, int final_istft, int n_stage, bool mse) {
#define GLS_LAUNCH_N(MODE, M, N)                                                                                         \
    {                                                                                                                    \
        if (MODE == 0 && p.seeded) hipLaunchKernelGGL((gl_stream_kernel<0, W, H, M, N, true>), grid, dim3(GL_THREADS), lds, s, p);   \
        else hipLaunchKernelGGL((gl_stream_kernel<MODE, W, H, M, N, false>), grid, dim3(GL_THREADS), lds, s, p);        \
    }
#ifdef GL_FAST_BUILD
    if (mse) return hipErrorInvalidValue;
    if (final_istft) GLS_LAUNCH_N(1, false, 1)
    else if (n_stage == 2) GLS_LAUNCH_N(0, false, 2)
    else if (n_stage == 3) GLS_LAUNCH_N(0, false, 3)
    else GLS_LAUNCH_N(0, false, 1)
#else
    if (n_stage == 3) GLS_LAUNCH_N(0, false, 3)
    else if (n_stage == 2) GLS_LAUNCH_N(0, false, 2)
    else if (final_istft) GLS_LAUNCH_N(1, false, 1)
    else if (mse) GLS_LAUNCH_N(0, true, 1)
    else GLS_LAUNCH_N(0, false, 1)
#endif
#undef GLS_LAUNCH_N
    return hipGetLastError();
}

// p.work_counter must point at a zeroed counter that no other launch uses; p planned by gl_plan_stream.
// n_stage = 2: two iterations in this launch (phase_in -> phase_out is then TWO Griffin-Lim iterations); not for the final
// iSTFT and not with the mse.
hipError_t launch_gl_stream(hipStream_t s, const GlParams& p_in, int n_cus, int final_istft, int n_stage) {
    GlParams p = p_in;
    p.n_stage = n_stage;
    p.ring_frames = gl_stream_ring_frames(p.win, p.hop, n_stage);
    if (p.ring_frames < GL_NW || n_stage < 1 || n_stage > 3 || (n_stage > 1 && (final_istft || p.mse_partial))) return hipErrorInvalidValue;
    const size_t lds = gl_stream_lds_bytes(p);
    // one workgroup per compute unit (256 registers x 8 waves), as many as the cut was made for
    const int nwg = p.n_workers > 0 && p.n_workers < n_cus ? p.n_workers : (p.n_items < n_cus ? p.n_items : n_cus);
    const dim3 grid(nwg);
    const bool mse = p.mse_partial != nullptr;
    if (p.win == 1102 && p.hop == 275) return gl_stream_launch_wh<1102, 275>(s, p, grid, lds, final_istft, n_stage, mse);
#ifndef GL_FAST_BUILD
    if (p.win == 800 && p.hop == 200) return gl_stream_launch_wh<800, 200>(s, p, grid, lds, final_istft, n_stage, mse);
#endif
    return hipErrorInvalidValue;
}

template <int W, int H>
static hipError_t gl_stream_configure_wh() {
    hipError_t e;
#ifndef GL_FAST_BUILD
    if ((e = gl_stream_set_attr<0, W, H, true>()) != hipSuccess) return e;
#endif
    if ((e = gl_stream_set_attr<0, W, H, false>()) != hipSuccess) return e;
    if ((e = gl_stream_set_attr<0, W, H, false, 2>()) != hipSuccess) return e;
    if ((e = gl_stream_set_attr<0, W, H, false, 3>()) != hipSuccess) return e;
    if ((e = gl_stream_set_attr<1, W, H, false>()) != hipSuccess) return e;
    return hipSuccess;
}
hipError_t gl_stream_configure() {
    hipError_t e = gl_stream_configure_wh<1102, 275>();
#ifndef GL_FAST_BUILD
    if (e == hipSuccess) e = gl_stream_configure_wh<800, 200>();
#endif
    return e;
}

void gl_build_wlane(const float* window, const float* rwss, int win, int hop, int T, float* out) {
    const int wpad = (NFFT - win) / 2;
    const int halo = (win + hop - 1) / hop - 1;
    const int t_ref = halo < T ? halo : T - 1;   // an interior frame (all `halo` neighbours either side exist) if there is one
    for (int c = 0; c < 16; ++c)
        for (int e = 0; e < 2; ++e)
            for (int lane = 0; lane < 64; ++lane) {
                const int nw = 2 * (lane + 64 * c) + e - wpad;
                const bool in = nw >= 0 && nw < win;
                const float w = in ? window[nw] * (0.5f / (float)MH) : 0.f;
                const float rw = in ? rwss[(size_t)t_ref * hop + wpad + nw] : 0.f;
                out[(0 * 64 + lane) * 32 + 2 * c + e] = w;
                out[(1 * 64 + lane) * 32 + 2 * c + e] = w * rw;
            }
}

static hipError_t stft_configure();

// Function attributes are per device: called once per handle (on the handle's device) by api_stages.hip.
hipError_t gl_configure() {
    const hipError_t e = gl_stream_configure();
    return e != hipSuccess ? e : stft_configure();
}

// ------------------------------------------------------------------------------------ analysis STFT
// librosa.stft(center=True, reflect, hann) of real signals: one wave per frame, signal read straight
// from global memory (this is the feature-extraction side, reference audio/features.py:62,145 --
// not part of the synthesis loop).  out [B][Tf][FP] complex.
__global__ __launch_bounds__(GL_THREADS) void stft_kernel(const float* __restrict__ wav, int n, int Tf,
                                                          const float* __restrict__ window, int win, int hop,
                                                          const cf* __restrict__ tw1024, const cf* __restrict__ tw2048,
                                                          cf* __restrict__ out, int FP) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* ex_all = reinterpret_cast<cf*>(smem_raw);
    cf* twR = ex_all + GL_NW * EX_CPLX;
    cf* twA = twR + 1024;
    float* wtab = reinterpret_cast<float*>(twA + 15 * 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cf* ex = ex_all + wave * EX_CPLX;
    for (int i = tid; i < win; i += GL_THREADS) wtab[i] = window[i];
    for (int i = tid; i < 1024; i += GL_THREADS) twR[i] = tw2048[i];
    for (int i = tid; i < 15 * 64; i += GL_THREADS) twA[i] = tw1024[(i & 63) * ((i >> 6) + 1)];
    FftTw tw;
#pragma unroll
    for (int d = 1; d < 4; ++d) tw.b[d - 1] = tw1024[16 * (lane & 15) * d];
    tw.a = twA + lane;
    __syncthreads();
    const int b = blockIdx.y;
    const int t = blockIdx.x * GL_NW + wave;
    if (t >= Tf) return;
    const float* y = wav + (size_t)b * n;
    const int wpad = (NFFT - win) >> 1;
    const int ylo = t * hop + wpad - MH;
    cf v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int nn = 2 * (lane + 64 * j);
        float x[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int nw = nn + e - wpad;
            if (nw >= 0 && nw < win) {
                int yi = ylo + nw;
                yi = yi < 0 ? -yi : yi;
                yi = yi >= n ? 2 * (n - 1) - yi : yi;
                x[e] = wtab[nw] * y[yi];
            }
        }
        v[j] = cmk(x[0], x[1]);
    }
    fft1024(v, ex, tw, lane);
#pragma unroll
    for (int c = 0; c < 16; ++c) ex[lane + 64 * c] = v[c];
    wave_lds_sync();
    cf* orow = out + ((size_t)b * Tf + t) * FP;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int k = lane + 64 * c;
        const cf zk = v[c];
        const cf zr = ex[(MH - k) & (MH - 1)];
        const cf e = cscale(cadd_conj(zk, zr), 0.5f);
        const cf o = cmul(cscale(csub_conj(zk, zr), 0.5f), twR[k]);
        orow[k] = cadd_mi(e, o);
    }
    if (lane == 0) orow[MH] = cmk(v[0].x - v[0].y, 0.f);
    if (lane < FP - MH - 1) orow[MH + 1 + lane] = cmk(0.f, 0.f);
}

static hipError_t stft_configure() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024);
}

hipError_t launch_stft(hipStream_t s, const float* wav, int B, int n, int Tf, const float* window, int win, int hop,
                       const float2* tw1024, const float2* tw2048, float2* out, int FP) {
    const size_t lds = (size_t)(GL_NW * EX_CPLX + 1024 + 15 * 64) * sizeof(cf) + (size_t)((win + 3) & ~3) * sizeof(float);
    dim3 grid((Tf + GL_NW - 1) / GL_NW, B);
    hipLaunchKernelGGL(stft_kernel, grid, dim3(GL_THREADS), lds, s, wav, n, Tf, window, win, hop,
                       reinterpret_cast<const cf*>(tw1024), reinterpret_cast<const cf*>(tw2048), reinterpret_cast<cf*>(out), FP);
    return hipGetLastError();
}

// complex (B,T,FP) -> (B,F,T): mode 0 interleaved complex64, mode 1 |z| ** power
__global__ void cplx_tf_to_ft_kernel(const cf* in, float* out, int F, int T, int FP, int mode, float power) {
    __shared__ cf tile[32][33];
    const int b = blockIdx.z;
    const int f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, f = f0 + tx;
        tile[i][tx] = (t < T && f < F) ? in[((size_t)b * T + t) * FP + f] : cmk(0.f, 0.f);
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int f = f0 + i, t = t0 + tx;
        if (f < F && t < T) {
            const cf z = tile[tx][i];
            const size_t o = ((size_t)b * F + f) * T + t;
            if (mode == 0) {
                reinterpret_cast<cf*>(out)[o] = z;
            } else {
                const float m = sqrtf(z.x * z.x + z.y * z.y);
                out[o] = power == 1.0f ? m : (power == 2.0f ? m * m : powf(m, power));
            }
        }
    }
}
hipError_t launch_cplx_tf_to_ft(hipStream_t s, const float2* in, float* out, int B, int F, int T, int FP, int mode,
                                float power) {
    dim3 grid((T + 31) / 32, (F + 31) / 32, B);
    hipLaunchKernelGGL(cplx_tf_to_ft_kernel, grid, dim3(32, 8), 0, s, reinterpret_cast<const cf*>(in), out, F, T, FP, mode, power);
    return hipGetLastError();
}

// elementwise dB conversions of reference audio/conversion.py: mode 0 magnitude_to_decibel (:5-29),
// 1 decibel_to_magnitude (:32-53), 2 normalize_decibel (:56-78), 3 inv_normalize_decibel (:81-102)
__global__ void db_convert_kernel(const float* in, float* out, size_t n, int mode, float ref_db, float max_db) {
    const float range = fabsf(ref_db) + fabsf(max_db);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = in[i];
        float y;
        if (mode == 0) y = (float)(20.0 * log10((double)fmaxf(1e-5f, x)));   // analysis side: exact rounding
        else if (mode == 1) y = exp2f(x * (0.05f * 3.3219280948873623f));
        else if (mode == 2) y = fminf(fmaxf(1.0f + (x - ref_db) / range, 0.f), 1.f);
        else y = (fminf(fmaxf(x, 0.f), 1.f) - 1.0f) * range + ref_db;
        out[i] = y;
    }
}
hipError_t launch_db_convert(hipStream_t s, const float* in, float* out, size_t n, int mode, float ref_db, float max_db) {
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(db_convert_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, in, out, n, mode, ref_db, max_db);
    return hipGetLastError();
}

// any value below lim?  (decibel_to_magnitude's assertion, reference audio/conversion.py:47-49)
__global__ void any_below_kernel(const float* in, size_t n, float lim, int* flag) {
    bool hit = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        hit |= in[i] < lim;
    if (hit) *flag = 1;
}
hipError_t launch_any_below(hipStream_t s, const float* in, size_t n, float lim, int* flag) {
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(any_below_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, in, n, lim, flag);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ small kernels
// mse[b] = sum(partials[b][:]) / (F*T), fixed order.
__global__ void gl_mse_reduce_kernel(const float* partial, int nchunks, float denom, float* mse) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < nchunks; ++i) s += partial[(size_t)b * nchunks + i];
        mse[b] = s / denom;
    }
}
hipError_t launch_gl_mse_reduce(hipStream_t s, const float* partial, int B, int nchunks, float denom, float* mse) {
    hipLaunchKernelGGL(gl_mse_reduce_kernel, dim3(B), dim3(64), 0, s, partial, nchunks, denom, mse);
    return hipGetLastError();
}

// (B,F,T) reference layout -> internal (B,T,FP) magnitude (abs taken, as griffin_lim_v2 does)
__global__ void mag_ft_to_tf_kernel(const float* in, float* out, int F, int T, int FP) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int f = f0 + i, t = t0 + tx;
        tile[i][tx] = (f < F && t < T) ? fabsf(in[((size_t)b * F + f) * T + t]) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, f = f0 + tx;
        if (t < T && f < FP) out[((size_t)b * T + t) * FP + f] = tile[tx][i];
    }
}
hipError_t launch_mag_ft_to_tf(hipStream_t s, const float* in, float* out, int B, int F, int T, int FP) {
    dim3 grid((T + 31) / 32, (FP + 31) / 32, B);
    hipLaunchKernelGGL(mag_ft_to_tf_kernel, grid, dim3(32, 8), 0, s, in, out, F, T, FP);
    return hipGetLastError();
}

// internal (B,T,FP) -> (B,F,T)
__global__ void tf_to_ft_kernel(const float* in, float* out, int F, int T, int FP) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, f = f0 + tx;
        tile[i][tx] = (t < T && f < F) ? in[((size_t)b * T + t) * FP + f] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int f = f0 + i, t = t0 + tx;
        if (f < F && t < T) out[((size_t)b * F + f) * T + t] = tile[tx][i];
    }
}
hipError_t launch_tf_to_ft(hipStream_t s, const float* in, float* out, int B, int F, int T, int FP) {
    dim3 grid((T + 31) / 32, (F + 31) / 32, B);
    hipLaunchKernelGGL(tf_to_ft_kernel, grid, dim3(32, 8), 0, s, in, out, F, T, FP);
    return hipGetLastError();
}

// angles = exp(2 pi i u): u from init (B,F,T) reference layout, or the seeded start (gl_seed_phasor: only needed as codes when
// no iteration follows -- the first launch of an iteration makes it itself); out (B,T,FP) phasor codes (no magnitudes
// needed: the state is the phasor alone).  Running it on a side stream beside the post-net was tried and cost 1.3 ms
// per step instead of saving 0.17: a third busy stream slows the Griffin-Lim launches of the main one.
__global__ void phase_init_kernel(const float* init_ft, uint64_t seed, gl_state_t* out, int F, int T, int FP) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;
    for (int i = ty; i < 32; i += 8) {
        const int f = f0 + i, t = t0 + tx;
        float u = 0.f;
        if (f < F && t < T) {
            const size_t idx = ((size_t)b * F + f) * T + t;
            u = init_ft ? init_ft[idx] : 0.f;
        }
        tile[i][tx] = u;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, f = f0 + tx;
        if (t < T && f < FP) {
            float sn, cs;
            sincospif(2.0f * tile[tx][i], &sn, &cs);
            cf e = cmk(cs, sn);   // phasor of exp(2 pi i u)
            if (!init_ft) e = f < F ? gl_seed_phasor(seed, ((unsigned long long)b * F + f) * T + t) : cmk(1.f, 0.f);   // the seeded start
            out[((size_t)b * T + t) * FP + f] = gl_state_encode(e);
        }
    }
}
hipError_t launch_phase_init(hipStream_t s, const float* init_ft, uint64_t seed, void* out, int B, int F, int T, int FP) {
    dim3 grid((T + 31) / 32, (FP + 31) / 32, B);
    hipLaunchKernelGGL(phase_init_kernel, grid, dim3(32, 8), 0, s, init_ft, seed, reinterpret_cast<gl_state_t*>(out), F, T, FP);
    return hipGetLastError();
}

// linear (B,T,F) network output -> internal magnitude (B,T,FP):
//   db = (clip(x,0,1) - 1) * (|ref| + |max|) + ref;  mag = 10^(db/20);  mag ** power
// below_flag (optional): set when some value de-normalises to less than -100 dB (decibel_to_magnitude's
// assertion, reference audio/conversion.py:47-49)
__global__ void denorm_power_kernel(const float* lin, float* mag, size_t rows, int F, int FP, float ref_db,
                                    float range_db, float power, int* below_flag) {
    const size_t row = blockIdx.x;
    if (row >= rows) return;
    const float* in = lin + row * F;
    float* out = mag + row * FP;
    bool below = false;
    for (int f = threadIdx.x; f < FP; f += blockDim.x) {
        float v = 0.f;
        if (f < F) {
            const float db = denorm_db(in[f], ref_db, range_db);
            below |= db < -100.0f;
            v = db_pow(db, power);
        }
        out[f] = v;
    }
    if (below_flag && below) *below_flag = 1;
}
hipError_t launch_denorm_power(hipStream_t s, const float* lin, float* mag, size_t rows, int F, int FP,
                               float ref_db, float max_db, float power, int* below_flag) {
    hipLaunchKernelGGL(denorm_power_kernel, dim3((unsigned)rows), dim3(256), 0, s, lin, mag, rows, F, FP, ref_db,
                       fabsf(ref_db) + fabsf(max_db), power, below_flag);
    return hipGetLastError();
}

// peak normalisation: wav /= max|wav| per utterance unless the peak is below FLT_MIN
__global__ __launch_bounds__(1024) void peak_normalize_kernel(float* wav, int n) {
    __shared__ float red[16];
    float* w = wav + (size_t)blockIdx.x * n;
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
    if (m < 1.17549435e-38f) m = 1.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) w[i] = w[i] / m;
}
// wav /= max(partials[b][:]) (unless below FLT_MIN): second half of the fused peak normalisation
__global__ void peak_scale_kernel(float* wav, int n, const float* partial, int nparts) {
    const int b = blockIdx.y;
    float m = 0.f;
    for (int i = 0; i < nparts; ++i) m = fmaxf(m, partial[(size_t)b * nparts + i]);
    if (m < 1.17549435e-38f) m = 1.0f;
    float* w = wav + (size_t)b * n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) w[i] = w[i] / m;
}
hipError_t launch_peak_scale(hipStream_t s, float* wav, int B, int n, const float* partial, int nparts) {
    hipLaunchKernelGGL(peak_scale_kernel, dim3(64, B), dim3(256), 0, s, wav, n, partial, nparts);
    return hipGetLastError();
}

hipError_t launch_peak_normalize(hipStream_t s, float* wav, int B, int n) {
    hipLaunchKernelGGL(peak_normalize_kernel, dim3(B), dim3(1024), 0, s, wav, n);
    return hipGetLastError();
}

#ifdef GL_CLOCK
extern "C" void gl_clock_dump() {
    static unsigned long long host[4096][2];
    unsigned n = 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(gl_clock_n), sizeof(n));
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(gl_clock_log), sizeof(host));
    if (n > 4096) n = 4096;
    for (unsigned i = 0; i < n; ++i)
        printf("launch %4u: workgroup 0 ran %.1f us at %.0f MHz\n", i, (double)host[i][1] * 0.01, (double)host[i][0] * 100.0 / (double)host[i][1]);
}
#endif
}  // namespace tts
