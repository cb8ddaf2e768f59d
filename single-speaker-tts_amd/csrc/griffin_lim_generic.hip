// Griffin-Lim, STFT and iSTFT for every power-of-two n_fft from 256 to 4096 and any window / hop pair (gfx950).
//
// The reference passes n_fft, win_length and hop_length as arguments everywhere on its audio surface
// (audio/synthesis.py:5-40, 43-125; audio/features.py:5-86, 116-145); griffin_lim.hip is specialised to the model's
// configuration (n_fft 2048, 1102 / 275: one wave per transform, both windows in registers, the overlap-add in an LDS
// ring).  This file is the general form behind the same C entry points: same arithmetic per frame, nothing assumed
// about the sizes beyond n_fft = 2^m, and deliberately simple --
//   glg_istft_kernel   one workgroup per frame: X = |S| e^{i phi} (Hermitian extension, imaginary parts of the DC and
//                      Nyquist bins ignored as numpy.fft.irfft does) -> n_fft-point inverse FFT in LDS -> synthesis window
//                      -> the frame's win_length windowed samples to HBM          (librosa.istft, synthesis.py:91-105)
//   glg_ola_kernel     overlap-add as a GATHER: every output sample adds the <= ceil(win / hop) frames that cover it, in
//                      frame order (fixed summation order: bit-reproducible), times 1 / window-sum-square, trimmed by
//                      n_fft / 2 at either end
//   glg_stft_kernel    one workgroup per frame: reflect-padded, windowed samples -> FFT -> the new unit phasors (or the
//                      complex spectrum for tts_stft), optionally the frame's squared magnitude error
//                                                                                 (librosa.stft, synthesis.py:108-112)
// The state between iterations is a float2 unit phasor per bin.  FFT (round 6): the real n_fft-point transforms run as
// n_fft / 2-point COMPLEX transforms of (even, odd) sample pairs with a split / merge pass (as in griffin_lim.hip), in place in LDS
// on bit-reversed input, two radix-2 stages per pass (a radix-4 butterfly in registers: half the passes, barriers and LDS
// traffic of the radix-2 form), twiddles from a table computed in double precision on the host; a frame's workgroup is a quarter
// of the half-size transform's points (64 ... 256 threads: one radix-4 group per thread and pass).  4.4 x less LDS traffic per
// frame than round 4's full-size radix-2 transform, which bounded these kernels.  Measured per iteration at B = 64, T = 1000
// (tools/gl_generic_bench.py): n_fft 4096 / 2400 / 600 5365 -> 1909 us, 2048 / 1200 / 300 2156 -> 928, 1024 / 800 / 200
// 1012 -> 524, 512 / 400 / 100 481 -> 251 (the streaming kernel at 2048: 239).  What is left is mostly HBM: a float2 phasor per
// bin written and read, and the windowed frames' round trip -- 30 KB per frame and iteration at 2048 / 1200 / 300 against the
// streaming kernel's 8 (32-bit phasor codes, overlap-add in LDS).  Still one workgroup per frame: the configurations the path
// is measured on run in griffin_lim.hip.
#include "tts_common.h"
#include "griffin_lim.h"

namespace tts {

#define GLG_THREADS 256

// A complex number as two SCALAR floats, and the file is built with -fno-slp-vectorize (build.py): no v_pk_*_f32 instruction is
// selected for these kernels.  Round 6 measured why (profiles/r06_experiment_packed_f32_beside_mfma.txt): written with a float2
// vector type the compiler made the butterflies from packed-f32 VOP3P instructions, and whenever waves of the MFMA GEMM
// (gemm_f32_kernel, another stream) shared the compute unit, single frames came out wrong -- the low dword of a packed result in
// lanes 48-63 of one wave, as stored to LDS by the next instruction: 129 of 200 calls beside GEMM launches, 0 of 3100 for the
// same source without packed selection (and 0 of 120 for every other stage of the library under the same neighbour).  The
// kernels are bound by memory: the scalar form costs nothing (926 / 523 us per iteration at n_fft 2048 / 1024 either way).
struct __attribute__((aligned(8))) gcf { float x, y; };
__device__ __forceinline__ gcf operator+(gcf a, gcf b) { return gcf{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ gcf operator-(gcf a, gcf b) { return gcf{a.x - b.x, a.y - b.y}; }

__device__ __forceinline__ gcf gmul(gcf a, gcf b) { return (gcf){a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

__device__ __forceinline__ gcf gconj(gcf a) { return (gcf){a.x, -a.y}; }
__device__ __forceinline__ gcf gmul_i(gcf a) { return (gcf){-a.y, a.x}; }     // i a
__device__ __forceinline__ gcf gmul_mi(gcf a) { return (gcf){a.y, -a.x}; }    // -i a

// In-place decimation-in-time FFT of `a`: M = 1 << mm complex values in LDS, ALREADY in bit-reversed order.  tw[k] =
// exp(-2 pi i k / (2 M)), k < M (the table of the real transform of 2 M points): W_M^j = tw[2 j].  INVERSE conjugates the
// twiddles (no 1 / M scale).  Two radix-2 stages per pass: the four values (base + q half, q < 4) of a radix-4 group go through
// stage s (pairs (0,1), (2,3), one twiddle) and stage s + 1 (pairs (0,2), (1,3), twiddles w and -i w) in registers; a last single
// stage when mm is odd.
template <bool INVERSE>
__device__ __forceinline__ void glg_fft(gcf* a, const gcf* __restrict__ tw, int M, int mm) {
    int s = 1;
    for (; s + 1 <= mm; s += 2) {
        const int half = 1 << (s - 1);
        __syncthreads();
        for (int i = threadIdx.x; i < M / 4; i += blockDim.x) {
            const int j = i & (half - 1);
            const int base = ((i - j) << 2) + j;
            gcf w1 = tw[(j << (mm - s)) << 1];          // stage s:     W_M^(j M / 2^s)
            gcf w2 = tw[(j << (mm - s - 1)) << 1];      // stage s + 1: W_M^(j M / 2^(s+1)); its partner at j + half is -i (forward) times that
            if (INVERSE) { w1.y = -w1.y; w2.y = -w2.y; }
            const gcf x0 = a[base], x1 = gmul(a[base + half], w1), x2 = a[base + 2 * half], x3 = gmul(a[base + 3 * half], w1);
            const gcf u0 = x0 + x1, u1 = x0 - x1, u2 = gmul(x2 + x3, w2);
            const gcf t3 = gmul(x2 - x3, w2);
            const gcf u3 = INVERSE ? gmul_i(t3) : gmul_mi(t3);
            a[base] = u0 + u2;
            a[base + half] = u1 + u3;
            a[base + 2 * half] = u0 - u2;
            a[base + 3 * half] = u1 - u3;
        }
    }
    if (s == mm) {   // one radix-2 stage left
        const int half = 1 << (s - 1);
        __syncthreads();
        for (int i = threadIdx.x; i < M / 2; i += blockDim.x) {
            const int j = i & (half - 1);
            const int base = ((i - j) << 1) + j;
            gcf w = tw[(j << (mm - s)) << 1];
            if (INVERSE) w.y = -w.y;
            const gcf u = a[base], v = gmul(a[base + half], w);
            a[base] = u + v;
            a[base + half] = u - v;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int glg_bitrev(int i, int m) { return (int)(__brev((unsigned)i) >> (32 - m)); }

// ------------------------------------------------------------------------------------------------ inverse transform
// frames[b][t][j] = window[j] * irfft(|S| e^{i phi})[pad + j], j < win  (pad = (N - win) / 2)
__global__ __launch_bounds__(GLG_THREADS) void glg_istft_kernel(const float* __restrict__ mag, const gcf* __restrict__ ph,
                                                                const float* __restrict__ window, const gcf* __restrict__ tw,
                                                                float* __restrict__ frames, int T, int Fp, int N, int m, int win) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gcf* a = reinterpret_cast<gcf*>(smem);
    const int t = blockIdx.x, b = blockIdx.y;
    const size_t row = ((size_t)b * T + t) * Fp;
    // real inverse transform of N = 2 M points as a complex one of M: with z[n] = x[2n] + i x[2n+1],
    //   Z[k] = (X[k] + conj X[M-k]) + i conj(W_N^k) (X[k] - conj X[M-k])   (= 2 FFT_M(z)[k]),   x = IFFT_M(Z) / N
    const int M = N >> 1, mm = m - 1;
    for (int k = threadIdx.x; k < M; k += blockDim.x) {
        const float s0 = mag[row + k], s1 = mag[row + M - k];
        const gcf e0 = ph[row + k], e1 = ph[row + M - k];
        const gcf twk = tw[k];
        gcf xk = (gcf){s0 * e0.x, s0 * e0.y}, xm = (gcf){s1 * e1.x, s1 * e1.y};
        if (k == 0) { xk.y = 0.f; xm.y = 0.f; }   // irfft ignores the imaginary parts of the DC and Nyquist bins
        const gcf cm = gconj(xm);
        const gcf d = gmul(gconj(twk), xk - cm);
        a[glg_bitrev(k, mm)] = (xk + cm) + gmul_i(d);
    }
    glg_fft<true>(a, tw, M, mm);
    const int pad = (N - win) >> 1;
    const float inv = 1.0f / (float)N;
    float* out = frames + ((size_t)b * T + t) * win;
    for (int j = threadIdx.x; j < win; j += blockDim.x) {
        const int n = pad + j;
        const gcf z = a[n >> 1];
        out[j] = window[j] * (((n & 1) ? z.y : z.x) * inv);
    }
}

// wav[b][s] = rwss[s + N/2] * sum_t frames[b][t][s + N/2 - t hop - pad], frames in increasing t; s < L = hop (T - 1)
__global__ void glg_ola_kernel(const float* __restrict__ frames, const float* __restrict__ rwss, float* __restrict__ wav, int T, int N,
                               int win, int hop, int L) {
    const int b = blockIdx.y;
    const int pad = (N - win) >> 1;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < L; s += gridDim.x * blockDim.x) {
        const int n = s + (N >> 1);           // index in the padded signal
        const int q = n - pad;                // frame t covers it when 0 <= q - t hop < win
        int t_lo = (q - win + hop) / hop;     // ceil((q - win + 1) / hop)
        if (q - win + 1 <= 0) t_lo = 0;
        int t_hi = q / hop;
        if (t_hi > T - 1) t_hi = T - 1;
        float acc = 0.f;
        for (int t = t_lo; t <= t_hi; ++t) acc += frames[((size_t)b * T + t) * win + (q - t * hop)];
        wav[(size_t)b * L + s] = acc * rwss[n];
    }
}

// ------------------------------------------------------------------------------------------------ forward transform
// MODE 0: out_ph = unit phasors of the spectrum (the next iteration's estimate), optional mse partials (one per frame);
// MODE 1: out_z = the complex spectrum itself (tts_stft)
template <int MODE>
__global__ __launch_bounds__(GLG_THREADS) void glg_stft_kernel(const float* __restrict__ wav, int n, const float* __restrict__ window,
                                                               const gcf* __restrict__ tw, gcf* __restrict__ out, int Tf, int Fp, int N,
                                                               int m, int win, int hop, const float* __restrict__ mag,
                                                               float* __restrict__ mse_partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gcf* a = reinterpret_cast<gcf*>(smem);
    __shared__ float red[GLG_THREADS / 64];
    const int t = blockIdx.x, b = blockIdx.y;
    const int H = N >> 1, pad = (N - win) >> 1;
    const float* y = wav + (size_t)b * n;
    const int y0 = t * hop - H;               // signal index of padded-frame sample 0
    // real transform of N = 2 M points as a complex one of M: z[n] = x[2n] + i x[2n+1], Z = FFT_M(z),
    //   X[k] = ((Z[k] + conj Z[M-k]) - i W_N^k (Z[k] - conj Z[M-k])) / 2,  k <= M  (Z[M] = Z[0])
    const int M = N >> 1, mm = m - 1;
    auto sample = [&](int j) -> float {
        const int jw = j - pad;
        if (jw < 0 || jw >= win) return 0.f;
        int yi = y0 + j;
        yi = yi < 0 ? -yi : yi;                           // reflect (numpy.pad mode='reflect')
        yi = yi >= n ? 2 * (n - 1) - yi : yi;
        return window[jw] * y[yi];
    };
    for (int q = threadIdx.x; q < M; q += blockDim.x) a[glg_bitrev(q, mm)] = (gcf){sample(2 * q), sample(2 * q + 1)};
    glg_fft<false>(a, tw, M, mm);
    const size_t row = ((size_t)b * Tf + t) * Fp;
    float err = 0.f;
    for (int k = threadIdx.x; k < Fp; k += blockDim.x) {
        gcf z = (gcf){0.f, 0.f};
        if (k <= H) {
            const gcf zk = a[k & (M - 1)], zm = gconj(a[(M - k) & (M - 1)]);
            if (k == 0) z = (gcf){zk.x + zk.y, 0.f};
            else if (k == M) z = (gcf){zk.x - zk.y, 0.f};
            else {
                const gcf e = zk + zm, d = gmul(tw[k], zk - zm);
                z = (gcf){0.5f * (e.x + d.y), 0.5f * (e.y - d.x)};   // (e - i d) / 2
            }
        }
        if (MODE == 1) {
            out[row + k] = z;
        } else {
            const float s2 = z.x * z.x + z.y * z.y;
            const float r = rsqrtf(s2);
            // numpy: exp(1j * angle(0)) = 1
            out[row + k] = (k <= H && s2 > 1.0e-37f) ? (gcf){z.x * r, z.y * r} : (gcf){1.f, 0.f};
            if (mse_partial && k <= H) {
                const float d = fabsf(mag[row + k]) - sqrtf(s2);
                err += d * d;
            }
        }
    }
    if (MODE == 0 && mse_partial) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) err += __shfl_xor(err, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = err;
        __syncthreads();
        if (threadIdx.x == 0) {   // (fixed order over the workgroup's waves: 1, 2 or 4)
            float e = red[0];
            for (int w = 1; w < (int)(blockDim.x >> 6); ++w) e += red[w];
            mse_partial[(size_t)b * Tf + t] = e;
        }
    }
}

// initial unit phasors: exp(2 pi i u) from a (B, F, T) array of U[0,1) numbers, or a counter-based draw from the seed
__global__ void glg_phase_init_kernel(const float* __restrict__ init_ft, unsigned long long seed, gcf* __restrict__ out, int F, int T, int Fp) {
    const int b = blockIdx.z, t = blockIdx.y;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < Fp; f += gridDim.x * blockDim.x) {
        gcf e = (gcf){1.f, 0.f};
        if (f < F) {
            const unsigned long long idx = ((unsigned long long)b * F + f) * T + t;
            float u;
            if (init_ft) {
                u = init_ft[idx];
            } else {
                unsigned x = (unsigned)idx ^ ((unsigned)(idx >> 32) * 0x9E3779B9u) ^ (unsigned)seed ^ ((unsigned)(seed >> 32) * 0x85EBCA6Bu);
                x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
                u = (float)(x >> 8) * (1.0f / 16777216.0f);
            }
            float sn, cs;
            sincospif(2.0f * u, &sn, &cs);
            e = (gcf){cs, sn};
        }
        out[((size_t)b * T + t) * Fp + f] = e;
    }
}

// threads per frame: a quarter of the half-size transform's points (one radix-4 group each), 64 ... 256
static int glg_threads(int N) { const int t = N / 8; return t < 64 ? 64 : (t > GLG_THREADS ? GLG_THREADS : t); }
static size_t glg_lds(int N) { return (size_t)(N / 2) * sizeof(gcf); }   // the half-size complex transform of a real frame

hipError_t glg_configure() {
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&glg_istft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&glg_stft_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)) != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&glg_stft_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
}

static int glg_log2(int N) {
    int m = 0;
    while ((1 << m) < N) ++m;
    return m;
}

bool glg_supports(int n_fft) { return n_fft >= 256 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0; }

hipError_t launch_glg_phase_init(hipStream_t s, const float* init_ft, uint64_t seed, float2* out, int B, int F, int T, int Fp) {
    hipLaunchKernelGGL(glg_phase_init_kernel, dim3((Fp + 255) / 256, T, B), dim3(256), 0, s, init_ft, (unsigned long long)seed,
                       reinterpret_cast<gcf*>(out), F, T, Fp);
    return hipGetLastError();
}

hipError_t launch_glg_istft(hipStream_t s, const float* mag, const float2* ph, const float* window, const float* rwss, const float2* tw,
                            float* frames, float* wav, int B, int T, int Fp, int n_fft, int win, int hop) {
    const int m = glg_log2(n_fft);
    hipLaunchKernelGGL(glg_istft_kernel, dim3(T, B), dim3(glg_threads(n_fft)), glg_lds(n_fft), s, mag, reinterpret_cast<const gcf*>(ph), window,
                       reinterpret_cast<const gcf*>(tw), frames, T, Fp, n_fft, m, win);
    const int L = hop * (T - 1);
    hipLaunchKernelGGL(glg_ola_kernel, dim3((L + 255) / 256 > 1024 ? 1024 : (L + 255) / 256, B), dim3(256), 0, s, frames, rwss, wav, T, n_fft,
                       win, hop, L);
    return hipGetLastError();
}

hipError_t launch_glg_stft(hipStream_t s, const float* wav, int n, const float* window, const float2* tw, float2* out, int B, int Tf, int Fp,
                           int n_fft, int win, int hop, int mode, const float* mag, float* mse_partial) {
    const int m = glg_log2(n_fft);
    if (mode == 1)
        hipLaunchKernelGGL((glg_stft_kernel<1>), dim3(Tf, B), dim3(glg_threads(n_fft)), glg_lds(n_fft), s, wav, n, window,
                           reinterpret_cast<const gcf*>(tw), reinterpret_cast<gcf*>(out), Tf, Fp, n_fft, m, win, hop, mag, mse_partial);
    else
        hipLaunchKernelGGL((glg_stft_kernel<0>), dim3(Tf, B), dim3(glg_threads(n_fft)), glg_lds(n_fft), s, wav, n, window,
                           reinterpret_cast<const gcf*>(tw), reinterpret_cast<gcf*>(out), Tf, Fp, n_fft, m, win, hop, mag, mse_partial);
    return hipGetLastError();
}

}  // namespace tts
