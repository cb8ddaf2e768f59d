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
// The state between iterations is a float2 unit phasor per bin.  FFT: iterative radix-2 on bit-reversed input, twiddles
// from a table computed in double precision on the host.  Not tuned: bound by HBM (the windowed frames make a round trip)
// and by one barrier per FFT stage; the configuration the path is measured on runs in griffin_lim.hip.
#include "tts_common.h"
#include "griffin_lim.h"

namespace tts {

#define GLG_THREADS 256

typedef float gcf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ gcf gmul(gcf a, gcf b) { return (gcf){a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// in-place radix-2 decimation-in-time FFT of `a` (N = 1 << m complex values in LDS, ALREADY in bit-reversed order);
// tw[k] = exp(-2 pi i k / N), k < N / 2; INVERSE conjugates the twiddles (no 1/N scale)
template <bool INVERSE>
__device__ __forceinline__ void glg_fft(gcf* a, const gcf* __restrict__ tw, int N, int m) {
    for (int s = 1; s <= m; ++s) {
        const int half = 1 << (s - 1);
        __syncthreads();
        for (int i = threadIdx.x; i < N / 2; i += GLG_THREADS) {
            const int j = i & (half - 1);
            const int base = ((i - j) << 1) + j;
            gcf w = tw[j << (m - s)];
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
    const int H = N >> 1;
    for (int k = threadIdx.x; k <= H; k += GLG_THREADS) {
        const float s = mag[row + k];
        const gcf e = ph[row + k];
        gcf x = (gcf){s * e.x, s * e.y};
        if (k == 0 || k == H) x.y = 0.f;   // irfft ignores them
        a[glg_bitrev(k, m)] = x;
        if (k > 0 && k < H) a[glg_bitrev(N - k, m)] = (gcf){x.x, -x.y};
    }
    glg_fft<true>(a, tw, N, m);
    const int pad = (N - win) >> 1;
    const float inv = 1.0f / (float)N;
    float* out = frames + ((size_t)b * T + t) * win;
    for (int j = threadIdx.x; j < win; j += GLG_THREADS) out[j] = window[j] * (a[pad + j].x * inv);
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
    for (int j = threadIdx.x; j < N; j += GLG_THREADS) {
        float x = 0.f;
        const int jw = j - pad;
        if (jw >= 0 && jw < win) {
            int yi = y0 + j;
            yi = yi < 0 ? -yi : yi;                       // reflect (numpy.pad mode='reflect')
            yi = yi >= n ? 2 * (n - 1) - yi : yi;
            x = window[jw] * y[yi];
        }
        a[glg_bitrev(j, m)] = (gcf){x, 0.f};
    }
    glg_fft<false>(a, tw, N, m);
    const size_t row = ((size_t)b * Tf + t) * Fp;
    float err = 0.f;
    for (int k = threadIdx.x; k < Fp; k += GLG_THREADS) {
        gcf z = (gcf){0.f, 0.f};
        if (k <= H) z = a[k];
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
        if (threadIdx.x == 0) mse_partial[(size_t)b * Tf + t] = (red[0] + red[1]) + (red[2] + red[3]);
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

static size_t glg_lds(int N) { return (size_t)N * sizeof(gcf); }

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
    hipLaunchKernelGGL(glg_istft_kernel, dim3(T, B), dim3(GLG_THREADS), glg_lds(n_fft), s, mag, reinterpret_cast<const gcf*>(ph), window,
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
        hipLaunchKernelGGL((glg_stft_kernel<1>), dim3(Tf, B), dim3(GLG_THREADS), glg_lds(n_fft), s, wav, n, window,
                           reinterpret_cast<const gcf*>(tw), reinterpret_cast<gcf*>(out), Tf, Fp, n_fft, m, win, hop, mag, mse_partial);
    else
        hipLaunchKernelGGL((glg_stft_kernel<0>), dim3(Tf, B), dim3(GLG_THREADS), glg_lds(n_fft), s, wav, n, window,
                           reinterpret_cast<const gcf*>(tw), reinterpret_cast<gcf*>(out), Tf, Fp, n_fft, m, win, hop, mag, mse_partial);
    return hipGetLastError();
}

}  // namespace tts
