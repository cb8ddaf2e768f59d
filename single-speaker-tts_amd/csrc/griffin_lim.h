// Griffin-Lim / audio kernel launchers shared between griffin_lim.hip and the api_*.hip files.
#pragma once
#include "tts_common.h"
#include <vector>

namespace tts {

#define TTS_GL_FP 1056      // padded row length of the frame-major spectra (F = 1025): rows start on 128-byte lines
#define TTS_GL_NFFT 2048

struct GlParams {
    const float* mag;        // [B][T][FP]
    const float2* phase_in;  // [B][T][FP] 32-bit phasor codes of the current estimate (the pointer type is historical)
    float2* phase_out;       // [B][T][FP]       (iteration)
    float* wav;              // [B][hop*(T-1)]   (final iSTFT)
    float* mse_partial;      // [B][slots_per_utt] or null
    float* peak_partial;     // [B][slots_per_utt] or null (final iSTFT: per-run max |wav|)
    const float* window;     // [win] periodic hann
    const float* wlane;      // [2][64 lanes][16][2] per-lane window images (gl_build_wlane): analysis / interior synthesis
    const float* rwss;       // [n_fft + hop*(T-1)] 1 / window sum-square (librosa window_sumsquare) where it is > tiny, else 1
    const float2* tw1024;    // exp(-2 pi i k / 1024), k < 1024
    const float2* tw2048;    // exp(-2 pi i k / 2048), k < 1024
    const float2* tables;    // [tw2048 (1024) | W1024^{lane*k2} as [k2-1][lane] (15*64)]: per-lane twiddles, coalesced
    int T, FP, win, hop;
    int B;                   // utterances
    int ncol;                // ceil(win / hop): frames that overlap a sample, halo = ncol - 1
    // work items of a launch (gl_plan_stream): a table of RUNS in device memory, {utterance, first frame, frames, slot};
    // item ids are drawn in table order (first every workgroup's first run, then the runs that follow them).  A run's
    // partial results (mse, peak) go to slot (w & 0xffff) of its utterance; the run with the utterance's last slot carries
    // in w >> 16 how many slots up to slots_per_utt it has to zero (utterances are not all cut into the same number of runs).
    const int4* items;
    int n_items, slots_per_utt;
    int n_workers;           // workgroups the cut was made for (= the launch's grid)
    int ring_frames;         // streaming form (gl_stream_kernel): frames an LDS ring holds
    int n_stage;             // ... iterations per launch (1..3), set by launch_gl_stream
    // seeded start (no initial-phase array): the first launch of a call makes the initial phasor of every bin itself
    // (gl_seed_phasor of the seed and the bin's index in the reference's (B, F, T) layout) instead of reading codes
    int seeded, F;
    unsigned long long seed;
    unsigned* work_counter;  // zeroed counter of THIS launch: the persistent workgroups draw item ids from it
    unsigned* clear_counter; // null, or the counter of an EARLIER launch on this stream: one thread zeroes it for a later launch
    unsigned long long* dbg; // tools only (-DGL_TIMELINE builds): [GL waves][64] s_memrealtime stamps of workgroup 0
};

// out[2*16*2*64]: set 0 = window[n] / n_fft, set 1 = set 0 * rwss at an interior frame; n = 2*(lane + 64 c) + e
void gl_build_wlane(const float* window, const float* rwss, int win, int hop, int T, float* out);
// streaming form of the iteration / final iSTFT (gl_stream_kernel): no chunks, a run is one stream through an LDS ring
int gl_stream_ring_frames(int win, int hop, int n_stage = 1);   // 0: the window / hop pair does not fit
bool gl_stream_instantiated(int win, int hop);                  // the (window, hop) pairs gl_stream_kernel is compiled for (n_fft 2048)
// needs T, B, win, hop, ncol; sets items / n_items / slots_per_utt (the cut for launches of n_stage iterations on n_workers
// workgroups; the table is uploaded to the current device once per shape and set of speeds and lives as long as the process)
// stream: the stream the launches that use the cut will be enqueued on (a table that is new is uploaded there)
hipError_t gl_plan_stream(GlParams& p, int n_workers, int n_stage = 1, int force_runs = 0, int force_run_len = 0, hipStream_t stream = nullptr);
// the same cut on the host alone (no device): items[n][4] = {utterance, first frame, frames, slot word}; returns n
int gl_plan_items(int T, int B, int win, int hop, int n_workers, int n_stage, int force_runs, int force_run_len,
                  std::vector<int4>* items, int* slots_per_utt, int* workers_out = nullptr);
hipError_t launch_gl_stream(hipStream_t s, const GlParams& p, int n_cus, int final_istft, int n_stage = 1);
hipError_t gl_configure();
size_t gl_state_bytes();   // bytes per bin of the state between launches (4: a phasor code)
hipError_t launch_gl_mse_reduce(hipStream_t s, const float* partial, int B, int nchunks, float denom, float* mse);
hipError_t launch_mag_ft_to_tf(hipStream_t s, const float* in, float* out, int B, int F, int T, int FP);
hipError_t launch_tf_to_ft(hipStream_t s, const float* in, float* out, int B, int F, int T, int FP);
hipError_t launch_phase_init(hipStream_t s, const float* init_ft, uint64_t seed, void* out, int B, int F, int T, int FP);
hipError_t launch_denorm_power(hipStream_t s, const float* lin, float* mag, size_t rows, int F, int FP,
                               float ref_db, float max_db, float power, int* below_flag);
hipError_t launch_peak_normalize(hipStream_t s, float* wav, int B, int n);
hipError_t launch_peak_scale(hipStream_t s, float* wav, int B, int n, const float* partial, int nparts);
hipError_t launch_stft(hipStream_t s, const float* wav, int B, int n, int Tf, const float* window, int win, int hop,
                       const float2* tw1024, const float2* tw2048, float2* out, int FP);
hipError_t launch_cplx_tf_to_ft(hipStream_t s, const float2* in, float* out, int B, int F, int T, int FP, int mode,
                                float power);
hipError_t launch_db_convert(hipStream_t s, const float* in, float* out, size_t n, int mode, float ref_db, float max_db);
hipError_t launch_any_below(hipStream_t s, const float* in, size_t n, float lim, int* flag);

// general power-of-two path (griffin_lim_generic.hip): one workgroup per frame, FFT in LDS; spectra [B][T][Fp], state =
// float2 unit phasors; tw = exp(-2 pi i k / n_fft), k < n_fft / 2
bool glg_supports(int n_fft);   // power of two, 256 .. 4096
hipError_t glg_configure();
hipError_t launch_glg_phase_init(hipStream_t s, const float* init_ft, uint64_t seed, float2* out, int B, int F, int T, int Fp);
// frames [B][T][win] scratch; wav [B][hop (T - 1)]: inverse transform of every frame, then the overlap-add (a gather in frame order)
hipError_t launch_glg_istft(hipStream_t s, const float* mag, const float2* ph, const float* window, const float* rwss, const float2* tw,
                            float* frames, float* wav, int B, int T, int Fp, int n_fft, int win, int hop);
// mode 0: out = unit phasors of the spectrum (+ per-frame squared magnitude error against mag when mse_partial != null);
// mode 1: out = the complex spectrum
hipError_t launch_glg_stft(hipStream_t s, const float* wav, int n, const float* window, const float2* tw, float2* out, int B, int Tf, int Fp,
                           int n_fft, int win, int hop, int mode, const float* mag, float* mse_partial);

}  // namespace tts
