/*
 * sstts_hip.h -- C ABI of the MI355X-native Tacotron inference hot path.
 *
 * One shared library (libsstts_hip.so, hand-written HIP for gfx950), plain pointers and
 * sizes, no C++/torch types.  The reference (yweweler/single-speaker-tts) has no FFI: its
 * boundary is the Python module surface of tacotron.model / tacotron.inference /
 * audio.synthesis / audio.conversion / audio.features.  Each entry point below names the
 * reference interface (file:line under the reference tree) whose arithmetic it replaces; the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - All tensor pointers are DEVICE pointers (HIP), float32 row-major, batch-major,
 *     channels-last unless stated; ids are int32.  The caller owns every in/out buffer;
 *     the library owns weights and scratch inside the handle.  tts_malloc/tts_memcpy_* let
 *     a host without any other GPU runtime (plain ctypes + numpy) drive the library.
 *   - Every function returns 0 on success or a negative tts_status code and never throws.
 *     tts_last_error(handle) returns a human-readable description of the last failure.
 *   - A handle is bound to one device and one stream; calls on one handle must be
 *     serialised by the caller.  Distinct handles (one per GPU / process) are independent:
 *     every entry point that takes a handle switches to the handle's device for the call and
 *     restores the caller's current device on return.
 *   - Calls are asynchronous on the handle's stream unless stated; tts_synchronize waits.
 *   - Same inputs (and the same init_phase / seed) give bit-identical outputs run to run:
 *     every reduction has a fixed order, no float atomics are used.
 */
#ifndef SSTTS_HIP_H
#define SSTTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tts_handle_s* tts_handle_t;

enum tts_status {
    TTS_OK = 0,
    TTS_ERR_INVALID = -1,      /* bad argument / shape */
    TTS_ERR_NOT_LOADED = -2,   /* weights missing or not finalised */
    TTS_ERR_HIP = -3,          /* HIP runtime error */
    TTS_ERR_DB_RANGE = -4,     /* dB < -100: reference audio/conversion.py:47-49 AssertionError */
    TTS_ERR_UNSUPPORTED = -5   /* configuration outside what the kernels implement */
};

/* Architecture hyper-parameters; field names and defaults follow the reference's
 * model_params (tacotron/params/model.py:8-153).  ALWAYS start from tts_default_config: it fills the defaults and
 * `struct_size`, and tts_create refuses a struct whose size is not the library's (a caller built against another version of
 * this header, or a zero-initialised struct) instead of reading past its end or taking zeros for settings. */
typedef struct tts_config {
    int32_t struct_size;         /* sizeof(tts_config_t) of the header the caller was built with (set by tts_default_config) */
    int32_t vocabulary_size;     /* 39  */
    int32_t embedding_size;      /* 256 */
    int32_t enc_prenet_units[2]; /* 256, 128 */
    int32_t enc_n_banks;         /* 16  */
    int32_t enc_n_filters;       /* 128 */
    int32_t enc_proj_filters[2]; /* 128, 128 (kernel size 3; relu, linear) */
    int32_t post_n_banks;        /* 8   */
    int32_t post_n_filters;      /* 128 */
    int32_t post_proj_filters[2];/* 256, 80 */
    int32_t n_highway_layers;    /* 4   */
    int32_t n_highway_units;     /* 128 */
    int32_t n_gru_units;         /* 128 (CBHG bi-GRU, both encoder and post-net) */
    int32_t dec_prenet_units[2]; /* 256, 128 */
    int32_t n_attention_units;   /* 256 */
    int32_t n_decoder_gru_units; /* 256 */
    int32_t n_decoder_gru_layers;/* 2   */
    int32_t n_mels;              /* 80  */
    int32_t reduction;           /* 5   */
    int32_t n_fft;               /* 2048 */
    int32_t force_cudnn;         /* 0: tf GRUCell (TF-CPU parity target); 1: CudnnCompatibleGRUCell */
    /* model_params.attention (tacotron/params/model.py:112-128).  The local mechanism is the reference's
     * experimental LocalLuongAttention (tacotron/attention.py:109-342) in its default sub-mode:
     * AttentionMode.MONOTONIC or PREDICTIVE with AttentionScore.DOT; GENERAL / CONCAT raise NotImplementedError
     * in the reference. */
    int32_t attention_mechanism; /* TTS_ATTENTION_LUONG (default) | TTS_ATTENTION_LOCAL_LUONG */
    int32_t luong_local_window_d;/* 10: window = 2D+1 memory positions around the decoder step index */
    int32_t luong_force_gaussian;/* 1: reported alignments are gaussian-weighted (attention.py:73-80) */
    int32_t luong_local_mode;    /* TTS_LOCAL_MONOTONIC (default): window centre = decoder step index, clamped into
                                  * the memory; TTS_LOCAL_PREDICTIVE: centre p = T_s sigmoid(v_p^T tanh(W_p h)) per
                                  * utterance (attention.py:246-258), two more weights in the manifest.  A predicted
                                  * window that leaves the memory makes tts_decoder_forward / tts_synthesize return
                                  * TTS_ERR_UNSUPPORTED (the reference fails at run time there, attention.py:288-304);
                                  * in this mode those calls synchronise the stream to read that condition. */
    int32_t apply_post_processing;/* 1 (default): post-net CBHG in front of the final Dense; 0: the final Dense straight on the
                                  * mel spectrogram, manifest entry dense/kernel (n_mels, 1 + n_fft / 2) and no post_process/...
                                  * weights (reference tacotron/model.py:388-391, params/model.py apply_post_processing) */
} tts_config_t;

enum tts_attention { TTS_ATTENTION_LUONG = 0, TTS_ATTENTION_LOCAL_LUONG = 1 };
enum tts_local_mode { TTS_LOCAL_MONOTONIC = 0, TTS_LOCAL_PREDICTIVE = 1 };

/* ---- lifecycle ------------------------------------------------------------------------ */
const char* tts_version(void);
int tts_default_config(tts_config_t* cfg);
/* Replaces graph construction `Tacotron(inputs, Mode.PREDICT)` (tacotron/model.py:35-112). */
int tts_create(const tts_config_t* cfg, int device_id, tts_handle_t* out);
int tts_destroy(tts_handle_t h);
const char* tts_last_error(tts_handle_t h);        /* h may be NULL: last create-time error */
/* Use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream. */
int tts_set_stream(tts_handle_t h, void* hip_stream);
/* Options: "use_graph" (launch-per-layer decoder loop replayed from a hipGraph; default 0; 1 is refused with
 * TTS_ERR_UNSUPPORTED when the process runs the library on a HIP runtime older than the one it was built with -- a host that
 * loads another ROCm's libamdhip64 first, e.g. by importing torch, serves the library with that one, and graph replays were
 * wrong on PyTorch's bundled 7.0 runtime; see csrc/api_internal.h, `use_graph`), "profile" (record
 * per-stage HIP events, default 0), "pipeline" (default 1: tts_synthesize runs the encoder and
 * the decoder loop of a call on a second stream so that they overlap the Griffin-Lim iterations of
 * the PREVIOUS call still in flight; inputs must be complete when the call is made, which is why 1 applies
 * only while the library owns its stream -- 2 pipelines on a stream adopted with tts_set_stream as well, the
 * caller vouching for its inputs), "reserve_cus" (default 32: compute units kept free of
 * Griffin-Lim workgroups for that second stream, 0 = none), "hold_lds_kb" (default 64: LDS one sleeper workgroup
 * of that reservation allocates), "persistent_decoder" (the whole decoder loop as ONE launch of co-resident
 * workgroup clusters; two decoder GRU layers, layer widths of the reference: 1 (default) = the weight-stationary kernel
 * wherever it covers the configuration and its workgroups fit -- pipelined calls of up to 64 utterances, unpipelined calls of
 * up to 512 -- with 16 or 32 utterances per cluster of 16 compute units as the free units allow (the same bits either way,
 * so a call's spectrograms do not depend on what the handle ran before); 2 = any persistent kernel whenever the configuration
 * allows; 0 = never: one launch per layer; tts_synchronize reports TTS_ERR_HIP if one of its bounded waits timed out),
 * "gl_pair" (Griffin-Lim iterations per launch, 1..3, default 3: the spectrum passes from one iteration to the next in
 * registers; identical arithmetic per iteration), "gl_wide_from" (pipelined calls: the first Griffin-Lim launch that is cut
 * for all compute units instead of all but "reserve_cus" -- the next call's decoder has left them by then; -1 (default) =
 * from a model of the two durations, -2 = never, n >= 0 = launch n; the waveform's bits do not depend on the cut),
 * "fused_tail" (default 1: lifter + highway stack + GRU input
 * projections of a CBHG as one launch; 0 = layer by layer), "enc_stream" (default 1: under the call pipeline with the
 * persistent decoder the encoder of a call runs on a stream of its own, one inter-Griffin-Lim gap ahead of its decoder, so
 * that the decoders of consecutive calls follow each other without a pause; tts_synthesize then waits on the HOST until the
 * device has reached the post-net of the call two back -- at most ~2.5 calls are ever queued; 0 = the encoder in front of
 * its decoder on the front stream, calls never block).
 * Test and diagnostic hooks -- per handle, inert (and refused with a non-zero value) until "debug_hooks" has been set to 1
 * on the same handle; nothing in the process environment changes what a call computes: "pd_debug_delay" (workgroup 3 of
 * every persistent-decoder cluster stages its tile that many x ~3.4 us late), "gl_runs" / "gl_run_len" (force the cut of
 * an utterance's frames into Griffin-Lim runs: runs per utterance / frames per run; the waveform's bits do not depend on
 * the cut), "gl_workers" (plan and launch Griffin-Lim for that many workgroups instead of one per free compute unit),
 * "pd_rows" (16 / 32: utterances per cluster of the weight-stationary decoder instead of the library's choice),
 * "timeline" (tts_profile_get prints the absolute times of every profiled span); tts_debug_hold.
 * Initial phases of Griffin-Lim: `init_phase` (a (B, F, T) array of U[0,1) numbers, angle = 2 pi u) or, when it is NULL,
 * a counter-based draw from `seed` made inside the first iteration's launch (the reference draws np.random.rand per call,
 * audio/synthesis.py:91). */
int tts_set_option(tts_handle_t h, const char* key, int value);
int tts_synchronize(tts_handle_t h);

/* Numerics: float32 throughout.  The dense and convolution layers form their f32 products from exact three-way bf16 splits of
 * both operands on the bf16 matrix pipe (same measured error against float64 as f32-input MFMA, tests/test_gpu_gemm.py).  One
 * difference from IEEE f32 arithmetic on non-finite values: an output that depends on a +-Inf operand is NaN (the split of Inf
 * contains Inf - Inf), not +-Inf; NaN operands give NaN; every output that depends on finite operands only is unaffected. */

/* ---- weights: replaces tf.train.Saver().restore (tacotron/inference.py:55,71) ---------- */
/* Manifest: names follow the TF variable scopes (see single-speaker-tts_amd/tacotron/weights.py). */
int tts_manifest_size(tts_handle_t h);
int tts_manifest_entry(tts_handle_t h, int index, const char** name, int64_t shape[4], int* ndim);
/* host_data: HOST pointer, TensorFlow layout (Dense (in,out); conv (k,in,out); GRU gates
 * (in+units, 2*units) = rows [input;state], columns [r|u]). */
int tts_set_weight(tts_handle_t h, const char* name, const float* host_data, const int64_t* shape, int ndim);
/* All tensors of the manifest, concatenated in manifest order (the RCCL broadcast unit). */
int tts_load_weights_blob(tts_handle_t h, const float* host_blob, size_t n_floats);
/* Validates completeness, folds batch-norm, packs device layouts.  Synchronous. */
int tts_finalize_weights(tts_handle_t h);

/* ---- device memory helpers ---------------------------------------------------------- */
/* tts_malloc / tts_free act on the device that is CURRENT on the calling thread; with more than one GPU in
 * a process use the handle-bound pair, which allocates on the handle's device whatever is current. */
int tts_malloc(void** dptr, size_t bytes);
int tts_free(void* dptr);
int tts_device_malloc(tts_handle_t h, void** dptr, size_t bytes);
int tts_device_free(tts_handle_t h, void* dptr);
int tts_memcpy_h2d(tts_handle_t h, void* dst, const void* src, size_t bytes);  /* synchronous */
int tts_memcpy_d2h(tts_handle_t h, void* dst, const void* src, size_t bytes);  /* synchronous */
int tts_memset(tts_handle_t h, void* dst, int value, size_t bytes);

/* ---- network stages -------------------------------------------------------------------- */
/* Tacotron.encoder (tacotron/model.py:124-173): embedding + pre-net + CBHG.
 * ids int32 [B*Ts] -> memory float [B*Ts*2*n_gru_units].  An id outside [0, vocabulary_size) reads as
 * a zero embedding row (TensorFlow's GPU kernel does the same; its CPU kernel raises, which the Python
 * mirror reproduces for host arrays -- device-resident ids cannot be inspected without a sync). */
int tts_encoder_forward(tts_handle_t h, const int32_t* ids, int B, int Ts, float* memory);
/* Tacotron.decoder in Mode.PREDICT (tacotron/model.py:175-334; wrappers.py:94-124;
 * helpers.py:83-110,161-205): n_steps strictly sequential steps (reference: 1000//5 = 200).
 * memory [B*Ts*256] -> reduced mel [B * n_steps * (reduction*n_mels)] (== output_mel_spec
 * reshaped, model.py:383) and alignment_history [n_steps * B * Ts] (may be NULL). */
int tts_decoder_forward(tts_handle_t h, const float* memory, int B, int Ts, int n_steps,
                        float* mel, float* alignments);
/* Tacotron.post_process + final Dense (tacotron/model.py:336-363, 394-398).
 * mel [B*T*n_mels] -> output_linear_spec [B*T*(1+n_fft/2)]. */
int tts_postnet_forward(tts_handle_t h, const float* mel, int B, int T, float* linear);

/* ---- spectrogram de-normalisation ------------------------------------------------------ */
/* inference() post-step + synthesize() power (tacotron/inference.py:93-101,175;
 * audio/conversion.py:81-102, 32-53): per utterance transpose to (F,T),
 * db = (clip(x,0,1)-1)*(|ref|+|max|)+ref, mag = 10^(db/20), mag ** power.
 * linear [B*T*F] -> mag [B*F*T].  Returns TTS_ERR_DB_RANGE when some value de-normalises to less than
 * -100 dB (decibel_to_magnitude's assertion; checked on the data, and only for constants that allow it:
 * ref - |ref| - |max| < -100 -- the call then synchronises the stream). */
int tts_denorm_power(tts_handle_t h, const float* linear, int B, int T, int F,
                     float ref_db, float max_db, float power, float* mag);

/* ---- Griffin-Lim ------------------------------------------------------------------------ */
/* griffin_lim_v2 / spectrogram_to_wav (audio/synthesis.py:5-125), batched.
 * mag [B*F*T] (F = 1+n_fft/2, reference layout (F,T) per utterance);
 * init_phase [B*F*T] of U[0,1) numbers = what np.random.rand returns at synthesis.py:85,
 * or NULL to draw them on device from `seed`;  wav [B * hop*(T-1)];  mse [B] or NULL
 * (mean squared magnitude error of the last iteration, synthesis.py:112). */
int tts_griffin_lim(tts_handle_t h, const float* mag, const float* init_phase, uint64_t seed,
                    int B, int T, int n_iter, int win_length, int hop_length, int n_fft,
                    float* wav, float* mse);
/* librosa.output.write_wav(norm=True) scaling (audio/io.py:53): wav /= max|wav| per
 * utterance unless the peak is below FLT_MIN.  In place, wav [B*n]. */
int tts_peak_normalize(tts_handle_t h, float* wav, int B, int n);

/* ---- analysis features (audio/features.py:5-86,116-145) and dB helpers ---------------- */
/* librosa.stft(wav, n_fft, hop, win) as linear_scale_spectrogram returns it (features.py:145):
 * centre/reflect padding, periodic hann.  wav [B*n] -> out complex64 interleaved
 * [B*F*n_frames*2], n_frames = 1 + n/hop. */
int tts_stft(tts_handle_t h, const float* wav, int B, int n, int n_fft, int win_length,
             int hop_length, float* out);
/* abs(stft) ** power (features.py:62-71).  wav [B*n] -> lin [B*F*n_frames]. */
int tts_stft_magnitude(tts_handle_t h, const float* wav, int B, int n, int n_fft, int win_length,
                       int hop_length, float power, float* lin);
/* np.dot(librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=True), lin) (features.py:75-84).
 * lin [B*F*n_frames] -> mel [B*n_mels*n_frames]. */
int tts_mel_spectrogram(tts_handle_t h, const float* lin, int B, int n_frames, int n_fft,
                        int sampling_rate, int n_mels, float fmin, float fmax, float* mel);
/* Elementwise audio/conversion.py: mode 0 magnitude_to_decibel (:5-29), 1 decibel_to_magnitude
 * (:32-53; TTS_ERR_DB_RANGE if any input < -100, synchronous), 2 normalize_decibel (:56-78),
 * 3 inv_normalize_decibel (:81-102).  in/out [n] (may alias). */
int tts_db_convert(tts_handle_t h, const float* in, size_t n, int mode, float ref_db, float max_db,
                   float* out);

/* ---- end to end -------------------------------------------------------------------------- */
/* ids -> waveform: encoder, decoder (n_steps), post-net, de-normalise, ** power, Griffin-Lim,
 * optional peak normalisation; replaces tacotron/inference.py:162-200 minus file IO.
 * Optional outputs (NULL to skip): mel [B*n_steps*r*n_mels], alignments [n_steps*B*Ts],
 * linear [B*T*F].  wav [B * hop*(T-1)], T = n_steps*reduction. */
typedef struct tts_synth_params {
    int32_t n_steps;        /* 200 */
    float ref_db, max_db;   /* 6.02, 99.89 (mel constants, as the reference uses them) */
    float power;            /* 1.3 */
    int32_t n_iter;         /* 50 */
    int32_t win_length, hop_length;  /* 1102, 275 */
    uint64_t seed;          /* random initial phase when init_phase == NULL */
    int32_t peak_normalize; /* save_wav(norm=True) */
    int32_t host_outputs;   /* tts_synthesize_host only: TTS_HOST_* flags of what travels to host memory besides the
                               waveforms (0 = nothing); ignored by tts_synthesize, whose optional outputs are pointers */
} tts_synth_params_t;
#define TTS_HOST_LINEAR 1       /* linear spectrograms [B*T*F]: what the reference's inference() hands back (tacotron/inference.py:75-101) */
#define TTS_HOST_ALIGNMENTS 2   /* alignments [n_steps*B*Ts] (tacotron/model.py:552-598 dumps them) */
int tts_synthesize(tts_handle_t h, const int32_t* ids, int B, int Ts, const tts_synth_params_t* p,
                   const float* init_phase /* [B*F*T] or NULL */, float* wav, float* mel,
                   float* alignments, float* linear);

/* The same call for a caller that lives in HOST memory -- what the reference's inference() / serve() are
 * (tacotron/inference.py:75-101,185-200, serve.py:89-126: host id arrays in, host waveforms out) -- without giving up the
 * call pipeline: the ids are copied to a pinned staging buffer of the handle and uploaded on a copy stream, the
 * call is enqueued behind that upload (the initial phases are drawn on the device from p->seed, as the reference
 * draws them with np.random), and the waveforms are downloaded into pinned memory of the handle on a second copy
 * stream as soon as they are complete.  The call returns at once with a ticket; tts_wait_host blocks until that
 * call's waveforms have arrived and hands out the pinned buffer, [B * hop*(T-1)] floats, valid until the THIRD
 * tts_synthesize_host call after the one that produced it (three buffer sets rotate: the device holds three calls at
 * once -- the encoder of call k + 2, the decoder of k + 1, the post-net / Griffin-Lim of k).  Keep at most three calls in
 * flight: submit k + 2, then wait for k.  Bit-identical to tts_synthesize + tts_memcpy_d2h. */
int tts_synthesize_host(tts_handle_t h, const int32_t* ids_host, int B, int Ts, const tts_synth_params_t* p,
                        int* ticket);
int tts_wait_host(tts_handle_t h, int ticket, const float** wav_host, size_t* n_floats);
/* The optional outputs of the same call (p->host_outputs), in pinned memory of the handle with the lifetime of the
 * waveform buffer: linear spectrograms [B][T][F] as the network emits them (normalised dB, before the de-normalisation),
 * alignments [n_steps][B][Ts].  Pointers are NULL / counts 0 for outputs the call did not ask for.  Waits like
 * tts_wait_host (and reports what it reports); the waveforms are fetched with tts_wait_host itself. */
int tts_wait_host_outputs(tts_handle_t h, int ticket, const float** linear_host, size_t* n_linear, const float** align_host,
                          size_t* n_align);

/* ---- profiling -------------------------------------------------------------------------- */
/* With option "profile"=1 the library brackets its stages with HIP events on the handle's
 * stream.  Stages: "encoder", "decoder", "postnet", "denorm", "gl_iter", "gl_final", "debug_gemm"
 * (launches of tts_debug_gemm).
 * Returns accumulated milliseconds and the number of kernel launches covered since the last
 * tts_profile_reset.  Synchronises the stream. */
int tts_profile_reset(tts_handle_t h);
/* Which decoder a call of this shape takes with the handle's current options: 0 = launch per layer (decoder.hip),
 * 1 = persistent with streamed weights (decoder_persistent.hip), 2 = persistent, weight-stationary (decoder_ws.hip);
 * `pipelined` != 0: as a call under tts_synthesize's call pipeline (reserve_cus compute units), else a stand-alone call.
 * Host-only (reads the configuration; nothing is enqueued). */
int tts_decoder_kernel_choice(tts_handle_t h, int B, int T_sent, int pipelined);
/* Test hook: device pointer and size of a named internal scratch buffer of the last call
 * ("enc.bank", "enc.p1", "post.xproj", ...); contents are only valid until the next call. */
int tts_debug_workspace(tts_handle_t h, const char* name, void** dptr, size_t* bytes);
/* Host-only: the Griffin-Lim work-item cut for T frames x B utterances on n_workers compute units.
 * items[n][4] = {utterance, first frame, frames, slot word} in the order the workgroups draw them (at most max_items
 * are written); *ring_frames = frames the LDS ring of the kernel holds; returns the number of items. */
int tts_debug_gl_plan(int T, int B, int win_length, int hop_length, int n_workers, int* items, int max_items, int* ring_frames);
/* Diagnostic: one GEMM / conv1d launch on device buffers (A [M][Cin] rows of sequences of length T, Wt [N][ktaps*Cin]). */
int tts_debug_gemm(tts_handle_t h, const float* A, const float* Wt, float* C, int M, int N, int Cin, int ktaps, int T,
                   int pool);
/* Diagnostic (needs the option "debug_hooks" = 1 on the handle): keep n_wgs workgroup slots of lds_kb KB LDS busy for ms
 * milliseconds on a private stream. */
int tts_debug_hold(tts_handle_t h, int n_wgs, int lds_kb, double ms);
int tts_profile_get(tts_handle_t h, const char* stage, float* ms_total, int64_t* launches);
/* The UUID of the handle's device as 32 hex digits + NUL (hipDeviceGetUuid) and its compute-unit count: what the
 * multi-GPU bench compares across ranks (no two ranks of one node on the same device; no reference counterpart --
 * the reference is a one-process TensorFlow session, tacotron/inference.py:44-55). */
int tts_device_info(tts_handle_t h, char uuid_hex[33], int* n_compute_units);

#ifdef __cplusplus
}
#endif
#endif /* SSTTS_HIP_H */
