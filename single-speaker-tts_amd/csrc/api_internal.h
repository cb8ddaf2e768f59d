// Internals shared by the three translation units of the C ABI (api_handle.hip: handle, workspace allocator, weights, options;
// api_stages.hip: the stage entry points and what they enqueue; api_pipeline.hip: tts_synthesize and its host-memory form,
// the scheduler of the three streams).  Not part of the interface: include/sstts_hip.h is.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>
#include "decoder.h"
#include "griffin_lim.h"
#include "tts_common.h"
#include <algorithm>
#include <map>

namespace tts_api {
using namespace tts;


extern thread_local std::string g_create_error;   // tts_create failures (no handle to keep the message in)

struct ManifestEntry {
    std::string name;
    std::vector<int64_t> shape;
    size_t numel() const {
        size_t n = 1;
        for (auto d : shape) n *= (size_t)d;
        return n;
    }
};

struct CbhgWeights {
    int n_banks = 0, n_filters = 0, c_in = 0, proj_filters[2] = {0, 0};
    // device pointers into the arena
    std::vector<const float*> bank_wt, bank_b, bank_scale, bank_shift;
    const float* proj_wt[2];
    const float* proj_b[2];
    const float* proj_scale[2];
    const float* proj_shift[2];
    const float* lifter_wt;
    const float* lifter_b;
    std::vector<const float*> hw_wt, hw_b;
    const float* gru_in_wt;   // [2*3H][units]
    const float* gru_in_b;    // [2*3H]
    const float* gru_rec;     // packed recurrent weights, both directions
};

enum Stage { ST_ENCODER = 0, ST_DECODER, ST_POSTNET, ST_DENORM, ST_GL_ITER, ST_GL_FINAL, ST_DEBUG_GEMM, ST_COUNT };
extern const char* const kStageNames[ST_COUNT];

struct ProfSpan {
    hipEvent_t a, b;
    int stage;
    int64_t launches;
};


}  // namespace tts_api
using namespace tts_api;   // (the handle is a global type: include/sstts_hip.h declares tts_handle_s)

#ifndef TTS_USE_GRAPH_DEFAULT
#define TTS_USE_GRAPH_DEFAULT 0   // (tools: -DTTS_USE_GRAPH_DEFAULT=1 builds a library whose handles replay the decoder graph)
#endif
struct tts_handle_s {
    tts_config_t cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // Launch-per-layer decoder: replay the whole loop from one executable hipGraph instead of enqueueing its ~10 launches per
    // step (9.30 against 9.42 ms for 200 steps at B = 64: the dependent launches are GPU-bound at ~4.7 us each).  OFF by default,
    // and REFUSED on a HIP runtime older than the one the library was built and validated with (graph_runtime_ok below).
    // Round 5 saw replays return wrong mel spectrograms "in a long-lived process"; round 6 found what that process had in
    // common: it had imported torch before the library, so the library ran on PyTorch's BUNDLED libamdhip64 (HIP 7.0.51831, same
    // soname) instead of /opt/rocm's 7.2.26015.  On that runtime a cached decoder graph replays wrongly after other work on the
    // handle (tools/graph_probe.py --torch: 5 of 5, garbage of 1e10...1e33 or last-bit differences; whole suite green there with
    // DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, i.e. without the runtime's pre-built AQL packets -- the graph's dec_gemm_kernel nodes use
    // 16 bytes of scratch); the same binary and sequence are right on the 7.2 runtime, graph on, every time
    // (profiles/r06_experiment_hipgraph.txt).
    int use_graph = TTS_USE_GRAPH_DEFAULT;   // (a tools build with the default ON still checks the runtime: tts_create)
    int fused_tail = 1;          // CBHG: lifter + highway stack + GRU input projections as one launch (cbhg_tail.hip)
    bool tail_configured = false;
    int profile = 0;
    // tts_synthesize pipelining: encoder + decoder (latency bound, few CUs) of call k+1 run on
    // `front` while post-net + Griffin-Lim (throughput bound) of call k run on `stream`.
    int pipeline = 1;      // on while the library owns its stream (see tts_synthesize); ~9 % on MI355X
    int reserve_cus = 32;  // CUs held for the front stream by LDS-hogging sleeper workgroups (reserve.hip)
    int hold_lds_kb = 64;  // LDS of one sleeper: > 80 KB guarantees one sleeper per CU
    hipStream_t aux = nullptr;      // stream the sleepers run on
    int* hold_flags = nullptr;      // two flag words, alternating per call
    hipEvent_t ev_aux = nullptr;
    unsigned call_count = 0;
    // The front stream (encoder, decoder, explicit initial phases of the NEXT call beside this call's post-net and Griffin-Lim):
    // greatest priority.  Measured alternatives for the calls of the persistent decoder (which keeps its CUs by being
    // resident): lowest priority was 0.15 ms per step better while the main stream was the longer one and 0.1 ms worse once
    // Griffin-Lim's run cut had given it slack; the main stream's own priority is as good as the greatest in a device-resident
    // loop but HALVES the throughput of tts_synthesize_host -- streams of one priority share a few hardware queues, and with
    // the two copy streams of the host path the front stream lands on the main stream's queue.
    hipStream_t front = nullptr;
    // The encoder of a pipelined call runs on a stream of its own (round 4): it depends on the ids only, so it need not
    // queue behind the previous call's decoder on the front stream -- it runs as soon as the decoder of the call TWO back
    // has finished with this parity's `memory` buffer, i.e. one inter-Griffin-Lim gap earlier, and the decoders follow
    // each other back to back (the step was enc + dec = 17.2 ms against 15.6 ms of post-net + Griffin-Lim).
    hipStream_t encs = nullptr;
    int enc_stream = 1;   // option "enc_stream": 0 = the encoder on the front stream in front of its decoder (round 3)
    hipEvent_t ev_enc_ready[2] = {nullptr, nullptr};   // encoder of the last call of this parity done (enc stream)
    hipEvent_t ev_dec_done[2] = {nullptr, nullptr};    // decoder of the last call of this parity done (front stream)
    // ... and not before the main stream has reached the post-net of that call (the Griffin-Lim phase before it is over):
    // an encoder let loose during a Griffin-Lim phase gets its compute units one launch boundary at a time (3 ms for 0.75 ms
    // of work) and slows those launches by 15 %; in the gap it shares the chip with the post-net, as before
    hipEvent_t ev_gap[2] = {nullptr, nullptr};
    bool enc_ready_pending[2] = {false, false}, dec_done_pending[2] = {false, false}, gap_pending[2] = {false, false};
    hipEvent_t ev_front_done = nullptr;
    hipEvent_t ev_post_done[2] = {nullptr, nullptr};   // post-net of the calls of even / odd parity
    bool post_pending[2] = {false, false};
    bool gl_wide_used[2] = {false, false};   // the Griffin-Lim phase of that parity's last call ends in launches on ALL compute units
    hipEvent_t ev_gl_done[2] = {nullptr, nullptr};     // Griffin-Lim of the calls of even / odd parity (its phase buffers are free)
    bool gl_pending[2] = {false, false};
    bool front_pending = false;     // ev_front_done has been recorded at least once
    hipEvent_t ev_serial_done = nullptr;   // encoder + decoder of an UNPIPELINED call (they ran on the main stream)
    bool serial_pending = false;           // ... has been recorded since the front stream last waited for it
    unsigned syn_calls = 0;
    int syn_shape[3] = {0, 0, 0};   // (B, Ts, n_steps) of the previous tts_synthesize call
    int last_enc_ahead = -1;        // did the previous PIPELINED call run its encoder ahead on `encs` (1) or on `front` (0)?
    bool in_synthesize = false;     // the stage entry points are being called by tts_synthesize (which orders the streams itself)
    // persistent decoder (decoder_ws.hip / decoder_persistent.hip): 0 never, 2 whenever a kernel covers the configuration,
    // 1 (default) where it was measured to be the faster choice: pd_choice() below has the rule and the numbers.
    int persistent_decoder = 1;
    // which persistent kernel: 1 (default) = the weight-stationary one (decoder_ws.hip: clusters of 16 workgroups x 32
    // utterances, weights in registers) wherever it covers the configuration and its 16 * ceil(B / 32) workgroups fit the
    // budget, else decoder_persistent.hip (8 x 16, weights streamed from L2 every step); 0 = always the latter
    int pd_ws = 1;
    // (the decoder's output projection -- one GEMM over all steps -- runs on the MAIN stream in front of the post-net under the
    //  call pipeline: `defer_projection`.  On the front stream behind its decoder it gave the same 14.45 ms per step in round 5;
    //  the option that switched it is gone)
    bool ws_configured = false;
    // Round 5's two GEMM variants, measured and not faster (profiles/r05_experiment_gemm_presplit.txt, HISTORY.md part C): weights
    // pre-split into the kernel's bf16 LDS images ("gemm_presplit": images made on first use per weight matrix, keyed by its
    // address in the arena; tts_finalize_weights drops them) and the producer / consumer form of the kernel ("gemm_ps").  Their
    // kernels are only compiled into a tools build of gemm_f32.hip (-DGEMM_EXPERIMENTS); the shipped library refuses both options.
    struct WeightImage { unsigned char* p = nullptr; size_t bytes = 0; int N = 0, K = 0, Cin = 0; };
    std::map<const float*, WeightImage> wimg;
    int gemm_presplit = 0;
    int gemm_ps = 0;
    int gl_pair = 3;                 // Griffin-Lim iterations per launch (1..3) where nothing per-iteration is asked for
    // First Griffin-Lim launch of a pipelined call that is cut for all compute units (gl_run, `wide_from`): -1 = by the rule
    // in gl_wide_from() below, -2 = never, >= 0 = that launch index.
    int gl_wide = -1;
    int n_cus_dev = 0;
    bool pd_configured = false;
    // Test / diagnostic hooks, all per handle and all inert unless the option "debug_hooks" has been set to 1 on THIS handle
    // (include/sstts_hip.h): nothing in the environment and no other handle can change what a call computes.
    int debug_hooks = 0;
    int pd_debug_delay = 0;   // PdParams::dbg_delay: workgroup 3 of every decoder cluster stages late
    int gl_runs = 0;          // Griffin-Lim run cut: runs per utterance (0 = planned)
    int gl_run_len = 0;       // ... or frames per full run (0 = planned)
    int timeline = 0;         // print the absolute stage times of every profiled span (prof_collect)
    int gl_workers = 0;       // Griffin-Lim: plan and launch for this many workgroups (0 = the free compute units)
    // Griffin-Lim work counters: a ring of slots, zeroed once; a launch takes the next slot and zeroes its predecessor's
    unsigned* gl_ring = nullptr;       // the ring the bookkeeping below refers to (a re-allocated workspace starts over)
    unsigned gl_ring_seq = 0;
    unsigned* gl_ring_last = nullptr;  // slot of the most recent launch (dirty)
    hipStream_t gl_ring_stream = nullptr;
    float* pre_keys = nullptr;   // attention keys of the memory the next tts_decoder_forward gets, already computed (tts_synthesize)
    bool pd_used = false;            // a persistent launch has been enqueued since the last status check
    unsigned* pd_sync = nullptr;     // counters + status word of the last persistent launch
    int pd_clusters = 0;
    int* cur_hold_flag = nullptr;    // set by tts_synthesize around its decoder call: the sleepers' flag
    int cur_cu_budget = 0;           // ... and the compute units the front stream may count on (0 = the whole chip)
    bool dec_chip_idle = false;      // tts_synthesize: the main stream had nothing in flight when this call's decoder was enqueued
    int pd_rows = 0;                 // tests ("pd_rows" behind "debug_hooks"): utterances per cluster of the weight-stationary decoder, 16 / 32
    int pd_rows_used = 0;            // ... of the last launch

    // host-memory calls (tts_synthesize_host): pinned staging of the ids, device copies, pinned waveform buffers and the
    // device buffers they are copied from, one set per call in flight (ticket mod 3: the device pipeline holds three calls
    // at once since round 4 -- encoder of k + 2, decoder of k + 1, Griffin-Lim of k); two copy streams
    struct {
        hipStream_t in = nullptr, out = nullptr;
        int32_t* ids_pinned[3] = {nullptr, nullptr, nullptr};
        int32_t* ids_dev[3] = {nullptr, nullptr, nullptr};
        size_t ids_bytes = 0;
        float* wav_dev[3] = {nullptr, nullptr, nullptr};
        float* wav_pinned[3] = {nullptr, nullptr, nullptr};
        size_t wav_bytes = 0;
        hipEvent_t ev_h2d[3] = {nullptr, nullptr, nullptr};      // upload of the ids done
        hipEvent_t ev_enc[3] = {nullptr, nullptr, nullptr};      // encoder done with the ids buffer
        hipEvent_t ev_ready[3] = {nullptr, nullptr, nullptr};    // waveforms complete on the device
        hipEvent_t ev_d2h[3] = {nullptr, nullptr, nullptr};      // waveforms have arrived in pinned memory
        bool d2h_pending[3] = {false, false, false}, enc_pending[3] = {false, false, false};
        size_t n_floats[3] = {0, 0, 0};
        // optional outputs of a host call (tts_synth_params_t::host_outputs): linear spectrograms and alignments
        float* lin_dev[3] = {nullptr, nullptr, nullptr};
        float* lin_pinned[3] = {nullptr, nullptr, nullptr};
        size_t lin_bytes = 0;
        float* ali_dev[3] = {nullptr, nullptr, nullptr};
        float* ali_pinned[3] = {nullptr, nullptr, nullptr};
        size_t ali_bytes = 0;
        size_t n_lin[3] = {0, 0, 0}, n_ali[3] = {0, 0, 0};
        bool failed[3] = {false, false, false};   // this set's call ended on a decoder timeout: EVERY wait on its ticket fails
        int* status_pinned = nullptr;   // [3][2]: the persistent decoder's sticky status word ([.][1]) as it stood behind
                                        // each call's download
        int tickets = 0;
    } hio;
    // Under the call pipeline the decoder's output projection (y history -> mel, one GEMM) is not issued behind the decoder
    // on the front stream, where it gets the decoder's 32 compute units (0.3 ms), but at the head of the post-net on the
    // main stream (0.03 ms); the y history is then a buffer per call parity.
    bool defer_projection = false, has_pending_proj = false;
    int defer_parity = 0;
    GemmGroup pending_proj;
    hipEvent_t input_event = nullptr;   // set around a tts_synthesize call: its first kernel waits for this event
    hipEvent_t enc_done_event = nullptr;   // ... and this one is recorded behind its encoder

    std::vector<ManifestEntry> manifest;
    std::map<std::string, std::vector<float>> host_w;
    bool finalized = false;

    // device weight arena
    float* arena = nullptr;
    size_t arena_floats = 0;

    const float* embedding = nullptr;
    const float* enc_pre_wt[2];
    const float* enc_pre_b[2];
    CbhgWeights enc, post;
    const float* mem_wt = nullptr;
    DecoderWeights dec;
    const float* dense_wt = nullptr;
    const float* dense_b = nullptr;
    const float* zeros = nullptr;   // 1024 zero floats inside the arena

    // workspace (grow-only)
    std::map<std::string, DevBuf> ws;

    // decoder graph cache
    hipGraphExec_t dec_graph = nullptr;
    // A launch of dec_graph is complete: recorded behind every hipGraphLaunch, waited for by the HOST before the same
    // executable graph is launched again or destroyed (never two launches of one hipGraphExec_t in flight, never one
    // destroyed under a launch).
    hipEvent_t ev_graph_done = nullptr;
    bool graph_in_flight = false;
    hipGraph_t dec_graph_src = nullptr;   // the captured graph the executable one was instantiated from: kept alive with it
    struct {   // everything the captured launches have baked in: shapes and EVERY pointer (decoder_impl)
        const void* memory = nullptr;
        const void* keys = nullptr;
        void* align = nullptr;
        int B = 0, Ts = 0, n_steps = 0;
        DecoderScratch sc;
        DecoderWeights w;
    } dec_key;

    // Griffin-Lim tables
    struct {
        int win = 0, hop = 0, T = 0;
        float* window = nullptr;
        float* wss = nullptr;      // reciprocal window sum-square
        float* wlane = nullptr;    // per-lane window images of the Griffin-Lim kernel
        float2* tw1024 = nullptr;
        float2* tw2048 = nullptr;
        float2* tables = nullptr;
        bool configured = false;
        int n_cus = 0;
    } gl;

    // general power-of-two path (griffin_lim_generic.hip): twiddles per n_fft, window tables of the last configuration
    struct {
        std::map<int, float2*> tw;          // n_fft -> exp(-2 pi i k / n_fft), k < n_fft / 2
        int n_fft = 0, win = 0, hop = 0, T = 0;
        float* window = nullptr;
        float* rwss = nullptr;
        bool configured = false;
    } glg;

    // analysis-side tables (STFT window, mel basis)
    struct {
        int win = 0;
        float* window = nullptr;
        int sr = 0, n_fft = 0, n_mels = 0;
        float fmin = 0, fmax = 0;
        float* mel_wt = nullptr;   // [n_mels][FP]
        int* flag = nullptr;
    } an;

    // profiling
    std::vector<ProfSpan> spans;
    double prof_ms[ST_COUNT] = {0};
    int64_t prof_launches[ST_COUNT] = {0};
};


namespace tts_api {

#define HIPCHK(h, expr)                                                                         \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
            return TTS_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)


#define HIPCHK(h, expr)                                                                         \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
            return TTS_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

// Every entry point that takes a handle runs on the handle's device, whatever device is current on the calling
// thread (one process may hold handles on several GPUs, or a caller may have switched devices after tts_create);
// the caller's current device is restored on return.
struct DeviceScope {
    int prev = -1;
    bool changed = false;
    explicit DeviceScope(tts_handle_t h) {
        if (!h) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != h->device) changed = hipSetDevice(h->device) == hipSuccess;
    }
    ~DeviceScope() {
        if (changed) hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

#define WS(h, name, type, count, var)                                             \
    type* var = nullptr;                                                          \
    {                                                                             \
        void* _p = nullptr;                                                       \
        int _rc = ws_get(h, name, (size_t)(count) * sizeof(type), &_p);           \
        if (_rc != TTS_OK) return _rc;                                            \
        var = reinterpret_cast<type*>(_p);                                        \
    }

// ------------------------------------------------------------------------------------ profiling
struct ProfScope {
    tts_handle_t h;
    int idx = -1;
    ProfScope(tts_handle_t h_, int stage, int64_t launches) : h(h_) {
        if (!h->profile) return;
        ProfSpan s{};
        if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return;
        s.stage = stage;
        s.launches = launches;
        hipEventRecord(s.a, h->stream);
        h->spans.push_back(s);
        idx = (int)h->spans.size() - 1;
    }
    ~ProfScope() {
        if (idx >= 0) hipEventRecord(h->spans[idx].b, h->stream);
    }
};

struct SynthScope {   // tts_synthesize is running: the stage entry points leave the stream ordering to it
    tts_handle_t h;
    explicit SynthScope(tts_handle_t h_) : h(h_) { h->in_synthesize = true; }
    ~SynthScope() { h->in_synthesize = false; }
};

// ---- defined in api_handle.hip / api_stages.hip / api_pipeline.hip
int fail(tts_handle_t h, int code, const std::string& msg);
void build_manifest(tts_handle_t h);
int check_status(tts_handle_t h);
int sync_all(tts_handle_t h);
int graph_quiesce(tts_handle_t h);
int graph_drop(tts_handle_t h);
int ws_get(tts_handle_t h, const char* name, size_t bytes, void** out);
void prof_collect(tts_handle_t h);
GemmGroup dense_group(const float* A, int lda, const float* Wt, const float* bias, float* C, int ldc, int M, int N, int K, int act);
GemmGroup conv_group(const float* A, int Cin, int ktaps, int T, const float* Wt, const float* bias, const float* scale, const float* shift, float* C, int ldc, int coff, int M, int N, int act, int pool);
int gemm_attach_image(tts_handle_t h, GemmGroup& g, bool refresh = false);
void gemm_drop_images(tts_handle_t h);
int run_single(tts_handle_t h, const GemmGroup& g);
bool graph_runtime_ok(int* have);
int run_cbhg(tts_handle_t h, const CbhgWeights& w, const char* tag, const float* x, int B, int T, float* out, int64_t* launches);
int check_ready(tts_handle_t h);
int gl_tables(tts_handle_t h);
int device_cus(tts_handle_t h);
int stft_prepare(tts_handle_t h, int n, int win, int hop, int n_fft);
int stft_run(tts_handle_t h, const float* wav, int B, int n, int n_fft, int win, int hop, float2** out, int* Tf_out);
int gl_fp(int n_fft);
bool gl_is_streaming(int n_fft, int win, int hop);
int glg_twiddles(tts_handle_t h, int n_fft, const float2** out);
int glg_prepare(tts_handle_t h, int T, int win, int hop, int n_fft);
int gl_run_generic(tts_handle_t h, const float* mag_int, const float* init_ft, uint64_t seed, int B, int T, int n_iter, int win, int hop, int n_fft, float* wav, float* mse, bool peak_normalize);
int gl_prepare(tts_handle_t h, int T, int win, int hop, int n_fft);
int gl_run(tts_handle_t h, const float* mag_int, const float* init_ft, uint64_t seed, int B, int T, int n_iter, int win, int hop, int n_fft, float* wav, float* mse, bool peak_normalize = false, bool under_reservation = false, float2* const* phase_pair = nullptr, bool phase_ready = false, int wide_from = -1);
int standalone_begin(tts_handle_t h);
int standalone_end(tts_handle_t h);
int encoder_impl(tts_handle_t h, const int32_t* ids, int B, int Ts, float* memory);
int pd_kernel_for(tts_handle_t h, int B, int Ts, int budget);
int pd_choice(tts_handle_t h, int B, int Ts, int budget, bool pipelined);
int attention_keys(tts_handle_t h, const float* memory, int B, int Ts, float* keys);
int decoder_impl(tts_handle_t h, const float* memory, int B, int Ts, int n_steps, float* mel, float* alignments);
int postnet_impl(tts_handle_t h, const float* mel, int B, int T, float* linear, float* mag, float ref_db, float max_db, float power, int* db_flag = nullptr);
bool denorm_can_assert(float ref_db, float max_db);
int denorm_flag_arm(tts_handle_t h, int** flag);
int denorm_flag_read(tts_handle_t h);
int gl_wide_from(tts_handle_t h, int B, int Ts, int n_steps, int T, int n_iter);

#define WS(h, name, type, count, var)                                             \
    type* var = nullptr;                                                          \
    {                                                                             \
        void* _p = nullptr;                                                       \
        int _rc = ws_get(h, name, (size_t)(count) * sizeof(type), &_p);           \
        if (_rc != TTS_OK) return _rc;                                            \
        var = reinterpret_cast<type*>(_p);                                        \
    }

}  // namespace tts_api
