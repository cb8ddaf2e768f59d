// C ABI of libsstts_hip.so (see include/sstts_hip.h): handle, weights, workspace, stage drivers.
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

using namespace tts;

namespace {

thread_local std::string g_create_error;

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
const char* kStageNames[ST_COUNT] = {"encoder", "decoder", "postnet", "denorm", "gl_iter", "gl_final", "debug_gemm"};

struct ProfSpan {
    hipEvent_t a, b;
    int stage;
    int64_t launches;
};

}  // namespace

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
    // Round 5's two GEMM variants, measured and not faster (profiles/r05_experiment_gemm_presplit.txt): weights
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
    // option "deterministic" (default 0): 1 = a call's outputs are bit-identical whatever the handle ran before -- the decoder
    // already is (one kernel form's bits everywhere), this pins the Griffin-Lim run cut: the pipelined calls' cut for every
    // call, no wide launches (costs the pipelined step ~0.25 ms and an unpipelined call ~5 % of its Griffin-Lim phase)
    int deterministic = 0;
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

namespace {

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

int fail(tts_handle_t h, int code, const std::string& msg) {
    if (h) h->err = msg;
    else g_create_error = msg;
    return code;
}

// ------------------------------------------------------------------------------------ manifest
const char* kAtt = "decoder2/decoder/output_projection_wrapper/multi_rnn_cell/cell_0/attention_wrapper";
const char* kMrc = "decoder2/decoder/output_projection_wrapper/multi_rnn_cell";

std::string bn_name(int i) {
    return i == 0 ? std::string("batch_normalization") : "batch_normalization_" + std::to_string(i);
}

void add(std::vector<ManifestEntry>& m, const std::string& name, std::vector<int64_t> shape) {
    m.push_back({name, std::move(shape)});
}

void gru_entries(std::vector<ManifestEntry>& m, const std::string& scope, int n_in, int units, bool cudnn) {
    add(m, scope + "/gates/kernel", {n_in + units, 2 * units});
    add(m, scope + "/gates/bias", {2 * units});
    if (cudnn) {
        add(m, scope + "/candidate/input_projection/kernel", {n_in, units});
        add(m, scope + "/candidate/input_projection/bias", {units});
        add(m, scope + "/candidate/hidden_projection/kernel", {units, units});
        add(m, scope + "/candidate/hidden_projection/bias", {units});
    } else {
        add(m, scope + "/candidate/kernel", {n_in + units, units});
        add(m, scope + "/candidate/bias", {units});
    }
}

void cbhg_entries(std::vector<ManifestEntry>& m, const std::string& scope, int n_in, int n_banks, int n_filters,
                  const int proj[2], int hw_layers, int hw_units, int gru_units, bool cudnn) {
    for (int k = 1; k <= n_banks; ++k) {
        const std::string cs = scope + "/convolution_banks/conv-" + std::to_string(k) + "-" + std::to_string(n_filters);
        add(m, cs + "/kernel", {k, n_in, n_filters});
        add(m, cs + "/bias", {n_filters});
    }
    for (int i = 0; i < n_banks; ++i)
        for (const char* v : {"beta", "moving_mean", "moving_variance"})
            add(m, scope + "/convolution_banks/" + bn_name(i) + "/" + v, {n_filters});
    int c_in = n_banks * n_filters;
    for (int i = 0; i < 2; ++i) {
        const std::string ps =
            scope + "/projections/" + std::to_string(i + 1) + "-conv-3-" + std::to_string(proj[i]);
        add(m, ps + "/conv1d/kernel", {3, c_in, proj[i]});
        add(m, ps + "/conv1d/bias", {proj[i]});
        for (const char* v : {"gamma", "beta", "moving_mean", "moving_variance"})
            add(m, ps + "/batch_normalization/" + v, {proj[i]});
        c_in = proj[i];
    }
    add(m, scope + "/lifter/kernel", {c_in, hw_units});
    add(m, scope + "/lifter/bias", {hw_units});
    for (int l = 0; l < hw_layers; ++l)
        for (const char* g : {"H", "T"}) {
            const std::string hs = scope + "/highway_network/highway_layer_" + std::to_string(l) + "/" + g;
            add(m, hs + "/kernel", {hw_units, hw_units});
            add(m, hs + "/bias", {hw_units});
        }
    for (const char* d : {"fw", "bw"})
        gru_entries(m, scope + "/gru/" + d + "/gru_cell_" + d, hw_units, gru_units, cudnn);
}

void build_manifest(tts_handle_t h) {
    const tts_config_t& c = h->cfg;
    const bool cudnn = c.force_cudnn != 0;
    auto& m = h->manifest;
    m.clear();
    add(m, "encoder/embedding", {c.vocabulary_size, c.embedding_size});
    int n_in = c.embedding_size;
    for (int i = 0; i < 2; ++i) {
        const std::string s = "encoder/pre_net/" + std::to_string(i + 1) + "-FC-" + std::to_string(c.enc_prenet_units[i]);
        add(m, s + "/kernel", {n_in, c.enc_prenet_units[i]});
        add(m, s + "/bias", {c.enc_prenet_units[i]});
        n_in = c.enc_prenet_units[i];
    }
    cbhg_entries(m, "encoder", n_in, c.enc_n_banks, c.enc_n_filters, c.enc_proj_filters, c.n_highway_layers,
                 c.n_highway_units, c.n_gru_units, cudnn);
    const int mem = 2 * c.n_gru_units, att = c.n_attention_units;
    add(m, "decoder2/memory_layer/kernel", {mem, att});
    n_in = c.n_mels + att;
    for (int i = 0; i < 2; ++i) {
        const std::string s = std::string(kAtt) + "/pre_net/" + std::to_string(i + 1) + "-FC-" +
                              std::to_string(c.dec_prenet_units[i]);
        add(m, s + "/kernel", {n_in, c.dec_prenet_units[i]});
        add(m, s + "/bias", {c.dec_prenet_units[i]});
        n_in = c.dec_prenet_units[i];
    }
    gru_entries(m, std::string(kAtt) + "/gru_cell", n_in, att, cudnn);
    add(m, std::string(kAtt) + "/attention_layer/kernel", {att + mem, att});
    if (c.attention_mechanism == TTS_ATTENTION_LOCAL_LUONG && c.luong_local_mode == TTS_LOCAL_PREDICTIVE) {
        // tf.get_variable inside LocalLuongAttention.__call__ (reference tacotron/attention.py:247-250)
        add(m, std::string(kAtt) + "/local_luong_attention/local_v_p", {att, 1});
        add(m, std::string(kAtt) + "/local_luong_attention/local_w_p", {att, att});
    }
    for (int i = 0; i < c.n_decoder_gru_layers; ++i)
        gru_entries(m, std::string(kMrc) + "/cell_" + std::to_string(i + 1) + "/gru_cell",
                    i == 0 ? att : c.n_decoder_gru_units, c.n_decoder_gru_units, cudnn);
    add(m, "decoder2/decoder/output_projection_wrapper/kernel", {c.n_decoder_gru_units, c.n_mels * c.reduction});
    add(m, "decoder2/decoder/output_projection_wrapper/bias", {c.n_mels * c.reduction});
    // reference tacotron/model.py:388-398: the post-processing CBHG is optional; without it the final Dense takes the mel frames
    if (c.apply_post_processing)
        cbhg_entries(m, "post_process", c.n_mels, c.post_n_banks, c.post_n_filters, c.post_proj_filters,
                     c.n_highway_layers, c.n_highway_units, c.n_gru_units, cudnn);
    add(m, "dense/kernel", {c.apply_post_processing ? 2 * c.n_gru_units : c.n_mels, 1 + c.n_fft / 2});
    add(m, "dense/bias", {1 + c.n_fft / 2});
}

// ------------------------------------------------------------------------------------ packing
struct Packer {
    std::vector<float> host;   // staging for the whole arena
    size_t alloc(size_t n) {
        const size_t off = (host.size() + 63) & ~size_t(63);   // 256-byte aligned segments
        host.resize(off + n, 0.f);
        return off;
    }
};

const std::vector<float>& W(tts_handle_t h, const std::string& name) { return h->host_w.at(name); }

// [K][N] row-major (TF (in,out)) -> [N][K]
size_t pack_transposed(Packer& p, const float* src, int K, int N) {
    const size_t off = p.alloc((size_t)K * N);
    float* dst = p.host.data() + off;
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) dst[(size_t)n * K + k] = src[(size_t)k * N + n];
    return off;
}
size_t pack_copy(Packer& p, const float* src, size_t n) {
    const size_t off = p.alloc(n);
    std::memcpy(p.host.data() + off, src, n * sizeof(float));
    return off;
}

struct CbhgOffsets {
    std::vector<size_t> bank_wt, bank_b, bank_scale, bank_shift;
    size_t proj_wt[2], proj_b[2], proj_scale[2], proj_shift[2], lifter_wt, lifter_b;
    std::vector<size_t> hw_wt, hw_b;
    size_t gru_in_wt, gru_in_b, gru_rec;
};

const float kBnEps = 1e-3f;   // tf.layers.batch_normalization default epsilon

CbhgOffsets pack_cbhg(tts_handle_t h, Packer& p, const std::string& scope, int n_in, int n_banks, int n_filters,
                      const int proj[2], bool cudnn) {
    const tts_config_t& c = h->cfg;
    CbhgOffsets o;
    for (int k = 1; k <= n_banks; ++k) {
        const std::string cs = scope + "/convolution_banks/conv-" + std::to_string(k) + "-" + std::to_string(n_filters);
        // (k, in, out) is already [K = k*in][N = out] row-major
        o.bank_wt.push_back(pack_transposed(p, W(h, cs + "/kernel").data(), k * n_in, n_filters));
        o.bank_b.push_back(pack_copy(p, W(h, cs + "/bias").data(), n_filters));
        const std::string bs = scope + "/convolution_banks/" + bn_name(k - 1);
        std::vector<float> sc(n_filters), sh(n_filters);
        for (int i = 0; i < n_filters; ++i) {
            const double inv = 1.0 / std::sqrt((double)W(h, bs + "/moving_variance")[i] + (double)kBnEps);
            sc[i] = (float)inv;
            sh[i] = (float)((double)W(h, bs + "/beta")[i] - (double)W(h, bs + "/moving_mean")[i] * inv);
        }
        o.bank_scale.push_back(pack_copy(p, sc.data(), n_filters));
        o.bank_shift.push_back(pack_copy(p, sh.data(), n_filters));
    }
    int c_in = n_banks * n_filters;
    for (int i = 0; i < 2; ++i) {
        const std::string ps = scope + "/projections/" + std::to_string(i + 1) + "-conv-3-" + std::to_string(proj[i]);
        o.proj_wt[i] = pack_transposed(p, W(h, ps + "/conv1d/kernel").data(), 3 * c_in, proj[i]);
        o.proj_b[i] = pack_copy(p, W(h, ps + "/conv1d/bias").data(), proj[i]);
        std::vector<float> sc(proj[i]), sh(proj[i]);
        const std::string bs = ps + "/batch_normalization";
        for (int j = 0; j < proj[i]; ++j) {
            const double inv = (double)W(h, bs + "/gamma")[j] /
                               std::sqrt((double)W(h, bs + "/moving_variance")[j] + (double)kBnEps);
            sc[j] = (float)inv;
            sh[j] = (float)((double)W(h, bs + "/beta")[j] - (double)W(h, bs + "/moving_mean")[j] * inv);
        }
        o.proj_scale[i] = pack_copy(p, sc.data(), proj[i]);
        o.proj_shift[i] = pack_copy(p, sh.data(), proj[i]);
        c_in = proj[i];
    }
    const int U = c.n_highway_units;
    o.lifter_wt = pack_transposed(p, W(h, scope + "/lifter/kernel").data(), c_in, U);
    o.lifter_b = pack_copy(p, W(h, scope + "/lifter/bias").data(), U);
    for (int l = 0; l < c.n_highway_layers; ++l) {
        const std::string hs = scope + "/highway_network/highway_layer_" + std::to_string(l);
        const auto& kh = W(h, hs + "/H/kernel");
        const auto& kt = W(h, hs + "/T/kernel");
        const auto& bh = W(h, hs + "/H/bias");
        const auto& bt = W(h, hs + "/T/bias");
        // packed rows: span s (32 units): rows 64s + w = H unit 32s+w ; rows 64s + 32 + w = T unit 32s+w
        const size_t ow = p.alloc((size_t)2 * U * U);
        const size_t ob = p.alloc((size_t)2 * U);
        for (int u = 0; u < U; ++u) {
            const int s = u / 32, w = u % 32;
            const int rh = 64 * s + w, rt = 64 * s + 32 + w;
            for (int k = 0; k < U; ++k) {
                p.host[ow + (size_t)rh * U + k] = kh[(size_t)k * U + u];
                p.host[ow + (size_t)rt * U + k] = kt[(size_t)k * U + u];
            }
            p.host[ob + rh] = bh[u];
            p.host[ob + rt] = bt[u];
        }
        o.hw_wt.push_back(ow);
        o.hw_b.push_back(ob);
    }
    // bi-GRU: input projections [2][r|u|c] and recurrent blocks
    const int H = c.n_gru_units;
    o.gru_in_wt = p.alloc((size_t)6 * H * U);
    o.gru_in_b = p.alloc((size_t)6 * H);
    o.gru_rec = p.alloc(bigru_wrec_floats(H, cudnn));
    const size_t rec_stride = bigru_wrec_floats(H, cudnn) / 2;
    const char* dirs[2] = {"fw", "bw"};
    for (int d = 0; d < 2; ++d) {
        const std::string gs = scope + "/gru/" + dirs[d] + "/gru_cell_" + dirs[d];
        const auto& gk = W(h, gs + "/gates/kernel");   // [U + H][2H]
        const auto& gb = W(h, gs + "/gates/bias");
        const float* ck_in;    // [U][H] input part of the candidate
        int ck_in_ld;
        const float* ck_h;     // [H][H] recurrent part
        const float* cb;
        if (cudnn) {
            ck_in = W(h, gs + "/candidate/input_projection/kernel").data();
            ck_h = W(h, gs + "/candidate/hidden_projection/kernel").data();
            cb = W(h, gs + "/candidate/input_projection/bias").data();
        } else {
            ck_in = W(h, gs + "/candidate/kernel").data();
            ck_h = ck_in + (size_t)U * H;
            cb = W(h, gs + "/candidate/bias").data();
        }
        ck_in_ld = H;
        float* wt = p.host.data() + o.gru_in_wt + (size_t)d * 3 * H * U;
        float* bb = p.host.data() + o.gru_in_b + (size_t)d * 3 * H;
        for (int n = 0; n < 2 * H; ++n) {
            for (int k = 0; k < U; ++k) wt[(size_t)n * U + k] = gk[(size_t)k * 2 * H + n];
            bb[n] = gb[n];
        }
        for (int n = 0; n < H; ++n) {
            for (int k = 0; k < U; ++k) wt[(size_t)(2 * H + n) * U + k] = ck_in[(size_t)k * ck_in_ld + n];
            bb[2 * H + n] = cb[n];
        }
        float* rec = p.host.data() + o.gru_rec + (size_t)d * rec_stride;
        for (int k = 0; k < H; ++k)
            for (int n = 0; n < 2 * H; ++n) rec[(size_t)k * 2 * H + n] = gk[(size_t)(U + k) * 2 * H + n];
        float* rc = rec + (size_t)H * 2 * H;
        for (int k = 0; k < H; ++k)
            for (int n = 0; n < H; ++n) rc[(size_t)k * H + n] = ck_h[(size_t)k * H + n];
        if (cudnn) {
            const auto& hb = W(h, gs + "/candidate/hidden_projection/bias");
            for (int n = 0; n < H; ++n) rc[(size_t)H * H + n] = hb[n];
        }
    }
    return o;
}

void bind_cbhg(CbhgWeights& w, const CbhgOffsets& o, const float* base, int n_in, int n_banks, int n_filters,
               const int proj[2]) {
    w.n_banks = n_banks;
    w.n_filters = n_filters;
    w.c_in = n_in;
    w.proj_filters[0] = proj[0];
    w.proj_filters[1] = proj[1];
    for (int k = 0; k < n_banks; ++k) {
        w.bank_wt.push_back(base + o.bank_wt[k]);
        w.bank_b.push_back(base + o.bank_b[k]);
        w.bank_scale.push_back(base + o.bank_scale[k]);
        w.bank_shift.push_back(base + o.bank_shift[k]);
    }
    for (int i = 0; i < 2; ++i) {
        w.proj_wt[i] = base + o.proj_wt[i];
        w.proj_b[i] = base + o.proj_b[i];
        w.proj_scale[i] = base + o.proj_scale[i];
        w.proj_shift[i] = base + o.proj_shift[i];
    }
    w.lifter_wt = base + o.lifter_wt;
    w.lifter_b = base + o.lifter_b;
    for (size_t l = 0; l < o.hw_wt.size(); ++l) {
        w.hw_wt.push_back(base + o.hw_wt[l]);
        w.hw_b.push_back(base + o.hw_b[l]);
    }
    w.gru_in_wt = base + o.gru_in_wt;
    w.gru_in_b = base + o.gru_in_b;
    w.gru_rec = base + o.gru_rec;
}

struct GruOffsets {
    size_t gates_wt, gates_b, cand_wt, cand_b;
};

GruOffsets pack_dec_gru(tts_handle_t h, Packer& p, const std::string& scope, int n_in, int U, bool cudnn) {
    GruOffsets o{};
    const auto& gk = W(h, scope + "/gates/kernel");   // [n_in + U][2U]
    const auto& gb = W(h, scope + "/gates/bias");
    const int K = n_in + U;
    if (!cudnn) {
        o.gates_wt = pack_transposed(p, gk.data(), K, 2 * U);
        o.gates_b = pack_copy(p, gb.data(), 2 * U);
        o.cand_wt = pack_transposed(p, W(h, scope + "/candidate/kernel").data(), K, U);
        o.cand_b = pack_copy(p, W(h, scope + "/candidate/bias").data(), U);
        return o;
    }
    // [4U][K]: r | u | hh (h Wch, zero over the input rows) | xi (x Wci, zero over the state rows)
    const auto& ik = W(h, scope + "/candidate/input_projection/kernel");   // [n_in][U]
    const auto& ib = W(h, scope + "/candidate/input_projection/bias");
    const auto& hk = W(h, scope + "/candidate/hidden_projection/kernel");  // [U][U]
    const auto& hb = W(h, scope + "/candidate/hidden_projection/bias");
    o.gates_wt = p.alloc((size_t)4 * U * K);
    o.gates_b = p.alloc((size_t)4 * U);
    float* wt = p.host.data() + o.gates_wt;
    float* bb = p.host.data() + o.gates_b;
    for (int n = 0; n < 2 * U; ++n) {
        for (int k = 0; k < K; ++k) wt[(size_t)n * K + k] = gk[(size_t)k * 2 * U + n];
        bb[n] = gb[n];
    }
    for (int n = 0; n < U; ++n) {
        for (int k = 0; k < U; ++k) wt[(size_t)(2 * U + n) * K + n_in + k] = hk[(size_t)k * U + n];
        bb[2 * U + n] = hb[n];
        for (int k = 0; k < n_in; ++k) wt[(size_t)(3 * U + n) * K + k] = ik[(size_t)k * U + n];
        bb[3 * U + n] = ib[n];
    }
    o.cand_wt = o.gates_wt;
    o.cand_b = o.gates_b;
    return o;
}

// ------------------------------------------------------------------------------------ workspace
// Did every bounded wait of the persistent decoder's launches so far end by arrival?  The status word is STICKY on the
// device (no launch clears it): a timeout in call j is still there when call j + 1 has been queued behind it; the
// host clears the word when it has read it.  The caller has synchronised the streams the kernels ran on.
int check_status(tts_handle_t h) {
    if (h->pd_used) {
        h->pd_used = false;
        int status = 0;
        HIPCHK(h, hipMemcpy(&status, h->pd_sync + 64 * h->pd_clusters + 1, sizeof(int), hipMemcpyDeviceToHost));
        if (status) HIPCHK(h, hipMemset(h->pd_sync + 64 * h->pd_clusters + 1, 0, sizeof(int)));
        if (status) h->persistent_decoder = 0;   // every later call takes the launch-per-layer path by itself
        if (status)
            return fail(h, TTS_ERR_HIP,
                        "persistent decoder: a workgroup waited for its cluster longer than the bound (not all "
                        "workgroups were co-resident); the outputs of that call are invalid -- the handle has "
                        "switched to the launch-per-layer path (tts_set_option(h, \"persistent_decoder\", 1) switches back)");
    }
    return TTS_OK;
}

int sync_all(tts_handle_t h) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->front && h->front != h->stream) HIPCHK(h, hipStreamSynchronize(h->front));
    if (h->aux) HIPCHK(h, hipStreamSynchronize(h->aux));
    if (h->encs) HIPCHK(h, hipStreamSynchronize(h->encs));
    if (h->hio.in) HIPCHK(h, hipStreamSynchronize(h->hio.in));
    if (h->hio.out) HIPCHK(h, hipStreamSynchronize(h->hio.out));
    return check_status(h);
}

// the decoder graph's last launch has finished (see ev_graph_done)
int graph_quiesce(tts_handle_t h) {
    if (h->graph_in_flight) {
        HIPCHK(h, hipEventSynchronize(h->ev_graph_done));
        h->graph_in_flight = false;
    }
    return TTS_OK;
}
int graph_drop(tts_handle_t h) {
    if (h->dec_graph) {
        int rc = graph_quiesce(h);
        if (rc) return rc;
        hipGraphExecDestroy(h->dec_graph);
        h->dec_graph = nullptr;
    }
    if (h->dec_graph_src) {
        hipGraphDestroy(h->dec_graph_src);
        h->dec_graph_src = nullptr;
    }
    return TTS_OK;
}

int ws_get(tts_handle_t h, const char* name, size_t bytes, void** out) {
    DevBuf& b = h->ws[name];
    if (b.bytes < bytes) {
        if (b.p) {
            {
                int rc = sync_all(h);
                if (rc) return rc;
            }
            HIPCHK(h, hipFree(b.p));
            b.p = nullptr;
            b.bytes = 0;
            // pointers baked into the decoder graph may have changed
            {
                int rc = graph_drop(h);
                if (rc) return rc;
            }
        }
        HIPCHK(h, hipMalloc(&b.p, bytes));
        b.bytes = bytes;
    }
    *out = b.p;
    return TTS_OK;
}
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

void prof_collect(tts_handle_t h) {
    sync_all(h);
    if (!h->spans.empty() && h->debug_hooks && h->timeline) {   // diagnostic (option "timeline"): absolute stage times of every span
        for (auto& s : h->spans) {
            float t0 = 0.f, t1 = 0.f;
            if (hipEventElapsedTime(&t0, h->spans[0].a, s.a) == hipSuccess &&
                hipEventElapsedTime(&t1, h->spans[0].a, s.b) == hipSuccess)
                fprintf(stderr, "timeline %-8s %9.3f -> %9.3f ms (%.3f)\n", kStageNames[s.stage], t0, t1, t1 - t0);
        }
    }
    for (auto& s : h->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            h->prof_ms[s.stage] += ms;
            h->prof_launches[s.stage] += s.launches;
        }
        hipEventDestroy(s.a);
        hipEventDestroy(s.b);
    }
    h->spans.clear();
}

// ------------------------------------------------------------------------------------ GEMM helpers
GemmGroup dense_group(const float* A, int lda, const float* Wt, const float* bias, float* C, int ldc, int M, int N,
                      int K, int act) {
    GemmGroup g;
    std::memset(&g, 0, sizeof(g));
    g.A = A; g.Wt = Wt; g.bias = bias; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.T = M; g.Cin = K; g.padl = 0; g.pool = 0;
    g.ldc = ldc; g.coff = 0; g.act = act; g.epi = EPI_STD;
    return g;
}

GemmGroup conv_group(const float* A, int Cin, int ktaps, int T, const float* Wt, const float* bias,
                     const float* scale, const float* shift, float* C, int ldc, int coff, int M, int N, int act,
                     int pool) {
    GemmGroup g;
    std::memset(&g, 0, sizeof(g));
    g.A = A; g.Wt = Wt; g.bias = bias; g.scale = scale; g.shift = shift; g.C = C;
    g.M = M; g.N = N; g.K = ktaps * Cin;
    g.lda = Cin; g.T = T; g.Cin = Cin; g.padl = (ktaps - 1) / 2; g.pool = pool;
    g.ldc = ldc; g.coff = coff; g.act = act; g.epi = EPI_STD;
    return g;
}

// Attach the pre-split image of g.Wt (made now if this weight matrix has none yet for this (N, K, Cin); `refresh`: made
// again whatever the cache holds -- tts_debug_gemm, whose caller owns the weights and may have rewritten them).
int gemm_attach_image(tts_handle_t h, GemmGroup& g, bool refresh = false) {
    g.Wimg = nullptr;
    if (!h->gemm_presplit) return TTS_OK;
    auto& im = h->wimg[g.Wt];
    const size_t bytes = gemm_weight_image_bytes(g.N, g.K);
    const bool fresh = im.p == nullptr || im.N != g.N || im.K != g.K || im.Cin != g.Cin;
    if (fresh || refresh) {
        if (im.bytes < bytes) {
            if (im.p) {
                int rc = sync_all(h);   // (a launch that reads the old image may be in flight)
                if (rc) return rc;
                HIPCHK(h, hipFree(im.p));
                im.p = nullptr; im.bytes = 0;
            }
            HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&im.p), bytes));
            im.bytes = bytes;
        }
        im.N = g.N; im.K = g.K; im.Cin = g.Cin;
        HIPCHK(h, launch_gemm_pack_weights(h->stream, g.Wt, im.p, g.N, g.K, g.Cin));
        // a new image is complete before any stream may use it (the first call of a shape runs unpipelined and makes them
        // all; later calls find them in the cache)
        if (fresh) HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    g.Wimg = im.p;
    return TTS_OK;
}
void gemm_drop_images(tts_handle_t h) {
    for (auto& kv : h->wimg)
        if (kv.second.p) hipFree(kv.second.p);
    h->wimg.clear();
}

int run_single(tts_handle_t h, const GemmGroup& g) {
    GemmBatch b;
    std::memset(&b, 0, sizeof(b));
    b.g[0] = g;
    b.ps = h->gemm_ps;
    {
        int rc = gemm_attach_image(h, b.g[0]);
        if (rc) return rc;
    }
    HIPCHK(h, launch_gemm(h->stream, b, 1));
    return TTS_OK;
}

// CBHG (reference tacotron/layers.py:448-594) on x [B*T][c_in] -> out [B*T][2H].  Returns launches.
int run_cbhg(tts_handle_t h, const CbhgWeights& w, const char* tag, const float* x, int B, int T, float* out,
             int64_t* launches) {
    const tts_config_t& c = h->cfg;
    const int M = B * T;
    const int NB = w.n_banks, NF = w.n_filters;
    const int U = c.n_highway_units, H = c.n_gru_units;
    const std::string t(tag);
    WS(h, (t + ".bank").c_str(), float, (size_t)M * NB * NF, bank);
    WS(h, (t + ".p1").c_str(), float, (size_t)M * w.proj_filters[0], p1);
    WS(h, (t + ".p2").c_str(), float, (size_t)M * w.proj_filters[1], p2);
    WS(h, (t + ".hw0").c_str(), float, (size_t)M * U, hw0);
    WS(h, (t + ".hw1").c_str(), float, (size_t)M * U, hw1);
    WS(h, (t + ".xproj").c_str(), float, (size_t)M * 6 * H, xproj);

    // conv bank: one grouped launch, bank k writes channels [k*NF, (k+1)*NF)
    for (int k0 = 0; k0 < NB; k0 += TTS_GEMM_MAX_GROUPS) {
        GemmBatch b;
        std::memset(&b, 0, sizeof(b));
        const int ng = std::min(TTS_GEMM_MAX_GROUPS, NB - k0);
        // widest bank first: the groups are dispatched in order (blockIdx.z slowest), and a launch that ends with its
        // cheapest tiles (k = 1: 80 or 128 deep) has a shorter tail than one that ends with the k = 8 / 16 ones
        for (int i = 0; i < ng; ++i) {
            const int k = k0 + ng - 1 - i;
            b.g[i] = conv_group(x, w.c_in, k + 1, T, w.bank_wt[k], w.bank_b[k], w.bank_scale[k], w.bank_shift[k], bank,
                                NB * NF, k * NF, M, NF, ACT_RELU, 0);
        }
        for (int i = 0; i < ng; ++i) {
            int rc = gemm_attach_image(h, b.g[i]);
            if (rc) return rc;
        }
        b.ps = h->gemm_ps;
        HIPCHK(h, launch_gemm(h->stream, b, ng));
        ++*launches;
    }
    // projection 1: max-pool(2,1,SAME) fused into the loader, conv3 + relu + BN.  With few output tiles (the
    // encoder: 75 x 1 for 32 x 150 tokens, K = 6144) the K range is split over several workgroups per tile.
    {
        GemmGroup g = conv_group(bank, NB * NF, 3, T, w.proj_wt[0], w.proj_b[0], w.proj_scale[0], w.proj_shift[0], p1,
                                 w.proj_filters[0], 0, M, w.proj_filters[0], ACT_RELU, 1);
        const int slices = gemm_splitk_slices(g.K);
        if (slices > 1) {
            WS(h, (t + ".splitk").c_str(), float, (size_t)slices * M * g.N, part);
            {
                int rc = gemm_attach_image(h, g);
                if (rc) return rc;
            }
            HIPCHK(h, launch_gemm_splitk(h->stream, g, slices, part, h->gemm_ps));
            ++*launches;
        } else {
            int rc = run_single(h, g);
            if (rc) return rc;
        }
        ++*launches;
    }
    // projection 2: conv3 + BN (linear) + residual with the CBHG input
    {
        GemmGroup g = conv_group(p1, w.proj_filters[0], 3, T, w.proj_wt[1], w.proj_b[1], w.proj_scale[1],
                                 w.proj_shift[1], p2, w.proj_filters[1], 0, M, w.proj_filters[1], ACT_NONE, 0);
        g.R = x;
        g.ldr = w.c_in;
        int rc = run_single(h, g);
        if (rc) return rc;
        ++*launches;
    }
    if (h->fused_tail && cbhg_tail_supports(w.proj_filters[1], U, H, (int)w.hw_wt.size(), M)) {
        // lifter, highway stack and the GRU input projections in one launch: the rows stay in LDS between the layers
        if (!h->tail_configured) {
            HIPCHK(h, cbhg_tail_configure());
            h->tail_configured = true;
        }
        CbhgTailParams tp;
        std::memset(&tp, 0, sizeof(tp));
        tp.X = p2; tp.ldx = w.proj_filters[1]; tp.c_in = w.proj_filters[1];
        tp.lifter_wt = w.lifter_wt; tp.lifter_b = w.lifter_b;
        tp.n_hw = (int)w.hw_wt.size();
        for (int l = 0; l < tp.n_hw; ++l) { tp.hw_wt[l] = w.hw_wt[l]; tp.hw_b[l] = w.hw_b[l]; }
        tp.gru_wt = w.gru_in_wt; tp.gru_b = w.gru_in_b;
        tp.hw_out = hw0; tp.xproj = xproj; tp.M = M;
        HIPCHK(h, launch_cbhg_tail(h->stream, tp));
        ++*launches;
    } else {
        // lifter
        {
            int rc = run_single(h, dense_group(p2, w.proj_filters[1], w.lifter_wt, w.lifter_b, hw0, U, M, U,
                                               w.proj_filters[1], ACT_RELU));
            if (rc) return rc;
            ++*launches;
        }
        // highway layers (H|T in one GEMM, gate mix in the epilogue), ping-pong buffers
        float* cur = hw0;
        float* nxt = hw1;
        for (size_t l = 0; l < w.hw_wt.size(); ++l) {
            GemmGroup g = dense_group(cur, U, w.hw_wt[l], w.hw_b[l], nxt, U, M, 2 * U, U, ACT_NONE);
            g.epi = EPI_HIGHWAY;
            int rc = run_single(h, g);
            if (rc) return rc;
            ++*launches;
            std::swap(cur, nxt);
        }
        // GRU input projections for both directions, then the recurrent kernel
        {
            int rc = run_single(h, dense_group(cur, U, w.gru_in_wt, w.gru_in_b, xproj, 6 * H, M, 6 * H, U, ACT_NONE));
            if (rc) return rc;
            ++*launches;
        }
    }
    HIPCHK(h, launch_bigru(h->stream, xproj, 6 * H, w.gru_rec, out, B, T, H, c.force_cudnn));
    ++*launches;
    return TTS_OK;
}

int check_ready(tts_handle_t h) {
    if (!h) return TTS_ERR_INVALID;
    if (!h->finalized) return fail(h, TTS_ERR_NOT_LOADED, "weights not loaded: call tts_finalize_weights first");
    return TTS_OK;
}

// ------------------------------------------------------------------------------------ Griffin-Lim
int gl_tables(tts_handle_t h) {
    auto& g = h->gl;
    if (g.configured) return TTS_OK;
    HIPCHK(h, gl_configure());
    std::vector<float2> t1(1024), t2(1024);
    for (int k = 0; k < 1024; ++k) {
        const double a1 = -2.0 * M_PI * k / 1024.0, a2 = -2.0 * M_PI * k / 2048.0;
        t1[k] = make_float2((float)std::cos(a1), (float)std::sin(a1));
        t2[k] = make_float2((float)std::cos(a2), (float)std::sin(a2));
    }
    HIPCHK(h, hipMalloc(&g.tw1024, 1024 * sizeof(float2)));
    HIPCHK(h, hipMalloc(&g.tw2048, 1024 * sizeof(float2)));
    HIPCHK(h, hipMemcpy(g.tw1024, t1.data(), 1024 * sizeof(float2), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(g.tw2048, t2.data(), 1024 * sizeof(float2), hipMemcpyHostToDevice));
    {
        std::vector<float2> tb(1024 + 15 * 64);
        for (int k = 0; k < 1024; ++k) tb[k] = t2[k];
        for (int i = 0; i < 15 * 64; ++i) tb[1024 + i] = t1[(i & 63) * ((i >> 6) + 1)];
        HIPCHK(h, hipMalloc(&g.tables, tb.size() * sizeof(float2)));
        HIPCHK(h, hipMemcpy(g.tables, tb.data(), tb.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    {
        hipDeviceProp_t prop;
        HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
        g.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    g.configured = true;
    return TTS_OK;
}

static inline int device_cus(tts_handle_t h) { return h->gl.n_cus; }

static inline int gl_fp(int n_fft);
int glg_prepare(tts_handle_t h, int T, int win, int hop, int n_fft);
int glg_twiddles(tts_handle_t h, int n_fft, const float2** out);

int stft_prepare(tts_handle_t h, int n, int win, int hop, int n_fft) {
    if (n_fft != TTS_GL_NFFT) return fail(h, TTS_ERR_UNSUPPORTED, "stft: n_fft != 2048 takes the general kernels (stft_run)");
    if (win < 2 || win > n_fft || hop < 1) return fail(h, TTS_ERR_INVALID, "stft: need 2 <= win_length <= n_fft, hop >= 1");
    if (n <= n_fft / 2) return fail(h, TTS_ERR_INVALID, "stft: signal shorter than n_fft/2 (reflect padding undefined)");
    int rc = gl_tables(h);
    if (rc) return rc;
    auto& a = h->an;
    if (a.win != win) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (a.window) hipFree(a.window);
        a.window = nullptr;
        std::vector<float> wf(win);
        for (int i = 0; i < win; ++i) wf[i] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * i / win));
        HIPCHK(h, hipMalloc(&a.window, win * sizeof(float)));
        HIPCHK(h, hipMemcpy(a.window, wf.data(), win * sizeof(float), hipMemcpyHostToDevice));
        a.win = win;
    }
    return TTS_OK;
}

int stft_run(tts_handle_t h, const float* wav, int B, int n, int n_fft, int win, int hop, float2** out, int* Tf_out) {
    if (n_fft != TTS_GL_NFFT) {   // the general kernels: one workgroup per frame, FFT in LDS (griffin_lim_generic.hip)
        if (n <= n_fft / 2) return fail(h, TTS_ERR_INVALID, "stft: signal shorter than n_fft/2 (reflect padding undefined)");
        int rc = glg_prepare(h, 0, win, hop, n_fft);
        if (rc) return rc;
        const float2* tw = nullptr;
        if ((rc = glg_twiddles(h, n_fft, &tw))) return rc;
        const int Tf = 1 + n / hop, Fp = gl_fp(n_fft);
        WS(h, "an.stft", float2, (size_t)B * Tf * Fp, buf);
        HIPCHK(h, launch_glg_stft(h->stream, wav, n, h->glg.window, tw, buf, B, Tf, Fp, n_fft, win, hop, 1, nullptr, nullptr));
        *out = buf;
        *Tf_out = Tf;
        return TTS_OK;
    }
    int rc = stft_prepare(h, n, win, hop, n_fft);
    if (rc) return rc;
    const int Tf = 1 + n / hop;
    WS(h, "an.stft", float2, (size_t)B * Tf * TTS_GL_FP, buf);
    HIPCHK(h, launch_stft(h->stream, wav, B, n, Tf, h->an.window, win, hop, h->gl.tw1024, h->gl.tw2048, buf, TTS_GL_FP));
    *out = buf;
    *Tf_out = Tf;
    return TTS_OK;
}

// ---- general path: any power-of-two n_fft, any window / hop (griffin_lim_generic.hip)
static inline int gl_fp(int n_fft) { return ((n_fft / 2 + 1) + 31) & ~31; }   // padded row length (TTS_GL_FP for 2048)
// The streaming kernel is specialised to the model's configuration; everything else takes the general kernels.
static inline bool gl_is_streaming(int n_fft, int win, int hop) { return n_fft == TTS_GL_NFFT && win == 1102 && hop == 275; }

// periodic hann (scipy get_window('hann', win, fftbins=True)), float64 then float32
static void hann_window(int win, std::vector<double>& wd, std::vector<float>& wf) {
    wd.resize(win);
    wf.resize(win);
    for (int i = 0; i < win; ++i) {
        wd[i] = 0.5 - 0.5 * std::cos(2.0 * M_PI * i / win);
        wf[i] = (float)wd[i];
    }
}
// librosa window_sumsquare (float32 buffer, sequential += of the padded squared window) as its RECIPROCAL where librosa's
// istft divides (wss > tiny(float32)), 1 elsewhere
static void recip_window_sumsquare(const std::vector<double>& wd, int n_fft, int hop, int T, std::vector<float>& wss) {
    const int win = (int)wd.size();
    const size_t n = (size_t)n_fft + (size_t)hop * (T - 1);
    wss.assign(n, 0.f);
    const int lpad = (n_fft - win) / 2;
    for (int i = 0; i < T; ++i) {
        const size_t s0 = (size_t)i * hop;
        for (int j = 0; j < win; ++j) {
            const size_t idx = s0 + lpad + j;
            if (idx < n) wss[idx] = (float)((double)wss[idx] + wd[j] * wd[j]);
        }
    }
    for (size_t i = 0; i < n; ++i) wss[i] = wss[i] > 1.17549435e-38f ? (float)(1.0 / (double)wss[i]) : 1.0f;
}

int glg_twiddles(tts_handle_t h, int n_fft, const float2** out) {
    auto& g = h->glg;
    if (!g.configured) {
        HIPCHK(h, glg_configure());
        g.configured = true;
    }
    auto it = g.tw.find(n_fft);
    if (it == g.tw.end()) {
        std::vector<float2> t(n_fft / 2);
        for (int k = 0; k < n_fft / 2; ++k) {
            const double a = -2.0 * M_PI * k / (double)n_fft;
            t[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        float2* d = nullptr;
        HIPCHK(h, hipMalloc(&d, t.size() * sizeof(float2)));
        HIPCHK(h, hipMemcpy(d, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice));
        it = g.tw.emplace(n_fft, d).first;
    }
    *out = it->second;
    return TTS_OK;
}

// window tables of a configuration (T = 0: the analysis side needs the window only)
int glg_prepare(tts_handle_t h, int T, int win, int hop, int n_fft) {
    if (!glg_supports(n_fft))
        return fail(h, TTS_ERR_UNSUPPORTED, "n_fft must be a power of two between 256 and 4096");
    if (win < 2 || win > n_fft || hop < 1) return fail(h, TTS_ERR_INVALID, "need 2 <= win_length <= n_fft, hop_length >= 1");
    auto& g = h->glg;
    if (g.n_fft == n_fft && g.win == win && g.hop == hop && (T == 0 || g.T == T)) return TTS_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (g.window) hipFree(g.window);
    if (g.rwss) hipFree(g.rwss);
    g.window = g.rwss = nullptr;
    g.n_fft = 0;
    std::vector<double> wd;
    std::vector<float> wf;
    hann_window(win, wd, wf);
    HIPCHK(h, hipMalloc(&g.window, win * sizeof(float)));
    HIPCHK(h, hipMemcpy(g.window, wf.data(), win * sizeof(float), hipMemcpyHostToDevice));
    if (T > 0) {
        std::vector<float> wss;
        recip_window_sumsquare(wd, n_fft, hop, T, wss);
        HIPCHK(h, hipMalloc(&g.rwss, wss.size() * sizeof(float)));
        HIPCHK(h, hipMemcpy(g.rwss, wss.data(), wss.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    g.n_fft = n_fft; g.win = win; g.hop = hop; g.T = T;
    return TTS_OK;
}

// mag_int: [B][T][Fp] (Fp = gl_fp(n_fft)); init_ft: reference-layout U[0,1) numbers or null (then the seed)
int gl_run_generic(tts_handle_t h, const float* mag_int, const float* init_ft, uint64_t seed, int B, int T, int n_iter, int win,
                   int hop, int n_fft, float* wav, float* mse, bool peak_normalize) {
    if (T < 1) return fail(h, TTS_ERR_INVALID, "griffin_lim: T >= 1");
    if ((long long)hop * (T - 1) <= n_fft / 2)
        return fail(h, TTS_ERR_INVALID, "griffin_lim: signal shorter than n_fft/2 (reflect padding undefined)");
    int rc = glg_prepare(h, T, win, hop, n_fft);
    if (rc) return rc;
    const float2* tw = nullptr;
    if ((rc = glg_twiddles(h, n_fft, &tw))) return rc;
    const int F = 1 + n_fft / 2, Fp = gl_fp(n_fft), L = hop * (T - 1);
    WS(h, "glg.phase", float2, (size_t)B * T * Fp, ph);
    WS(h, "glg.frames", float, (size_t)B * T * win, frames);
    WS(h, "glg.mse_partial", float, (size_t)B * T, msep);
    float* sig = wav;   // every iteration's signal estimate lives in the caller's buffer: the last one is the result
    HIPCHK(h, launch_glg_phase_init(h->stream, init_ft, seed, ph, B, F, T, Fp));
    {
        ProfScope ps(h, ST_GL_ITER, 3 * (int64_t)n_iter);
        for (int it = 0; it < n_iter; ++it) {
            HIPCHK(h, launch_glg_istft(h->stream, mag_int, ph, h->glg.window, h->glg.rwss, tw, frames, sig, B, T, Fp, n_fft, win, hop));
            const bool want_mse = mse && it == n_iter - 1;
            HIPCHK(h, launch_glg_stft(h->stream, sig, L, h->glg.window, tw, ph, B, T, Fp, n_fft, win, hop, 0, mag_int,
                                      want_mse ? msep : nullptr));
        }
    }
    if (mse) {
        if (n_iter > 0) HIPCHK(h, launch_gl_mse_reduce(h->stream, msep, B, T, (float)((double)F * T), mse));
        else HIPCHK(h, hipMemsetAsync(mse, 0, B * sizeof(float), h->stream));
    }
    {
        ProfScope ps(h, ST_GL_FINAL, 2);
        HIPCHK(h, launch_glg_istft(h->stream, mag_int, ph, h->glg.window, h->glg.rwss, tw, frames, wav, B, T, Fp, n_fft, win, hop));
    }
    if (peak_normalize) HIPCHK(h, launch_peak_normalize(h->stream, wav, B, L));
    return TTS_OK;
}

int gl_prepare(tts_handle_t h, int T, int win, int hop, int n_fft) {
    if (!gl_is_streaming(n_fft, win, hop)) return fail(h, TTS_ERR_UNSUPPORTED, "griffin_lim: the streaming kernel runs the model's configuration only");
    if (win < 2 || win > n_fft || hop < 1 || T < 1)
        return fail(h, TTS_ERR_INVALID, "griffin_lim: need 2 <= win_length <= n_fft, hop_length >= 1, T >= 1");
    const int ncol = (win + hop - 1) / hop;
    if (ncol > 8) return fail(h, TTS_ERR_UNSUPPORTED, "griffin_lim: win_length / hop_length > 8 not supported");
    if ((long long)hop * (T - 1) <= n_fft / 2)
        return fail(h, TTS_ERR_INVALID, "griffin_lim: signal shorter than n_fft/2 (reflect padding undefined)");
    auto& g = h->gl;
    {
        int rc = gl_tables(h);
        if (rc) return rc;
    }
    if (g.win == win && g.hop == hop && g.T == T) return TTS_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (g.window) hipFree(g.window);
    if (g.wss) hipFree(g.wss);
    if (g.wlane) hipFree(g.wlane);
    g.window = g.wss = g.wlane = nullptr;
    // periodic hann (scipy get_window('hann', win, fftbins=True)), float64 then float32
    std::vector<double> wd(win);
    std::vector<float> wf(win);
    for (int i = 0; i < win; ++i) {
        wd[i] = 0.5 - 0.5 * std::cos(2.0 * M_PI * i / win);
        wf[i] = (float)wd[i];
    }
    // librosa window_sumsquare: float32 buffer, sequential += of the padded squared window
    const size_t n = (size_t)n_fft + (size_t)hop * (T - 1);
    std::vector<float> wss(n, 0.f);
    const int lpad = (n_fft - win) / 2;
    for (int i = 0; i < T; ++i) {
        const size_t s = (size_t)i * hop;
        for (int j = 0; j < win; ++j) {
            const size_t idx = s + lpad + j;
            if (idx < n) wss[idx] = (float)((double)wss[idx] + wd[j] * wd[j]);
        }
    }
    // the kernels multiply: 1 / wss where librosa's istft divides (wss > tiny(float32)), 1 elsewhere
    for (size_t i = 0; i < n; ++i) wss[i] = wss[i] > 1.17549435e-38f ? (float)(1.0 / (double)wss[i]) : 1.0f;
    HIPCHK(h, hipMalloc(&g.window, win * sizeof(float)));
    HIPCHK(h, hipMalloc(&g.wss, n * sizeof(float)));
    HIPCHK(h, hipMemcpy(g.window, wf.data(), win * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(g.wss, wss.data(), n * sizeof(float), hipMemcpyHostToDevice));
    {
        std::vector<float> wl(2 * 16 * 2 * 64);
        gl_build_wlane(wf.data(), wss.data(), win, hop, T, wl.data());
        HIPCHK(h, hipMalloc(&g.wlane, wl.size() * sizeof(float)));
        HIPCHK(h, hipMemcpy(g.wlane, wl.data(), wl.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    g.win = win;
    g.hop = hop;
    g.T = T;
    return TTS_OK;
}

// mag_int: internal [B][T][FP]; init_ft: reference-layout U[0,1) numbers or null.
// phase_pair: the two phasor-code buffers to iterate in (null: the handle's own pair); phase_ready: phase_pair[0] already
// holds the initial phasors (written on another stream, ordered by the caller's events).
int gl_run(tts_handle_t h, const float* mag_int, const float* init_ft, uint64_t seed, int B, int T, int n_iter,
           int win, int hop, int n_fft, float* wav, float* mse, bool peak_normalize = false,
           bool under_reservation = false, float2* const* phase_pair = nullptr, bool phase_ready = false,
           int wide_from = -1) {
    int rc = gl_prepare(h, T, win, hop, n_fft);
    if (rc) return rc;
    const int F = 1 + n_fft / 2, FP = TTS_GL_FP;
    float2 *ph0, *ph1;
    if (phase_pair) {
        ph0 = phase_pair[0]; ph1 = phase_pair[1];
    } else {   // 4 bytes per bin: the state between iterations is a 32-bit phasor code (griffin_lim.hip)
        WS(h, "gl.phase0", unsigned, (size_t)B * T * FP * (gl_state_bytes() / sizeof(unsigned)), own0);
        WS(h, "gl.phase1", unsigned, (size_t)B * T * FP * (gl_state_bytes() / sizeof(unsigned)), own1);
        ph0 = reinterpret_cast<float2*>(own0); ph1 = reinterpret_cast<float2*>(own1);
    }
    GlParams p;
    std::memset(&p, 0, sizeof(p));
    p.mag = mag_int;
    p.window = h->gl.window;
    p.rwss = h->gl.wss;
    p.wlane = h->gl.wlane;
    p.tw1024 = h->gl.tw1024;
    p.tw2048 = h->gl.tw2048;
    p.tables = h->gl.tables;
    p.T = T; p.FP = FP; p.win = win; p.hop = hop;
    p.ncol = (win + hop - 1) / hop;
    p.B = B;
    if (gl_stream_ring_frames(win, hop) < 1)
        return fail(h, TTS_ERR_UNSUPPORTED, "griffin_lim: this window / hop pair does not fit the LDS ring (hop beyond the window's 128-sample slots, or too long)");
    const int n_cus = (h->debug_hooks && h->gl_workers >= 16 && h->gl_workers <= device_cus(h)) ? h->gl_workers : device_cus(h);   // (tools: "gl_workers")
    // workgroups that really run side by side: the pipelined tts_synthesize keeps `reserve_cus` compute units
    // free of Griffin-Lim for its second stream
    const int held = (under_reservation && h->reserve_cus > 0) ? h->reserve_cus : 0;
    // gl_pair = iterations per launch (1, 2 or 3; default in the handle); the run cut is made for that launch form
    int per_launch = h->gl_pair;
    per_launch = per_launch < 1 ? 1 : (per_launch > 3 ? 3 : per_launch);
    while (per_launch > 1 && gl_stream_ring_frames(win, hop, per_launch) <= 0) --per_launch;
    // option "deterministic": ONE cut for every call of a shape, pipelined or not -- the cut of the pipelined calls (all but
    // `reserve_cus` workgroups), never the second, wide one; an unpipelined call then runs that cut on all compute units
    // (the cut decides the overlap-add order, the number of workgroups that draw its items does not)
    const int plan_held = (h->deterministic && h->reserve_cus > 0) ? h->reserve_cus : held;
    gl_plan_stream(p, n_cus - plan_held > 16 ? n_cus - plan_held : n_cus, per_launch, h->debug_hooks ? h->gl_runs : 0,
                   h->debug_hooks ? h->gl_run_len : 0);
    // wide_from >= 0 (the pipelined tts_synthesize, see gl_wide_from there): launches from that index on are cut for ALL
    // compute units -- the second stream's decoder has left its share by then.  A second cut, fixed per launch index, so
    // the waveform's bits stay a function of the call's arguments and options alone.
    GlParams pw = p;
    const bool two_cuts = held > 0 && wide_from >= 0 && n_cus - held > 16 && !h->deterministic &&
                          !(h->debug_hooks && (h->gl_runs || h->gl_run_len));
    if (two_cuts) gl_plan_stream(pw, n_cus, per_launch, 0, 0);
    const int nchunks = std::max(p.slots_per_utt, pw.slots_per_utt);
    WS(h, "gl.mse_partial", float, (size_t)B * nchunks, msep);
    // One zeroed work counter per launch (the persistent workgroups draw their item ids from it): slots of a ring that is
    // zeroed ONCE; a launch takes the next slot and zeroes the slot of the launch before it on the stream, which is drained
    // by then.  (Until round 4 a memset per call: two fill kernels and their dependencies, 0.1 ms between the post-net and
    // the first Griffin-Lim launch of every call, on the stream that bounds the step.)
    constexpr unsigned GL_RING = 256;
    WS(h, "gl.counter_ring", unsigned, GL_RING, ring);
    if (ring != h->gl_ring || h->gl_ring_stream != h->stream) {   // new buffer, or launches of another stream before these
        HIPCHK(h, hipMemsetAsync(ring, 0, GL_RING * sizeof(unsigned), h->stream));
        h->gl_ring = ring;
        h->gl_ring_stream = h->stream;
        h->gl_ring_last = nullptr;
    }
    auto next_counter = [&](GlParams& q) {
        q.clear_counter = h->gl_ring_last;
        q.work_counter = ring + (h->gl_ring_seq++ % GL_RING);
        h->gl_ring_last = q.work_counter;
    };
    // a seeded start with at least one iteration needs no codes: the first launch makes the initial phasors itself
    const bool seed_in_kernel = !init_ft && n_iter >= 1;
    if (!phase_ready && !seed_in_kernel) HIPCHK(h, launch_phase_init(h->stream, init_ft, seed, ph0, B, F, T, FP));
    p.F = pw.F = F;
    p.seed = pw.seed = seed;
    int launch_idx = 0, mse_chunks = p.slots_per_utt, peak_chunks = p.slots_per_utt;
    float2* cur = ph0;
    float2* nxt = ph1;
    const int free_cus = n_cus - held > 16 ? n_cus - held : n_cus;
    {
        ProfScope ps(h, ST_GL_ITER, n_iter);
        // the streaming kernel runs two iterations per launch (gl_stream_kernel, NST = 2) wherever no per-iteration result
        // is asked for: all of them, or all but the last (the mse is the last iteration's)
        for (int it = 0; it < n_iter;) {
            const bool want_mse = mse && it == n_iter - 1;
            const int left = n_iter - (mse ? 1 : 0) - it;   // iterations that may share a launch
            const int n_stage = left >= per_launch ? per_launch : (left >= 1 ? left : 1);
            const bool wide = two_cuts && launch_idx >= wide_from;
            GlParams& q = wide ? pw : p;
            q.phase_in = cur;
            q.phase_out = nxt;
            q.seeded = seed_in_kernel && it == 0;
            q.mse_partial = want_mse ? msep : nullptr;
            if (want_mse) mse_chunks = q.slots_per_utt;
            next_counter(q);
#ifdef GL_TIMELINE   // tools only: stamps of workgroup 0 during the last launch
            WS(h, "gl.timeline", unsigned long long, 1024 + 64 * 16, tl);
            if (it + n_stage >= n_iter) {
                HIPCHK(h, hipMemsetAsync(tl, 0, (1024 + 64 * 16) * sizeof(unsigned long long), h->stream));
                p.dbg = pw.dbg = tl;
            }
#endif
            // no more workgroups than the plan counts on: one that finds its compute unit taken (the call pipeline's other
            // stream) would start when the first of the others leaves, load its tables, find no item and only
            // lengthen the launch
            HIPCHK(h, launch_gl_stream(h->stream, q, wide ? n_cus : free_cus, 0, n_stage));
            std::swap(cur, nxt);
            it += n_stage;
            ++launch_idx;
        }
#ifdef GL_TIMELINE
        if (n_iter > 0) {
            std::vector<unsigned long long> host(1024 + 64 * 16);
            HIPCHK(h, hipMemcpyAsync(host.data(), p.dbg, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            {   // per-wave stamps of workgroup 0
                unsigned long long w0 = ~0ull;
                for (int i = 1024; i < 1024 + 64 * 16; ++i) if (host[i] && host[i] < w0) w0 = host[i];
                for (int w = 0; w < 16; ++w) {
                    bool any = false;
                    for (int i = 0; i < 64; ++i) any = any || host[1024 + w * 64 + i];
                    if (!any) continue;
                    fprintf(stderr, "wave %2d:", w);
                    for (int i = 0; i < 64; ++i) {
                        const unsigned long long v = host[1024 + w * 64 + i];
                        if (v) fprintf(stderr, " [%d]%.1f", i, (double)(v - w0) * 0.01);
                    }
                    fprintf(stderr, "\n");
                }
            }
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int w = 0; w < 512; ++w) if (host[2 * w]) { t0 = std::min(t0, host[2 * w]); t1 = std::max(t1, host[2 * w + 1]); }
            std::vector<double> ends, starts;
            for (int w = 0; w < 512; ++w) if (host[2 * w]) { starts.push_back((host[2 * w] - t0) * 0.01); ends.push_back((host[2 * w + 1] - t0) * 0.01); }
            std::sort(ends.begin(), ends.end());
            std::sort(starts.begin(), starts.end());
            const size_t n = ends.size();
            double mean = 0; for (double e : ends) mean += e; mean /= n ? n : 1;
            fprintf(stderr, "workgroups %zu: start last %.1f us; end min %.1f p10 %.1f median %.1f mean %.1f p90 %.1f max %.1f us\n", n,
                    starts.back(), ends.front(), ends[n / 10], ends[n / 2], mean, ends[n * 9 / 10], ends.back());
            p.dbg = pw.dbg = nullptr;
        }
#endif
    }
    if (mse) {
        if (n_iter > 0) {
            HIPCHK(h, launch_gl_mse_reduce(h->stream, msep, B, mse_chunks, (float)((double)F * T), mse));
        } else {
            HIPCHK(h, hipMemsetAsync(mse, 0, B * sizeof(float), h->stream));
        }
    }
    {
        ProfScope ps(h, ST_GL_FINAL, 1);
        const bool wide = two_cuts && launch_idx >= wide_from;
        GlParams& q = wide ? pw : p;
        q.seeded = 0;
        q.phase_in = cur;
        q.phase_out = nullptr;
        q.mse_partial = nullptr;
        q.wav = wav;
        q.peak_partial = peak_normalize ? msep : nullptr;   // the mse partials are consumed by now
        next_counter(q);
        HIPCHK(h, launch_gl_stream(h->stream, q, wide ? n_cus : free_cus, 1, 1));
        peak_chunks = q.slots_per_utt;
    }
    // (dividing by the peak inside the final launch -- by the workgroup that finishes an utterance's last run -- was built
    //  and measured: +0.09 ms on that launch against the 0.05 ms of this kernel)
    if (peak_normalize) HIPCHK(h, launch_peak_scale(h->stream, wav, B, hop * (T - 1), msep, peak_chunks));
    return TTS_OK;
}

}  // namespace

// ======================================================================================== C ABI
extern "C" {

// Is the HIP runtime this PROCESS runs on at least the one the library was built with (major.minor)?  A host that loaded another
// ROCm's libamdhip64 first (import torch: PyTorch bundles its own) serves the library with that one -- same soname.
static bool graph_runtime_ok(int* have) {
    int v = 0;
    if (hipRuntimeGetVersion(&v) != hipSuccess) v = 0;
    if (have) *have = v;
    return v / 100000 >= HIP_VERSION / 100000;   // HIP_VERSION = major * 10^7 + minor * 10^5 + patch
}

const char* tts_version(void) { return "sstts_hip 0.1.0 (gfx950)"; }

int tts_default_config(tts_config_t* c) {
    if (!c) return TTS_ERR_INVALID;
    std::memset(c, 0, sizeof(*c));
    c->struct_size = (int32_t)sizeof(tts_config_t);
    c->vocabulary_size = 39;
    c->embedding_size = 256;
    c->enc_prenet_units[0] = 256; c->enc_prenet_units[1] = 128;
    c->enc_n_banks = 16; c->enc_n_filters = 128;
    c->enc_proj_filters[0] = 128; c->enc_proj_filters[1] = 128;
    c->post_n_banks = 8; c->post_n_filters = 128;
    c->post_proj_filters[0] = 256; c->post_proj_filters[1] = 80;
    c->n_highway_layers = 4; c->n_highway_units = 128; c->n_gru_units = 128;
    c->dec_prenet_units[0] = 256; c->dec_prenet_units[1] = 128;
    c->n_attention_units = 256; c->n_decoder_gru_units = 256; c->n_decoder_gru_layers = 2;
    c->n_mels = 80; c->reduction = 5; c->n_fft = 2048; c->force_cudnn = 0;
    c->attention_mechanism = TTS_ATTENTION_LUONG;
    c->luong_local_window_d = 10;
    c->luong_force_gaussian = 1;
    c->luong_local_mode = TTS_LOCAL_MONOTONIC;
    c->apply_post_processing = 1;
    return TTS_OK;
}

int tts_create(const tts_config_t* cfg, int device_id, tts_handle_t* out) {
    if (!cfg || !out) return fail(nullptr, TTS_ERR_INVALID, "tts_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(tts_config_t))   // (the first field: read before anything behind it is trusted)
        return fail(nullptr, TTS_ERR_INVALID,
                    "tts_create: tts_config_t.struct_size is " + std::to_string(cfg->struct_size) + ", this library's struct has " +
                    std::to_string(sizeof(tts_config_t)) + " bytes: fill the struct with tts_default_config of the header the "
                    "library was built from");
    const tts_config_t& c = *cfg;
    // constraints of the kernels
    auto mult = [](int v, int m) { return v > 0 && v % m == 0; };
    if (c.n_gru_units != 128 || c.n_highway_units != 128)
        return fail(nullptr, TTS_ERR_UNSUPPORTED, "n_gru_units and n_highway_units must be 128");
    if (c.n_attention_units != 256 || c.n_decoder_gru_units != 256 || 2 * c.n_gru_units != 256)
        return fail(nullptr, TTS_ERR_UNSUPPORTED, "attention/decoder units must be 256");
    if (c.n_decoder_gru_layers < 1 || c.n_decoder_gru_layers > 4)
        return fail(nullptr, TTS_ERR_UNSUPPORTED, "1..4 decoder GRU layers supported");
    if (!mult(c.n_mels, 16) || !mult(c.embedding_size, 16) || !mult(c.enc_prenet_units[0], 16) ||
        !mult(c.enc_prenet_units[1], 16) || !mult(c.dec_prenet_units[0], 16) || !mult(c.dec_prenet_units[1], 16) ||
        !mult(c.enc_n_filters, 32) || !mult(c.post_n_filters, 32) || !mult(c.enc_proj_filters[0], 4) ||
        !mult(c.post_proj_filters[0], 4))
        return fail(nullptr, TTS_ERR_UNSUPPORTED, "layer widths must be multiples of 16 (filters: 32)");
    if (c.enc_proj_filters[1] != c.enc_prenet_units[1] || c.post_proj_filters[1] != c.n_mels)
        return fail(nullptr, TTS_ERR_INVALID, "last projection must match the CBHG input width (residual)");
    if (c.attention_mechanism != TTS_ATTENTION_LUONG && c.attention_mechanism != TTS_ATTENTION_LOCAL_LUONG)
        return fail(nullptr, TTS_ERR_UNSUPPORTED, "attention_mechanism must be TTS_ATTENTION_LUONG or TTS_ATTENTION_LOCAL_LUONG");
    if (c.attention_mechanism == TTS_ATTENTION_LOCAL_LUONG && c.luong_local_window_d < 1)
        return fail(nullptr, TTS_ERR_INVALID, "luong_local_window_d must be >= 1");
    if (c.luong_local_mode != TTS_LOCAL_MONOTONIC && c.luong_local_mode != TTS_LOCAL_PREDICTIVE)
        return fail(nullptr, TTS_ERR_INVALID, "luong_local_mode must be TTS_LOCAL_MONOTONIC or TTS_LOCAL_PREDICTIVE");
    if (c.enc_n_banks < 1 || c.post_n_banks < 1 || c.reduction < 1 || c.vocabulary_size < 1 || c.n_highway_layers < 0)
        return fail(nullptr, TTS_ERR_INVALID, "bad counts");
    if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, TTS_ERR_HIP, "hipSetDevice failed");
    auto h = new tts_handle_s();
    h->cfg = c;
    h->device = device_id;
    if (hipStreamCreate(&h->stream) != hipSuccess) {
        delete h;
        return fail(nullptr, TTS_ERR_HIP, "hipStreamCreate failed");
    }
    h->own_stream = true;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || cus < 1) cus = 256;
        h->n_cus_dev = cus;
    }
    if (h->use_graph && !graph_runtime_ok(nullptr)) h->use_graph = 0;
    build_manifest(h);
    *out = h;
    return TTS_OK;
}

int tts_destroy(tts_handle_t h) {
    DeviceScope dev_scope(h);
    if (!h) return TTS_OK;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    for (auto& s : h->spans) {
        hipEventDestroy(s.a);
        hipEventDestroy(s.b);
    }
    graph_drop(h);
    for (auto& kv : h->ws)
        if (kv.second.p) hipFree(kv.second.p);
    if (h->arena) hipFree(h->arena);
    if (h->gl.window) hipFree(h->gl.window);
    if (h->gl.wss) hipFree(h->gl.wss);
    if (h->gl.wlane) hipFree(h->gl.wlane);
    if (h->gl.tw1024) hipFree(h->gl.tw1024);
    if (h->gl.tw2048) hipFree(h->gl.tw2048);
    if (h->gl.tables) hipFree(h->gl.tables);
    if (h->an.window) hipFree(h->an.window);
    if (h->an.mel_wt) hipFree(h->an.mel_wt);
    if (h->an.flag) hipFree(h->an.flag);
    if (h->front) {
        hipStreamSynchronize(h->front);
        hipStreamDestroy(h->front);
    }
    if (h->aux) {
        hipStreamSynchronize(h->aux);
        hipStreamDestroy(h->aux);
    }
    if (h->hold_flags) hipFree(h->hold_flags);
    for (int i = 0; i < 3; ++i) {
        if (h->hio.ids_pinned[i]) hipHostFree(h->hio.ids_pinned[i]);
        if (h->hio.ids_dev[i]) hipFree(h->hio.ids_dev[i]);
        if (h->hio.wav_pinned[i]) hipHostFree(h->hio.wav_pinned[i]);
        if (h->hio.wav_dev[i]) hipFree(h->hio.wav_dev[i]);
        if (i == 0) {
            for (auto& kv : h->glg.tw) hipFree(kv.second);
            if (h->glg.window) hipFree(h->glg.window);
            if (h->glg.rwss) hipFree(h->glg.rwss);
        }
        if (h->hio.lin_pinned[i]) hipHostFree(h->hio.lin_pinned[i]);
        if (h->hio.lin_dev[i]) hipFree(h->hio.lin_dev[i]);
        if (h->hio.ali_pinned[i]) hipHostFree(h->hio.ali_pinned[i]);
        if (h->hio.ali_dev[i]) hipFree(h->hio.ali_dev[i]);
        if (h->hio.ev_h2d[i]) hipEventDestroy(h->hio.ev_h2d[i]);
        if (h->hio.ev_enc[i]) hipEventDestroy(h->hio.ev_enc[i]);
        if (h->hio.ev_ready[i]) hipEventDestroy(h->hio.ev_ready[i]);
        if (h->hio.ev_d2h[i]) hipEventDestroy(h->hio.ev_d2h[i]);
    }
    if (h->hio.status_pinned) hipHostFree(h->hio.status_pinned);
    if (h->hio.in) hipStreamDestroy(h->hio.in);
    if (h->hio.out) hipStreamDestroy(h->hio.out);
    if (h->ev_aux) hipEventDestroy(h->ev_aux);
    if (h->ev_front_done) hipEventDestroy(h->ev_front_done);
    for (int i = 0; i < 2; ++i) {
        if (h->ev_enc_ready[i]) hipEventDestroy(h->ev_enc_ready[i]);
        if (h->ev_dec_done[i]) hipEventDestroy(h->ev_dec_done[i]);
        if (h->ev_gap[i]) hipEventDestroy(h->ev_gap[i]);
    }
    if (h->encs) hipStreamDestroy(h->encs);
    if (h->ev_serial_done) hipEventDestroy(h->ev_serial_done);
    if (h->ev_graph_done) hipEventDestroy(h->ev_graph_done);
    gemm_drop_images(h);
    for (int i = 0; i < 2; ++i) {
        if (h->ev_post_done[i]) hipEventDestroy(h->ev_post_done[i]);
        if (h->ev_gl_done[i]) hipEventDestroy(h->ev_gl_done[i]);
    }
    if (h->own_stream) hipStreamDestroy(h->stream);
    delete h;
    return TTS_OK;
}

const char* tts_last_error(tts_handle_t h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int tts_set_stream(tts_handle_t h, void* s) {
    DeviceScope dev_scope(h);
    if (!h) return TTS_ERR_INVALID;
    {
        int rc = sync_all(h);
        if (rc) return rc;
    }
    h->post_pending[0] = h->post_pending[1] = false;
    h->gl_pending[0] = h->gl_pending[1] = false;
    h->gl_wide_used[0] = h->gl_wide_used[1] = false;
    {
        int rc = graph_drop(h);
        if (rc) return rc;
    }
    if (h->own_stream) hipStreamDestroy(h->stream);
    if (s) {
        h->stream = reinterpret_cast<hipStream_t>(s);
        h->own_stream = false;
    } else {
        HIPCHK(h, hipStreamCreate(&h->stream));
        h->own_stream = true;
    }
    return TTS_OK;
}

int tts_set_option(tts_handle_t h, const char* key, int value) {
    DeviceScope dev_scope(h);
    if (!h || !key) return TTS_ERR_INVALID;
    if (!std::strcmp(key, "use_graph")) {
        int have = 0;
        // (value 2 behind "debug_hooks": tools/graph_probe.py reproduces the problem on the other runtime with it)
        if (value && !(value == 2 && h->debug_hooks) && !graph_runtime_ok(&have))
            return fail(h, TTS_ERR_UNSUPPORTED,
                        "use_graph: this process runs the library on HIP runtime " + std::to_string(have) + ", older than the " +
                        std::to_string(HIP_VERSION) + " it was built with (a libamdhip64 loaded before the library, e.g. the one "
                        "PyTorch bundles); hipGraph replays of the decoder are wrong there (csrc/api.hip, `use_graph`) -- the "
                        "launches are enqueued directly instead");
        h->use_graph = value;
    }
    else if (!std::strcmp(key, "profile")) h->profile = value;
    else if (!std::strcmp(key, "fused_tail")) h->fused_tail = value;
    else if (!std::strcmp(key, "persistent_decoder")) {
        // the decoder form decides whether a pipelined call runs its encoder ahead on `encs` (tts_synthesize: enc_ahead_cfg):
        // a call of the other form may still be using the one set of encoder workspaces and the `memory` buffer of its
        // parity on `front`, which the encoder-ahead ordering (ev_dec_done of the call two back) does not cover
        if (value != h->persistent_decoder) {
            int rc = sync_all(h);
            if (rc) return rc;
        }
        h->persistent_decoder = value;
    }
    else if (!std::strcmp(key, "gl_pair")) h->gl_pair = value;
    else if (!std::strcmp(key, "gl_wide_from")) h->gl_wide = value < -2 ? -2 : value;
    else if (!std::strcmp(key, "deterministic")) h->deterministic = value ? 1 : 0;
    else if (!std::strcmp(key, "gemm_presplit") || !std::strcmp(key, "gemm_ps")) {
        if (value && !gemm_experiments_built())
            return fail(h, TTS_ERR_UNSUPPORTED, std::string(key) + ": a measured-and-not-faster GEMM variant of round 5; its kernels are only "
                        "in a tools build of gemm_f32.hip (-DGEMM_EXPERIMENTS, tools/build_variant.sh)");
        (key[5] == 'p' && key[6] == 'r' ? h->gemm_presplit : h->gemm_ps) = value;
    }
    else if (!std::strcmp(key, "pd_ws")) {
        if (value != h->pd_ws) {   // (may change whether a pipelined call's decoder is a persistent kernel at all)
            int rc = sync_all(h);
            if (rc) return rc;
        }
        h->pd_ws = value;
    }
    else if (!std::strcmp(key, "enc_stream")) {
        int rc = sync_all(h);
        if (rc) return rc;
        h->enc_stream = value;
    }
    else if (!std::strcmp(key, "debug_hooks")) h->debug_hooks = value;
    else if (!std::strcmp(key, "pd_debug_delay") || !std::strcmp(key, "gl_runs") || !std::strcmp(key, "gl_run_len") ||
             !std::strcmp(key, "timeline") || !std::strcmp(key, "gl_workers") || !std::strcmp(key, "pd_rows")) {
        if (!h->debug_hooks && value != 0)
            return fail(h, TTS_ERR_INVALID, std::string(key) + ": a test hook; set the option \"debug_hooks\" to 1 on this handle first");
        if (!std::strcmp(key, "pd_debug_delay")) h->pd_debug_delay = value;
        else if (!std::strcmp(key, "gl_runs")) h->gl_runs = value;
        else if (!std::strcmp(key, "gl_run_len")) h->gl_run_len = value;
        else if (!std::strcmp(key, "gl_workers")) h->gl_workers = value;
        else if (!std::strcmp(key, "pd_rows")) h->pd_rows = value;
        else h->timeline = value;
    }
    else if (!std::strcmp(key, "reserve_cus")) {
        if (value != h->reserve_cus) {   // decides the encoder-ahead form as well (see "persistent_decoder")
            int rc = sync_all(h);
            if (rc) return rc;
        }
        h->reserve_cus = value;
    } else if (!std::strcmp(key, "hold_lds_kb")) {
        if (value < 1 || value > 160) return fail(h, TTS_ERR_INVALID, "hold_lds_kb must be 1..160");
        h->hold_lds_kb = value;
    } else if (!std::strcmp(key, "pipeline")) {
        int rc = sync_all(h);
        if (rc) return rc;
        h->post_pending[0] = h->post_pending[1] = false;
        h->gl_pending[0] = h->gl_pending[1] = false;
        h->gl_wide_used[0] = h->gl_wide_used[1] = false;
        h->pipeline = value;
    }
    else return fail(h, TTS_ERR_INVALID, std::string("unknown option ") + key);
    return TTS_OK;
}

int tts_synchronize(tts_handle_t h) {
    DeviceScope dev_scope(h);
    if (!h) return TTS_ERR_INVALID;
    return sync_all(h);
}

int tts_manifest_size(tts_handle_t h) { return h ? (int)h->manifest.size() : TTS_ERR_INVALID; }

int tts_manifest_entry(tts_handle_t h, int i, const char** name, int64_t shape[4], int* ndim) {
    if (!h || i < 0 || i >= (int)h->manifest.size()) return TTS_ERR_INVALID;
    const auto& e = h->manifest[i];
    if (name) *name = e.name.c_str();
    if (ndim) *ndim = (int)e.shape.size();
    if (shape)
        for (size_t d = 0; d < 4; ++d) shape[d] = d < e.shape.size() ? e.shape[d] : 1;
    return TTS_OK;
}

int tts_set_weight(tts_handle_t h, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!h || !name || !data || !shape) return TTS_ERR_INVALID;
    for (const auto& e : h->manifest) {
        if (e.name != name) continue;
        if ((int)e.shape.size() != ndim) return fail(h, TTS_ERR_INVALID, std::string("rank mismatch for ") + name);
        for (int d = 0; d < ndim; ++d)
            if (e.shape[d] != shape[d]) return fail(h, TTS_ERR_INVALID, std::string("shape mismatch for ") + name);
        h->host_w[e.name].assign(data, data + e.numel());
        h->finalized = false;
        return TTS_OK;
    }
    return fail(h, TTS_ERR_INVALID, std::string("unknown weight ") + name);
}

int tts_load_weights_blob(tts_handle_t h, const float* blob, size_t n) {
    if (!h || !blob) return TTS_ERR_INVALID;
    size_t total = 0;
    for (const auto& e : h->manifest) total += e.numel();
    if (total != n)
        return fail(h, TTS_ERR_INVALID,
                    "blob has " + std::to_string(n) + " floats, manifest needs " + std::to_string(total));
    size_t off = 0;
    for (const auto& e : h->manifest) {
        h->host_w[e.name].assign(blob + off, blob + off + e.numel());
        off += e.numel();
    }
    h->finalized = false;
    return TTS_OK;
}

int tts_finalize_weights(tts_handle_t h) {
    DeviceScope dev_scope(h);
    if (!h) return TTS_ERR_INVALID;
    for (const auto& e : h->manifest)
        if (!h->host_w.count(e.name)) return fail(h, TTS_ERR_NOT_LOADED, "missing weight " + e.name);
    HIPCHK(h, hipSetDevice(h->device));
    const tts_config_t& c = h->cfg;
    const bool cudnn = c.force_cudnn != 0;
    Packer p;
    const size_t o_emb = pack_copy(p, W(h, "encoder/embedding").data(), (size_t)c.vocabulary_size * c.embedding_size);
    size_t o_epw[2], o_epb[2];
    int n_in = c.embedding_size;
    for (int i = 0; i < 2; ++i) {
        const std::string s = "encoder/pre_net/" + std::to_string(i + 1) + "-FC-" + std::to_string(c.enc_prenet_units[i]);
        o_epw[i] = pack_transposed(p, W(h, s + "/kernel").data(), n_in, c.enc_prenet_units[i]);
        o_epb[i] = pack_copy(p, W(h, s + "/bias").data(), c.enc_prenet_units[i]);
        n_in = c.enc_prenet_units[i];
    }
    const CbhgOffsets o_enc =
        pack_cbhg(h, p, "encoder", n_in, c.enc_n_banks, c.enc_n_filters, c.enc_proj_filters, cudnn);
    const int mem = 2 * c.n_gru_units, att = c.n_attention_units, U = c.n_decoder_gru_units;
    const size_t o_mem = pack_transposed(p, W(h, "decoder2/memory_layer/kernel").data(), mem, att);
    size_t o_dpw[2], o_dpb[2];
    n_in = c.n_mels + att;
    for (int i = 0; i < 2; ++i) {
        const std::string s = std::string(kAtt) + "/pre_net/" + std::to_string(i + 1) + "-FC-" +
                              std::to_string(c.dec_prenet_units[i]);
        o_dpw[i] = pack_transposed(p, W(h, s + "/kernel").data(), n_in, c.dec_prenet_units[i]);
        o_dpb[i] = pack_copy(p, W(h, s + "/bias").data(), c.dec_prenet_units[i]);
        n_in = c.dec_prenet_units[i];
    }
    // pre-net layer 1 with the output projection folded in (steps >= 1):
    //   x_t W1x = (y W_o + b_o)[-n_mels:] W1x = y (W_o[:, -n_mels:] W1x) + b_o[-n_mels:] W1x
    size_t o_dpfw, o_dpfb;
    {
        const int NM = c.n_mels, OUTW = c.n_mels * c.reduction, P1 = c.dec_prenet_units[0];
        const std::string s1 = std::string(kAtt) + "/pre_net/1-FC-" + std::to_string(P1);
        const auto& w1 = W(h, s1 + "/kernel");   // [NM + att][P1]
        const auto& b1 = W(h, s1 + "/bias");
        const auto& wo = W(h, "decoder2/decoder/output_projection_wrapper/kernel");   // [U][OUTW]
        const auto& bo = W(h, "decoder2/decoder/output_projection_wrapper/bias");
        const int K = U + att;
        o_dpfw = p.alloc((size_t)P1 * K);
        o_dpfb = p.alloc((size_t)P1);
        for (int n = 0; n < P1; ++n) {
            for (int k = 0; k < U; ++k) {
                double acc = 0.0;
                for (int j = 0; j < NM; ++j)
                    acc += (double)wo[(size_t)k * OUTW + (OUTW - NM) + j] * (double)w1[(size_t)j * P1 + n];
                p.host[o_dpfw + (size_t)n * K + k] = (float)acc;
            }
            for (int k = 0; k < att; ++k) p.host[o_dpfw + (size_t)n * K + U + k] = w1[(size_t)(NM + k) * P1 + n];
            double bacc = (double)b1[n];
            for (int j = 0; j < NM; ++j) bacc += (double)bo[(OUTW - NM) + j] * (double)w1[(size_t)j * P1 + n];
            p.host[o_dpfb + n] = (float)bacc;
        }
    }
    const GruOffsets o_ag = pack_dec_gru(h, p, std::string(kAtt) + "/gru_cell", n_in, att, cudnn);
    const size_t o_al = pack_transposed(p, W(h, std::string(kAtt) + "/attention_layer/kernel").data(), att + mem, att);
    const bool predictive = c.attention_mechanism == TTS_ATTENTION_LOCAL_LUONG && c.luong_local_mode == TTS_LOCAL_PREDICTIVE;
    size_t o_wp = 0, o_vp = 0;
    if (predictive) {
        o_wp = pack_copy(p, W(h, std::string(kAtt) + "/local_luong_attention/local_w_p").data(), (size_t)att * att);
        o_vp = pack_copy(p, W(h, std::string(kAtt) + "/local_luong_attention/local_v_p").data(), (size_t)att);
    }
    GruOffsets o_dg[4];
    for (int i = 0; i < c.n_decoder_gru_layers; ++i)
        o_dg[i] = pack_dec_gru(h, p, std::string(kMrc) + "/cell_" + std::to_string(i + 1) + "/gru_cell",
                               i == 0 ? att : U, U, cudnn);
    const int OUT = c.n_mels * c.reduction;
    const size_t o_ow = pack_transposed(p, W(h, "decoder2/decoder/output_projection_wrapper/kernel").data(), U, OUT);
    const size_t o_ob = pack_copy(p, W(h, "decoder2/decoder/output_projection_wrapper/bias").data(), OUT);
    CbhgOffsets o_post{};
    if (c.apply_post_processing)
        o_post = pack_cbhg(h, p, "post_process", c.n_mels, c.post_n_banks, c.post_n_filters, c.post_proj_filters, cudnn);
    const int F = 1 + c.n_fft / 2;
    const size_t o_dw = pack_transposed(p, W(h, "dense/kernel").data(), c.apply_post_processing ? mem : c.n_mels, F);
    const size_t o_db = pack_copy(p, W(h, "dense/bias").data(), F);
    const size_t o_zero = p.alloc(1024);
    // the decoder's weights once more, in the register order of the weight-stationary persistent kernel (decoder_ws.hip):
    // TF GRUCell form, the default layer sizes (decoder_ws_supports checks the rest per call)
    size_t o_wsw = 0, o_wsb = 0;
    const bool ws_image = c.n_decoder_gru_layers == 2 && att == 256 && U == 256 && mem == 256 &&
                          c.dec_prenet_units[0] == 256 && c.dec_prenet_units[1] == 128 && c.n_mels <= 256;
    if (ws_image) {
        o_wsw = p.alloc(decoder_ws_wimg_floats());
        o_wsb = p.alloc(decoder_ws_bimg_floats());
        const float* hb = p.host.data();   // (no allocation below this line)
        DecWsHostWeights hw;
        hw.w1f = hb + o_dpfw; hw.b1f = hb + o_dpfb; hw.b1 = hb + o_dpb[0]; hw.w2 = hb + o_dpw[1]; hw.b2 = hb + o_dpb[1];
        hw.ag_w = hb + o_ag.gates_wt; hw.ag_b = hb + o_ag.gates_b; hw.ac_w = hb + o_ag.cand_wt; hw.ac_b = hb + o_ag.cand_b;
        hw.al_w = hb + o_al;
        for (int l = 0; l < 2; ++l) {
            hw.g_gw[l] = hb + o_dg[l].gates_wt; hw.g_gb[l] = hb + o_dg[l].gates_b;
            hw.g_cw[l] = hb + o_dg[l].cand_wt; hw.g_cb[l] = hb + o_dg[l].cand_b;
        }
        hw.cudnn = cudnn ? 1 : 0;
        decoder_ws_pack(hw, p.host.data() + o_wsw, p.host.data() + o_wsb);
    }

    {
        int rc = sync_all(h);
        if (rc) return rc;
    }
    {
        int rc = graph_drop(h);
        if (rc) return rc;
    }
    gemm_drop_images(h);   // (keyed by addresses inside the old arena)
    if (h->arena) hipFree(h->arena);
    h->arena = nullptr;
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->arena), p.host.size() * sizeof(float)));
    h->arena_floats = p.host.size();
    HIPCHK(h, hipMemcpy(h->arena, p.host.data(), p.host.size() * sizeof(float), hipMemcpyHostToDevice));
    const float* base = h->arena;
    h->embedding = base + o_emb;
    for (int i = 0; i < 2; ++i) {
        h->enc_pre_wt[i] = base + o_epw[i];
        h->enc_pre_b[i] = base + o_epb[i];
    }
    h->enc = CbhgWeights();
    h->post = CbhgWeights();
    bind_cbhg(h->enc, o_enc, base, c.enc_prenet_units[1], c.enc_n_banks, c.enc_n_filters, c.enc_proj_filters);
    if (c.apply_post_processing) bind_cbhg(h->post, o_post, base, c.n_mels, c.post_n_banks, c.post_n_filters, c.post_proj_filters);
    h->mem_wt = base + o_mem;
    DecoderWeights& d = h->dec;
    std::memset(&d, 0, sizeof(d));
    d.prenet1_wt = base + o_dpw[0]; d.prenet1_b = base + o_dpb[0];
    d.prenet1f_wt = base + o_dpfw; d.prenet1f_b = base + o_dpfb;
    d.prenet2_wt = base + o_dpw[1]; d.prenet2_b = base + o_dpb[1];
    d.att_gru = {base + o_ag.gates_wt, base + o_ag.gates_b, base + o_ag.cand_wt, base + o_ag.cand_b};
    d.attn_layer_wt = base + o_al;
    for (int i = 0; i < c.n_decoder_gru_layers; ++i)
        d.gru[i] = {base + o_dg[i].gates_wt, base + o_dg[i].gates_b, base + o_dg[i].cand_wt, base + o_dg[i].cand_b};
    d.out_wt = base + o_ow; d.out_b = base + o_ob;
    d.n_layers = c.n_decoder_gru_layers; d.att_units = att; d.dec_units = U; d.mem_units = mem;
    d.local_d = c.attention_mechanism == TTS_ATTENTION_LOCAL_LUONG ? c.luong_local_window_d : 0;
    d.local_gaussian = c.luong_force_gaussian != 0;
    d.local_predictive = predictive ? 1 : 0;
    d.local_wp = predictive ? base + o_wp : nullptr;
    d.local_vp = predictive ? base + o_vp : nullptr;
    d.n_mels = c.n_mels; d.reduction = c.reduction;
    d.prenet1_units = c.dec_prenet_units[0]; d.prenet2_units = c.dec_prenet_units[1];
    d.ws_wimg = ws_image ? base + o_wsw : nullptr;
    d.ws_bimg = ws_image ? base + o_wsb : nullptr;
    h->dense_wt = base + o_dw;
    h->dense_b = base + o_db;
    h->zeros = base + o_zero;
    std::memset(&h->dec_key, 0, sizeof(h->dec_key));
    h->host_w.clear();   // the packed copy on device is the only one kept
    h->finalized = true;
    return TTS_OK;
}

int tts_malloc(void** dptr, size_t bytes) {
    if (!dptr) return TTS_ERR_INVALID;
    return hipMalloc(dptr, bytes ? bytes : 4) == hipSuccess ? TTS_OK : TTS_ERR_HIP;
}
int tts_free(void* dptr) { return hipFree(dptr) == hipSuccess ? TTS_OK : TTS_ERR_HIP; }
int tts_device_malloc(tts_handle_t h, void** dptr, size_t bytes) {
    if (!h || !dptr) return TTS_ERR_INVALID;
    DeviceScope dev_scope(h);
    HIPCHK(h, hipMalloc(dptr, bytes ? bytes : 4));
    return TTS_OK;
}
int tts_device_free(tts_handle_t h, void* dptr) {
    if (!h) return TTS_ERR_INVALID;
    DeviceScope dev_scope(h);
    HIPCHK(h, hipFree(dptr));
    return TTS_OK;
}
int tts_memcpy_h2d(tts_handle_t h, void* dst, const void* src, size_t bytes) {
    DeviceScope dev_scope(h);
    if (!h) return TTS_ERR_INVALID;
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return TTS_OK;
}
int tts_memcpy_d2h(tts_handle_t h, void* dst, const void* src, size_t bytes) {
    DeviceScope dev_scope(h);
    if (!h) return TTS_ERR_INVALID;
    if (h->front) HIPCHK(h, hipStreamSynchronize(h->front));   // optional outputs of a pipelined synthesize
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // the bytes are host-visible from here on: a timed-out persistent kernel must not pass for a result
    return check_status(h);
}
int tts_memset(tts_handle_t h, void* dst, int value, size_t bytes) {
    DeviceScope dev_scope(h);
    if (!h) return TTS_ERR_INVALID;
    HIPCHK(h, hipMemsetAsync(dst, value, bytes, h->stream));
    return TTS_OK;
}

// ---------------------------------------------------------------------------------------- stages
// A stage entry point called by the USER (not by tts_synthesize) runs on the main stream in the one set of enc.* / dec.*
// workspaces that the pipelined calls use on the encoder and front streams: it starts behind whatever those streams still
// hold, and the next pipelined call's encoder and decoder start behind it (ev_serial_done, as for an unpipelined
// tts_synthesize).  Stream order alone covers the post-net (main stream on both sides).
static int standalone_begin(tts_handle_t h) {
    if (h->in_synthesize || !h->encs) return TTS_OK;
    for (int i = 0; i < 2; ++i) {
        if (h->enc_ready_pending[i]) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_enc_ready[i], 0));
        if (h->dec_done_pending[i]) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_dec_done[i], 0));
    }
    return TTS_OK;
}
static int standalone_end(tts_handle_t h) {
    if (h->in_synthesize || !h->front) return TTS_OK;
    if (!h->ev_serial_done) HIPCHK(h, hipEventCreateWithFlags(&h->ev_serial_done, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(h->ev_serial_done, h->stream));
    h->serial_pending = true;
    return TTS_OK;
}
struct SynthScope {   // tts_synthesize is running: the stage entry points leave the stream ordering to it
    tts_handle_t h;
    explicit SynthScope(tts_handle_t h_) : h(h_) { h->in_synthesize = true; }
    ~SynthScope() { h->in_synthesize = false; }
};

static int encoder_impl(tts_handle_t h, const int32_t* ids, int B, int Ts, float* memory);
int tts_encoder_forward(tts_handle_t h, const int32_t* ids, int B, int Ts, float* memory) {
    DeviceScope dev_scope(h);
    int rc = check_ready(h);
    if (rc) return rc;
    if (!ids || !memory || B < 1 || Ts < 1) return fail(h, TTS_ERR_INVALID, "encoder_forward: bad arguments");
    if ((rc = standalone_begin(h))) return rc;
    if ((rc = encoder_impl(h, ids, B, Ts, memory))) return rc;
    return standalone_end(h);
}
static int encoder_impl(tts_handle_t h, const int32_t* ids, int B, int Ts, float* memory) {
    int rc = TTS_OK;
    const tts_config_t& c = h->cfg;
    const int M = B * Ts;
    WS(h, "enc.pre1", float, (size_t)M * c.enc_prenet_units[0], pre1);
    WS(h, "enc.pre2", float, (size_t)M * c.enc_prenet_units[1], pre2);
    int64_t launches = 0;
    ProfScope ps(h, ST_ENCODER, 0);
    {   // embedding lookup fused into the first pre-net GEMM (row gather)
        GemmGroup g = dense_group(h->embedding, c.embedding_size, h->enc_pre_wt[0], h->enc_pre_b[0], pre1,
                                  c.enc_prenet_units[0], M, c.enc_prenet_units[0], c.embedding_size, ACT_RELU);
        g.gather = ids;
        g.gather_rows = c.vocabulary_size;
        if ((rc = run_single(h, g))) return rc;
    }
    if ((rc = run_single(h, dense_group(pre1, c.enc_prenet_units[0], h->enc_pre_wt[1], h->enc_pre_b[1], pre2,
                                        c.enc_prenet_units[1], M, c.enc_prenet_units[1], c.enc_prenet_units[0],
                                        ACT_RELU))))
        return rc;
    launches += 2;
    if ((rc = run_cbhg(h, h->enc, "enc", pre2, B, Ts, memory, &launches))) return rc;
    if (ps.idx >= 0) h->spans[ps.idx].launches = launches;
    return TTS_OK;
}

// Which persistent decoder kernel a call of this shape takes when `budget` compute units are free for it:
// 2 = weight-stationary (decoder_ws.hip), 1 = decoder_persistent.hip, 0 = neither (launch-per-layer path).
static int pd_kernel_for(tts_handle_t h, int B, int Ts, int budget) {
    const int cudnn = h->cfg.force_cudnn;
    if (h->pd_ws && decoder_ws_supports(h->dec, cudnn, B, Ts) && decoder_ws_workgroups(B) <= budget) return 2;
    if (decoder_persistent_supports(h->dec, cudnn, B, Ts) && decoder_persistent_workgroups(B) <= budget) return 1;
    return 0;
}

// ... and which one the option "persistent_decoder" picks for a call: 0 never; 2 whenever a kernel covers the configuration;
// 1 (default) by what was measured (tools/pipeline_sweep.py, tools/latency_bench.py): the weight-stationary kernel wherever it
// covers the configuration and its workgroups fit -- under the call pipeline at every batch size (round 5: 8.5 against 10.1 ms
// per call at B = 1, 9.6 against 13.7 at 32, 12.4 against 16.4 at 48: the launch-per-layer decoder's ~2000 launches queue behind
// Griffin-Lim) and, since round 6, for unpipelined calls as well: with 16 utterances per cluster (decoder_impl picks the rows)
// the loop takes 6.15-6.2 ms at B = 1 ... 64 on an idle chip against 7.5 ... 9.4 ms launch per layer and 8.25 with 32 rows
// (profiles/r06_stage_benchmarks.txt).  Its bits do not depend on the rows per cluster, on the batch size or on whether the
// call was pipelined (tests/test_gpu_persistent.py, test_gpu_full_size.py::test_shard_invariance), so a call's spectrograms no
// longer depend on the call history of the handle.  decoder_persistent.hip (streamed weights: LocalLuongAttention, or "pd_ws"
// = 0) only under the pipeline with more than 48 utterances, where the step is bound by post-net + Griffin-Lim (rounds 2-4).
static int pd_choice(tts_handle_t h, int B, int Ts, int budget, bool pipelined) {
    if (h->persistent_decoder <= 0) return 0;
    const int k = pd_kernel_for(h, B, Ts, budget);
    if (h->persistent_decoder >= 2) return k;
    if (k == 2) return 2;
    if (k == 1) return (pipelined && B > 48) ? 1 : 0;
    return 0;
}

// The launch index from which a pipelined call's Griffin-Lim launches are cut for ALL compute units (gl_run, wide_from), or -1.
// Under the pipeline the next call's decoder starts with this call's first Griffin-Lim launch (its encoder ran in the gap
// before it) on the `reserve_cus` units that Griffin-Lim leaves free, and with the weight-stationary kernel it is done long
// before the last launch (8.9 ms of 12.0 at 64 utterances x 1000 frames): the launches after that would leave 32 units idle.
// Nothing orders the two streams here -- a 256-workgroup launch that finds units still taken runs as 224 workers and its
// last 32 items wait, which costs time (a launch of 0.5 ms becomes 1.0) and never bits -- so the index comes from a model of
// the two durations with half a launch of margin (measured at 64 x 1000 x 60 iterations, GRUCell form, profiles/r05_experiment_gl_wide.txt:
// never 14.74 ms per step, from launch 14: 14.79, 15: 14.49, 16: 14.48, 17: 14.52, 18: 14.56): decoder 0.045 ms per step (both GRU forms; measured 8.9 ms / 200
// steps beside Griffin-Lim), Griffin-Lim 3.1 ns per frame-iteration on the reduced unit count (0.60 ms per launch of
// 3 x 64 x 1000).  A function of the call's shape and the handle's options alone: the waveform's bits do not depend on timing.
static int gl_wide_from(tts_handle_t h, int B, int Ts, int n_steps, int T, int n_iter) {
    if (h->gl_wide == -2 || h->reserve_cus <= 0) return -1;
    const int pd = pd_choice(h, B, Ts, h->reserve_cus, true);
    if (pd == 0) return -1;                    // launch-per-layer decoder: sleeper workgroups hold the units through the whole phase
    if (h->gl_wide >= 0) return h->gl_wide;    // (tools: an explicit launch index)
    if (pd != 2) return -1;                    // the streamed-weights decoder outlasts Griffin-Lim
    const int per_launch = h->gl_pair < 1 ? 1 : (h->gl_pair > 3 ? 3 : h->gl_pair);
    const double launch_ms = 3.125e-6 * (double)B * T * per_launch;
    const double dec_ms = (h->cfg.force_cudnn ? 0.038 : 0.045) * n_steps + 0.1;   // (seven hand-offs per step instead of ten: 6.9 ms alone)
    const int n_launches = (n_iter + per_launch - 1) / per_launch;
    const int from = (int)std::ceil((dec_ms + 0.5 * launch_ms + 0.1) / launch_ms);
    return from <= n_launches ? from : -1;
}

// keys = memory_layer(memory), no bias (LuongAttention, reference tacotron/model.py:205-223; the values stay the raw memory)
static int attention_keys(tts_handle_t h, const float* memory, int B, int Ts, float* keys) {
    const int A = h->cfg.n_attention_units, mem = 2 * h->cfg.n_gru_units;
    return run_single(h, dense_group(memory, mem, h->mem_wt, nullptr, keys, A, B * Ts, A, mem, ACT_NONE));
}

static int decoder_impl(tts_handle_t h, const float* memory, int B, int Ts, int n_steps, float* mel, float* alignments);
int tts_decoder_forward(tts_handle_t h, const float* memory, int B, int Ts, int n_steps, float* mel,
                        float* alignments) {
    DeviceScope dev_scope(h);
    int rc = check_ready(h);
    if (rc) return rc;
    if (!memory || !mel || B < 1 || Ts < 1 || n_steps < 1) return fail(h, TTS_ERR_INVALID, "decoder_forward: bad arguments");
    if ((rc = standalone_begin(h))) return rc;
    if ((rc = decoder_impl(h, memory, B, Ts, n_steps, mel, alignments))) return rc;
    return standalone_end(h);
}
static int decoder_impl(tts_handle_t h, const float* memory, int B, int Ts, int n_steps, float* mel, float* alignments) {
    int rc = TTS_OK;
    const tts_config_t& c = h->cfg;
    if (h->dec.local_d > 0 && Ts < 2 * h->dec.local_d + 1)
        return fail(h, TTS_ERR_UNSUPPORTED,
                    "LocalLuongAttention: the memory must hold at least 2*D+1 positions (for shorter inputs the reference pads "
                    "the window's 2D+1 alignments to 4D+2-T_s entries, tacotron/attention.py:294-299,85-92: its attention state "
                    "changes shape and TensorFlow fails)");
    const int A = c.n_attention_units, U = c.n_decoder_gru_units, mem = 2 * c.n_gru_units;
    const int NL = c.n_decoder_gru_layers;
    WS(h, "dec.keys", float, (size_t)B * Ts * A, keys_ws);
    // (the call pipeline computes the keys behind the encoder, on the encoder's stream, in a buffer of the call's parity)
    float* keys = h->pre_keys ? h->pre_keys : keys_ws;
    const bool have_keys = h->pre_keys != nullptr;
    h->pre_keys = nullptr;
    const size_t state_floats = (size_t)B * (A + 2 * ((size_t)A + (size_t)NL * U));   // att | h_att, h_dec[] | their second copies
    WS(h, "dec.state", float, state_floats, state);
    WS(h, "dec.tmp", float, (size_t)B * (c.dec_prenet_units[0] + c.dec_prenet_units[1] + 6 * (size_t)U), tmp);
    WS(h, "dec.ctx_parts", float, (size_t)TTS_ATT_PARTS * B * mem, ctx_parts);
    WS(h, "dec.att_stats", float, (size_t)n_steps * B * TTS_ATT_PARTS * 2, att_stats);
    // (the launch-per-layer path replays a captured graph with its buffers baked in: one y history there)
    const int pd_budget = h->cur_cu_budget > 0 ? h->cur_cu_budget : h->n_cus_dev;
    const int pd_kernel = pd_choice(h, B, Ts, pd_budget, h->cur_cu_budget > 0);
    const bool use_pd = pd_kernel != 0;
    const bool defer_proj = h->defer_projection && use_pd;
    WS(h, defer_proj ? (h->defer_parity ? "dec.yhist.odd" : "dec.yhist.even") : "dec.yhist", float, (size_t)B * n_steps * U, yhist);
    WS(h, "dec.align_raw", float, (size_t)n_steps * B * Ts, align_raw);
    DecoderScratch sc;
    std::memset(&sc, 0, sizeof(sc));
    sc.state = state;
    sc.state_bytes = state_floats * sizeof(float);
    sc.att = state;
    sc.h_att = state + (size_t)B * A;
    for (int l = 0; l < NL; ++l) sc.h_dec[l] = state + (size_t)B * (2 * A + (size_t)l * U);
    {
        float* alt = state + (size_t)B * (2 * A + (size_t)NL * U);
        sc.h_att_alt = alt;
        for (int l = 0; l < NL; ++l) sc.h_dec_alt[l] = alt + (size_t)B * (A + (size_t)l * U);
    }
    float* t = tmp;
    sc.p1 = t; t += (size_t)B * c.dec_prenet_units[0];
    sc.p2 = t; t += (size_t)B * c.dec_prenet_units[1];
    sc.rh = t; t += (size_t)B * U;
    sc.u = t; t += (size_t)B * U;
    sc.hh = t; t += (size_t)B * U;
    sc.xi = t; t += (size_t)B * U;
    sc.y0 = t; t += (size_t)B * U;
    sc.y1 = t; t += (size_t)B * U;
    sc.ctx_parts = ctx_parts;
    sc.att_stats = att_stats;
    sc.yhist = yhist;
    sc.align_raw = align_raw;
    sc.zeros = h->zeros;
    const bool predictive = h->dec.local_d > 0 && h->dec.local_predictive;
    if (predictive) {
        WS(h, "dec.p_hist", float, (size_t)n_steps * B, p_hist);
        WS(h, "dec.err_flag", int, 4, err_flag);
        sc.p_hist = p_hist;
        sc.err_flag = err_flag;
    }

    const int OUT = c.n_mels * c.reduction;
    const int64_t per_step = 2 + 2 + 1 + 1 + 2 * NL;
    ProfScope ps(h, ST_DECODER, 3 + per_step * n_steps);
    // keys = memory_layer(memory), no bias (LuongAttention; values stay the raw memory)
    if (!have_keys && (rc = attention_keys(h, memory, B, Ts, keys))) return rc;

    if (pd_kernel == 2) {
        if (!h->ws_configured) {
            HIPCHK(h, decoder_ws_configure());
            h->ws_configured = true;
        }
        // Utterances per cluster of 16 workgroups: 16 wherever the 16 * ceil(B / 16) compute units are there for the launch --
        // every unpipelined call (the whole chip), pipelined calls of up to 2 x 16 utterances (the reserved units), and a
        // pipelined call that finds the main stream idle (the first of a burst: nothing runs beside its decoder) -- else 32.
        // The same bits either way (decoder_ws.hip), so the choice may look at the clock.
        int rows = 32;
        if (decoder_ws_workgroups(B, 16) <= pd_budget) rows = 16;
        else if (h->cur_cu_budget > 0 && h->dec_chip_idle && decoder_ws_workgroups(B, 16) <= h->n_cus_dev) rows = 16;
        if (h->debug_hooks && (h->pd_rows == 16 || h->pd_rows == 32) && decoder_ws_workgroups(B, h->pd_rows) <= h->n_cus_dev)
            rows = h->pd_rows;   // (tests: "pd_rows")
        const int clusters = decoder_ws_clusters(B, 16);   // layout of the sync words: that of the form with more clusters
        WS(h, "dec.ws_scratch", float, std::max(decoder_ws_scratch_floats(B, 16), decoder_ws_scratch_floats(B, 32)), ws_scratch);
        WS(h, "dec.ws_sync", unsigned, (size_t)64 * clusters + 2, ws_sync);
        if (ws_sync != h->pd_sync || clusters != h->pd_clusters)   // new buffer / new layout: the sticky status word starts clean
            HIPCHK(h, hipMemsetAsync(ws_sync + 64 * clusters + 1, 0, sizeof(unsigned), h->stream));
        HIPCHK(h, decoder_ws_enqueue(h->stream, h->dec, ws_scratch, yhist, memory, keys, B, Ts, n_steps, alignments, ws_sync,
                                     h->cur_hold_flag, c.force_cudnn, h->debug_hooks ? h->pd_debug_delay : 0, rows, clusters));
        h->pd_rows_used = rows;
        h->pd_sync = ws_sync;
        h->pd_clusters = clusters;
        h->pd_used = true;
    } else if (use_pd) {
        if (!h->pd_configured) {
            HIPCHK(h, decoder_persistent_configure());
            h->pd_configured = true;
        }
        const int clusters = (B + 15) / 16;
        WS(h, "dec.pd_sync", unsigned, (size_t)64 * clusters + 2, pd_sync);
        if (pd_sync != h->pd_sync || clusters != h->pd_clusters)   // new buffer / new layout: the sticky status word starts clean
            HIPCHK(h, hipMemsetAsync(pd_sync + 64 * clusters + 1, 0, sizeof(unsigned), h->stream));
        HIPCHK(h, decoder_persistent_enqueue(h->stream, h->dec, sc, memory, keys, B, Ts, n_steps, alignments, pd_sync,
                                             h->cur_hold_flag, c.force_cudnn, h->debug_hooks ? h->pd_debug_delay : 0));
        h->pd_sync = pd_sync;
        h->pd_clusters = clusters;
        h->pd_used = true;
    } else if (!h->use_graph) {
        HIPCHK(h, decoder_enqueue(h->stream, h->dec, sc, memory, keys, B, Ts, n_steps, alignments, c.force_cudnn));
    } else {
        auto& k = h->dec_key;
        // (the scratch and weight structs are plain pointers and ints, zeroed before they are filled: compared bytewise)
        if (!h->dec_graph || k.memory != memory || k.keys != keys || k.align != alignments || k.B != B || k.Ts != Ts ||
            k.n_steps != n_steps || std::memcmp(&k.sc, &sc, sizeof(sc)) != 0 || std::memcmp(&k.w, &h->dec, sizeof(h->dec)) != 0) {
            if ((rc = graph_drop(h))) return rc;
            hipGraph_t graph = nullptr;
            HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            hipError_t e = decoder_enqueue(h->stream, h->dec, sc, memory, keys, B, Ts, n_steps, alignments, c.force_cudnn);
            hipError_t e2 = hipStreamEndCapture(h->stream, &graph);
            if (e != hipSuccess || e2 != hipSuccess) {
                if (graph) hipGraphDestroy(graph);
                h->err = std::string("decoder graph capture failed: ") + hipGetErrorString(e != hipSuccess ? e : e2);
                return TTS_ERR_HIP;
            }
            e = hipGraphInstantiate(&h->dec_graph, graph, nullptr, nullptr, 0);
            // (the captured graph lives as long as the executable one: see dec_graph_src)
            h->dec_graph_src = graph;
            if (e != hipSuccess) {
                hipGraphDestroy(graph);
                h->dec_graph_src = nullptr;
                h->dec_graph = nullptr;
                h->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
                return TTS_ERR_HIP;
            }
            k.memory = memory; k.keys = keys; k.align = alignments; k.B = B; k.Ts = Ts; k.n_steps = n_steps;
            k.sc = sc; k.w = h->dec;
        }
        if ((rc = graph_quiesce(h))) return rc;   // (never two launches of one executable graph in flight)
        HIPCHK(h, hipGraphLaunch(h->dec_graph, h->stream));
        if (!h->ev_graph_done) HIPCHK(h, hipEventCreateWithFlags(&h->ev_graph_done, hipEventDisableTiming));
        HIPCHK(h, hipEventRecord(h->ev_graph_done, h->stream));
        h->graph_in_flight = true;
    }
    // OutputProjectionWrapper for all steps at once: mel[b][t][:] = y[b][t] W_o + b_o
    {
        const GemmGroup proj = dense_group(yhist, U, h->dec.out_wt, h->dec.out_b, mel, OUT, B * n_steps, OUT, U, ACT_NONE);
        if (defer_proj) {
            h->pending_proj = proj;
            h->has_pending_proj = true;
        } else if ((rc = run_single(h, proj))) {
            return rc;
        }
    }
    if (predictive) {
        // a predicted window that leaves the memory: the reference fails at run time (attention.py:288-304)
        int flag = 0;
        HIPCHK(h, hipMemcpyAsync(&flag, sc.err_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (flag)
            return fail(h, TTS_ERR_UNSUPPORTED,
                        "LocalLuongAttention (predictive): a predicted attention window leaves the memory; the "
                        "reference pads such windows inconsistently and fails there too");
    }
    return TTS_OK;
}

// post-net CBHG + final Dense; with mag != null the Dense epilogue also emits the de-normalised,
// power-raised magnitude in the internal frame-major layout [B*T][FP] (fused tts_denorm_power).
static int postnet_impl(tts_handle_t h, const float* mel, int B, int T, float* linear, float* mag, float ref_db,
                        float max_db, float power, int* db_flag = nullptr) {
    int rc = check_ready(h);
    if (rc) return rc;
    // linear may be null when only the de-normalised magnitude is wanted (tts_synthesize without linear_out)
    if (!mel || (!linear && !mag) || B < 1 || T < 1) return fail(h, TTS_ERR_INVALID, "postnet_forward: bad arguments");
    const tts_config_t& c = h->cfg;
    const int M = B * T, H2 = 2 * c.n_gru_units, F = 1 + c.n_fft / 2;
    int64_t launches = 0;
    ProfScope ps(h, ST_POSTNET, 0);
    // apply_post_processing = 0 (reference tacotron/model.py:388-391): no CBHG, the final Dense reads the mel frames
    const float* dense_in = mel;
    int dense_k = c.n_mels;
    if (c.apply_post_processing) {
        WS(h, "post.gru", float, (size_t)M * H2, gru);
        if ((rc = run_cbhg(h, h->post, "post", mel, B, T, gru, &launches))) return rc;
        dense_in = gru;
        dense_k = H2;
    }
    GemmGroup g = dense_group(dense_in, dense_k, h->dense_wt, h->dense_b, linear, F, M, F, dense_k, ACT_NONE);
    if (mag) {
        g.C2 = mag;
        g.ldc2 = gl_fp(c.n_fft);
        g.N2 = gl_fp(c.n_fft);
        g.d_ref = ref_db;
        g.d_range = std::fabs(ref_db) + std::fabs(max_db);
        g.d_pow = power;
        g.d_flag = db_flag;
    }
    if ((rc = run_single(h, g))) return rc;
    ++launches;
    if (ps.idx >= 0) h->spans[ps.idx].launches = launches;
    return TTS_OK;
}

int tts_postnet_forward(tts_handle_t h, const float* mel, int B, int T, float* linear) {
    DeviceScope dev_scope(h);
    if (!linear) return fail(h, TTS_ERR_INVALID, "postnet_forward: bad arguments");
    return postnet_impl(h, mel, B, T, linear, nullptr, 0.f, 0.f, 1.f);
}

// reference audio/conversion.py:47-49: decibel_to_magnitude raises AssertionError when some dB value is below
// -100.  The lowest value inv_normalize_decibel can produce is ref - (|ref| + |max|) (clip(x) == 0): with the
// reference's constants (6.02, 99.89) that is -93.87 dB, so the assertion cannot fire and nothing is checked.
// Constants that allow it get the data-dependent check the reference makes: the de-normalising kernels raise
// a device flag, which the caller reads back (one stream synchronisation, only in that configuration).
static bool denorm_can_assert(float ref_db, float max_db) {
    return ref_db - (std::fabs(ref_db) + std::fabs(max_db)) < -100.0f;
}
static int denorm_flag_arm(tts_handle_t h, int** flag) {
    if (!h->an.flag) HIPCHK(h, hipMalloc(&h->an.flag, sizeof(int)));
    HIPCHK(h, hipMemsetAsync(h->an.flag, 0, sizeof(int), h->stream));
    *flag = h->an.flag;
    return TTS_OK;
}
static int denorm_flag_read(tts_handle_t h) {
    int flag = 0;
    HIPCHK(h, hipMemcpyAsync(&flag, h->an.flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (flag)
        return fail(h, TTS_ERR_DB_RANGE,
                    "\"conversion.decibel_to_magnitude\" was asked to convert a dB value smaller -100 dB.");
    return TTS_OK;
}

int tts_denorm_power(tts_handle_t h, const float* linear, int B, int T, int F, float ref_db, float max_db, float power,
                     float* mag) {
    DeviceScope dev_scope(h);
    if (!h || !linear || !mag || B < 1 || T < 1 || F < 1) return fail(h, TTS_ERR_INVALID, "denorm_power: bad arguments");
    int rc;
    int* flag = nullptr;
    if (denorm_can_assert(ref_db, max_db) && (rc = denorm_flag_arm(h, &flag))) return rc;
    const int FP = (F + 3) & ~3;
    WS(h, "denorm.tmp", float, (size_t)B * T * FP, tmp);
    {
        ProfScope ps(h, ST_DENORM, 2);
        HIPCHK(h, launch_denorm_power(h->stream, linear, tmp, (size_t)B * T, F, FP, ref_db, max_db, power, flag));
        HIPCHK(h, launch_tf_to_ft(h->stream, tmp, mag, B, F, T, FP));
    }
    return flag ? denorm_flag_read(h) : TTS_OK;
}

int tts_griffin_lim(tts_handle_t h, const float* mag, const float* init_phase, uint64_t seed, int B, int T, int n_iter,
                    int win_length, int hop_length, int n_fft, float* wav, float* mse) {
    DeviceScope dev_scope(h);
    if (!h || !mag || !wav || B < 1 || n_iter < 0) return fail(h, TTS_ERR_INVALID, "griffin_lim: bad arguments");
    if (!gl_is_streaming(n_fft, win_length, hop_length)) {
        // any other power-of-two n_fft / window / hop: the general kernels (griffin_lim_generic.hip)
        if (!glg_supports(n_fft)) return fail(h, TTS_ERR_UNSUPPORTED, "griffin_lim: n_fft must be a power of two between 256 and 4096");
        if (T < 1 || win_length < 2 || win_length > n_fft || hop_length < 1)
            return fail(h, TTS_ERR_INVALID, "griffin_lim: need 2 <= win_length <= n_fft, hop_length >= 1, T >= 1");
        const int Fg = 1 + n_fft / 2, Fp = gl_fp(n_fft);
        WS(h, "gl.mag", float, (size_t)B * T * Fp, magg);
        HIPCHK(h, launch_mag_ft_to_tf(h->stream, mag, magg, B, Fg, T, Fp));
        return gl_run_generic(h, magg, init_phase, seed, B, T, n_iter, win_length, hop_length, n_fft, wav, mse, false);
    }
    int rc = gl_prepare(h, T, win_length, hop_length, n_fft);
    if (rc) return rc;
    const int F = 1 + n_fft / 2, FP = TTS_GL_FP;
    WS(h, "gl.mag", float, (size_t)B * T * FP, magi);
    HIPCHK(h, launch_mag_ft_to_tf(h->stream, mag, magi, B, F, T, FP));
    return gl_run(h, magi, init_phase, seed, B, T, n_iter, win_length, hop_length, n_fft, wav, mse);
}

int tts_peak_normalize(tts_handle_t h, float* wav, int B, int n) {
    DeviceScope dev_scope(h);
    if (!h || !wav || B < 1 || n < 1) return fail(h, TTS_ERR_INVALID, "peak_normalize: bad arguments");
    HIPCHK(h, launch_peak_normalize(h->stream, wav, B, n));
    return TTS_OK;
}

int tts_stft(tts_handle_t h, const float* wav, int B, int n, int n_fft, int win_length, int hop_length, float* out) {
    DeviceScope dev_scope(h);
    if (!h || !wav || !out || B < 1) return fail(h, TTS_ERR_INVALID, "stft: bad arguments");
    float2* buf;
    int Tf;
    int rc = stft_run(h, wav, B, n, n_fft, win_length, hop_length, &buf, &Tf);
    if (rc) return rc;
    HIPCHK(h, launch_cplx_tf_to_ft(h->stream, buf, out, B, 1 + n_fft / 2, Tf, gl_fp(n_fft), 0, 1.0f));
    return TTS_OK;
}

int tts_stft_magnitude(tts_handle_t h, const float* wav, int B, int n, int n_fft, int win_length, int hop_length,
                       float power, float* lin) {
    DeviceScope dev_scope(h);
    if (!h || !wav || !lin || B < 1) return fail(h, TTS_ERR_INVALID, "stft_magnitude: bad arguments");
    float2* buf;
    int Tf;
    int rc = stft_run(h, wav, B, n, n_fft, win_length, hop_length, &buf, &Tf);
    if (rc) return rc;
    HIPCHK(h, launch_cplx_tf_to_ft(h->stream, buf, lin, B, 1 + n_fft / 2, Tf, gl_fp(n_fft), 1, power));
    return TTS_OK;
}

int tts_mel_spectrogram(tts_handle_t h, const float* lin, int B, int n_frames, int n_fft, int sr, int n_mels, float fmin,
                        float fmax, float* mel) {
    DeviceScope dev_scope(h);
    if (!h || !lin || !mel || B < 1 || n_frames < 1 || n_mels < 1 || sr < 1)
        return fail(h, TTS_ERR_INVALID, "mel_spectrogram: bad arguments");
    if (n_fft < 2 || (n_fft & 1)) return fail(h, TTS_ERR_INVALID, "mel_spectrogram: n_fft must be even");
    const int F = 1 + n_fft / 2, FP = gl_fp(n_fft);
    auto& a = h->an;
    if (!a.mel_wt || a.sr != sr || a.n_fft != n_fft || a.n_mels != n_mels || a.fmin != fmin || a.fmax != fmax) {
        // librosa.filters.mel(htk=True, norm=1) [librosa-0.6]; reference audio/features.py:75-80
        auto hz2mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
        auto mel2hz = [](double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); };
        const double fmx = fmax > 0 ? fmax : sr / 2.0;
        std::vector<double> mel_f(n_mels + 2);
        const double m0 = hz2mel(fmin), m1 = hz2mel(fmx);
        for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel2hz(m0 + (m1 - m0) * i / (n_mels + 1));
        std::vector<float> wt((size_t)n_mels * FP, 0.f);
        for (int i = 0; i < n_mels; ++i) {
            const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
            for (int f = 0; f < F; ++f) {
                const double freq = (sr / 2.0) * f / (F - 1);
                const double lower = (freq - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
                const double upper = (mel_f[i + 2] - freq) / (mel_f[i + 2] - mel_f[i + 1]);
                const double v = std::max(0.0, std::min(lower, upper));
                wt[(size_t)i * FP + f] = (float)(v * enorm);
            }
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (a.mel_wt) hipFree(a.mel_wt);
        a.mel_wt = nullptr;
        HIPCHK(h, hipMalloc(&a.mel_wt, wt.size() * sizeof(float)));
        HIPCHK(h, hipMemcpy(a.mel_wt, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
        a.sr = sr; a.n_fft = n_fft; a.n_mels = n_mels; a.fmin = fmin; a.fmax = fmax;
    }
    WS(h, "an.lin_tf", float, (size_t)B * n_frames * FP, lin_tf);
    WS(h, "an.mel_tf", float, (size_t)B * n_frames * n_mels, mel_tf);
    HIPCHK(h, launch_mag_ft_to_tf(h->stream, lin, lin_tf, B, F, n_frames, FP));
    int rc = run_single(h, dense_group(lin_tf, FP, a.mel_wt, nullptr, mel_tf, n_mels, B * n_frames, n_mels, FP, ACT_NONE));
    if (rc) return rc;
    HIPCHK(h, launch_tf_to_ft(h->stream, mel_tf, mel, B, n_mels, n_frames, n_mels));
    return TTS_OK;
}

int tts_db_convert(tts_handle_t h, const float* in, size_t n, int mode, float ref_db, float max_db, float* out) {
    DeviceScope dev_scope(h);
    if (!h || !in || !out || mode < 0 || mode > 3) return fail(h, TTS_ERR_INVALID, "db_convert: bad arguments");
    if (n == 0) return TTS_OK;
    if (mode == 1) {
        // reference audio/conversion.py:47-49: AssertionError if any dB value < -100
        if (!h->an.flag) HIPCHK(h, hipMalloc(&h->an.flag, sizeof(int)));
        HIPCHK(h, hipMemsetAsync(h->an.flag, 0, sizeof(int), h->stream));
        HIPCHK(h, launch_any_below(h->stream, in, n, -100.0f, h->an.flag));
        int flag = 0;
        HIPCHK(h, hipMemcpyAsync(&flag, h->an.flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (flag)
            return fail(h, TTS_ERR_DB_RANGE,
                        "\"conversion.decibel_to_magnitude\" was asked to convert a dB value smaller -100 dB.");
    }
    HIPCHK(h, launch_db_convert(h->stream, in, out, n, mode, ref_db, max_db));
    return TTS_OK;
}

int tts_synthesize(tts_handle_t h, const int32_t* ids, int B, int Ts, const tts_synth_params_t* sp,
                   const float* init_phase, float* wav, float* mel_out, float* align_out, float* linear_out) {
    DeviceScope dev_scope(h);
    int rc = check_ready(h);
    if (rc) return rc;
    if (!ids || !sp || !wav) return fail(h, TTS_ERR_INVALID, "synthesize: bad arguments");
    SynthScope synth_scope(h);
    const tts_config_t& c = h->cfg;
    const int T = sp->n_steps * c.reduction;
    // n_fft is a model parameter (reference tacotron/params/model.py:13-24): the final Dense has 1 + n_fft / 2 outputs, its
    // de-normalising epilogue writes rows padded to gl_fp(n_fft), and every size but 2048 reconstructs in the general kernels
    if (!glg_supports(c.n_fft))
        return fail(h, TTS_ERR_UNSUPPORTED, "synthesize: n_fft must be a power of two between 256 and 4096");
    const int F = 1 + c.n_fft / 2, FP = gl_fp(c.n_fft);
    // the model's window / hop run in the streaming kernel; any other pair in the general kernels (same results to rounding)
    const bool gl_streaming = gl_is_streaming(c.n_fft, sp->win_length, sp->hop_length);
    if (gl_streaming && (rc = gl_prepare(h, T, sp->win_length, sp->hop_length, c.n_fft))) return rc;
    if (!gl_streaming && (sp->win_length < 2 || sp->win_length > c.n_fft || sp->hop_length < 1))
        return fail(h, TTS_ERR_INVALID, "synthesize: need 2 <= win_length <= n_fft, hop_length >= 1");
    if (!gl_streaming) {
        // the general kernels' tables and workspaces, sized HERE, before anything of this call is enqueued on the front or
        // encoder streams (a growing workspace synchronises every stream; gl_run_generic finds them in place)
        if ((long long)sp->hop_length * (T - 1) <= c.n_fft / 2)
            return fail(h, TTS_ERR_INVALID, "griffin_lim: signal shorter than n_fft/2 (reflect padding undefined)");
        if ((rc = glg_prepare(h, T, sp->win_length, sp->hop_length, c.n_fft))) return rc;
        const float2* tw_unused = nullptr;
        if ((rc = glg_twiddles(h, c.n_fft, &tw_unused))) return rc;
        WS(h, "glg.phase", float2, (size_t)B * T * FP, glg_ph);
        WS(h, "glg.frames", float, (size_t)B * T * sp->win_length, glg_fr);
        WS(h, "glg.mse_partial", float, (size_t)B * T, glg_ms);
        (void)glg_ph; (void)glg_fr; (void)glg_ms;
    }
    // (one encoder output per call parity: the encoder of call k + 1 writes one while the decoder of call k reads the other)
    WS(h, "syn.memory.even", float, (size_t)B * Ts * 2 * c.n_gru_units, memory_e);
    WS(h, "syn.memory.odd", float, (size_t)B * Ts * 2 * c.n_gru_units, memory_o);
    // (alternating only where the encoder really runs ahead: under the call pipeline with the persistent decoder.  The
    //  launch-per-layer decoder replays a hipGraph with its buffers baked in -- a second `memory` would re-capture it every call)
    const bool enc_ahead_cfg = h->enc_stream && h->pipeline && (h->own_stream || h->pipeline >= 2) &&
                               h->syn_shape[0] == B && h->syn_shape[1] == Ts && h->syn_shape[2] == sp->n_steps &&
                               h->reserve_cus > 0 && pd_choice(h, B, Ts, h->reserve_cus, true) != 0;
    float* memory = (enc_ahead_cfg && (h->syn_calls & 1)) ? memory_o : memory_e;   // (syn_calls is advanced below: this call's parity)
    // the attention keys of that memory, likewise: made behind the encoder on ITS stream, so that nothing but two fills
    // stands between two decoders on the front stream (the 0.04 ms GEMM was on the step's critical path there)
    WS(h, "syn.keys.even", float, (size_t)B * Ts * c.n_attention_units, keys_e);
    WS(h, "syn.keys.odd", float, (size_t)B * Ts * c.n_attention_units, keys_o);
    float* const keys_ahead = enc_ahead_cfg ? ((h->syn_calls & 1) ? keys_o : keys_e) : nullptr;
    // The decoder output is double-buffered by call parity: the encoder / decoder of call j+1 (second
    // stream) may then run while the post-net of call j still reads its mel spectrogram.
    const int parity = (int)(h->syn_calls++ & 1);
    float* mel = mel_out;
    if (!mel) {
        WS(h, "syn.mel0", float, (size_t)B * T * c.n_mels, melb0);
        WS(h, "syn.mel1", float, (size_t)B * T * c.n_mels, melb1);
        mel = parity ? melb1 : melb0;
    }
    float* linear = linear_out;   // null: the final Dense emits only the de-normalised magnitude (rows of 1028 floats;
                                  // the 1025-float rows of the linear spectrogram cannot be written in whole cache lines)
    WS(h, "gl.mag", float, (size_t)B * T * FP, magi);
    // Under the call pipeline the initial phasors of a call are written on the FRONT stream, behind its decoder (that
    // stream has slack, the main one bounds the step): the phasor-code buffers are then a pair per call parity, so that
    // the write does not wait for the previous call's Griffin-Lim.  All four are sized here, before anything is enqueued
    // (a growing workspace synchronises every stream).
    // (its own buffers, not the pair of the stand-alone tts_griffin_lim: 4 bytes per bin, the state is a phasor code)
    WS(h, "syn.phase0.even", unsigned, (size_t)B * T * FP * (gl_state_bytes() / sizeof(unsigned)), gph0e);
    WS(h, "syn.phase1.even", unsigned, (size_t)B * T * FP * (gl_state_bytes() / sizeof(unsigned)), gph1e);
    WS(h, "syn.phase0.odd", unsigned, (size_t)B * T * FP * (gl_state_bytes() / sizeof(unsigned)), gph0o);
    WS(h, "syn.phase1.odd", unsigned, (size_t)B * T * FP * (gl_state_bytes() / sizeof(unsigned)), gph1o);
    float2* const phase_pair[2] = {reinterpret_cast<float2*>(parity ? gph0o : gph0e), reinterpret_cast<float2*>(parity ? gph1o : gph1e)};
    // Pipelined only while the library owns its stream (inputs on a borrowed stream may still be in flight) and
    // from the second call of a shape on: the first call of a new (B, Ts, n_steps) grows the workspaces, which
    // synchronises every stream -- under the CU reservation that would park the host on the sleepers' 100 ms bound.
    const bool same_shape = h->syn_shape[0] == B && h->syn_shape[1] == Ts && h->syn_shape[2] == sp->n_steps;
    h->syn_shape[0] = B; h->syn_shape[1] = Ts; h->syn_shape[2] = sp->n_steps;
    // (a borrowed stream is pipelined only on request, pipeline = 2: the caller then vouches that the inputs of a call
    //  are complete when it is made -- the library cannot tell them from the previous call's work on that stream)
    const bool pipelined = h->pipeline && (h->own_stream || h->pipeline >= 2) && same_shape;
    if (pipelined) {
        if (!h->front) {
            int prio_least = 0, prio_greatest = 0;
            HIPCHK(h, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
            HIPCHK(h, hipStreamCreateWithPriority(&h->front, hipStreamNonBlocking, prio_greatest));
            HIPCHK(h, hipStreamCreateWithPriority(&h->aux, hipStreamNonBlocking, prio_greatest));
            // (lowest priority: at the front stream's priority the encoder takes more from the post-net beside it than the
            //  decoder's head start is worth -- 17.11 against 16.87 ms per step on one box; a third queue costs the
            //  Griffin-Lim launches 3-4 % whatever its priority, which is why the step is not the decoder's 15.6 ms)
            HIPCHK(h, hipStreamCreateWithPriority(&h->encs, hipStreamNonBlocking, prio_least));
            for (int i = 0; i < 2; ++i) {
                HIPCHK(h, hipEventCreateWithFlags(&h->ev_enc_ready[i], hipEventDisableTiming));
                HIPCHK(h, hipEventCreateWithFlags(&h->ev_dec_done[i], hipEventDisableTiming));
                HIPCHK(h, hipEventCreateWithFlags(&h->ev_gap[i], hipEventDisableTiming));
            }
            HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->hold_flags), 2 * sizeof(int)));
            HIPCHK(h, cu_hold_configure());
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_aux, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_front_done, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_post_done[0], hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_post_done[1], hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_gl_done[0], hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_gl_done[1], hipEventDisableTiming));
            // calls made before these events existed recorded nothing: the front stream's first work starts behind
            // everything that is on the main stream now
            HIPCHK(h, hipEventRecord(h->ev_aux, h->stream));
            HIPCHK(h, hipStreamWaitEvent(h->front, h->ev_aux, 0));
            HIPCHK(h, hipStreamWaitEvent(h->encs, h->ev_aux, 0));
        }
    }
    hipStream_t main_stream = h->stream;
    int* hold_flag = nullptr;
    if (pipelined) {
        // The persistent decoder keeps its compute units by being resident (Griffin-Lim is planned and launched for
        // the other n_cus - reserve_cus), and the encoder in front of it may queue behind Griffin-Lim workgroups
        // without costing the step anything: no sleepers then.  The launch-per-layer decoder (configurations the
        // persistent kernel does not cover) still needs the reservation for its ~2000 dependent launches.
        const bool pd_path = h->reserve_cus > 0 && pd_choice(h, B, Ts, h->reserve_cus, true) != 0;
        // the post-net of the call two back read the mel buffer this call's decoder writes; with a caller's
        // mel buffer (possibly the same one every call) the previous call's post-net has to finish as well
        if (h->post_pending[parity]) HIPCHK(h, hipStreamWaitEvent(h->front, h->ev_post_done[parity], 0));
        // an unpipelined call in between ran its encoder and decoder on the MAIN stream, in the scratch buffers this
        // call's encoder and decoder are about to use on the front stream
        if (h->serial_pending) {
            HIPCHK(h, hipStreamWaitEvent(h->front, h->ev_serial_done, 0));
            HIPCHK(h, hipStreamWaitEvent(h->encs, h->ev_serial_done, 0));
            h->serial_pending = false;
        }
        if (mel_out && h->post_pending[parity ^ 1])
            HIPCHK(h, hipStreamWaitEvent(h->front, h->ev_post_done[parity ^ 1], 0));
        // The call two back ends its Griffin-Lim phase in launches cut for ALL compute units (gl_wide_from): this call's
        // decoder must not take 32 of them away in the middle of those, so it starts behind the post-net of the call before
        // it, i.e. behind that whole phase.  In the steady state this is where it starts anyway (its encoder runs beside that
        // post-net); it matters while a burst of calls fills the pipeline, when the decoders -- 8.9 ms against 14.5 per call
        // on the main stream -- would run ahead back to back (profiles/r05_step_timeline.txt before the gate: the wide
        // launches of calls 2 and 3 took 0.77 instead of 0.55 ms).  Only then: where the decoder is the longer stage (small
        // batches) there are no wide launches, and this wait would put the post-net into the decoders' chain.
        if (h->gl_wide_used[parity] && h->post_pending[parity ^ 1])
            HIPCHK(h, hipStreamWaitEvent(h->front, h->ev_post_done[parity ^ 1], 0));
        if (h->reserve_cus > 0 && !pd_path) {
            // reserve CUs for the front stream while the previous call's Griffin-Lim fills the rest
            hold_flag = h->hold_flags + (h->call_count++ & 1);
            // the persistent decoder releases its call's sleepers as soon as it is resident: the next set must not
            // start (and take another `reserve_cus` away from Griffin-Lim) before that decoder has finished
            if (h->front_pending) HIPCHK(h, hipStreamWaitEvent(h->aux, h->ev_front_done, 0));
            HIPCHK(h, hipMemsetAsync(hold_flag, 0, sizeof(int), h->aux));
            HIPCHK(h, hipEventRecord(h->ev_aux, h->aux));
            HIPCHK(h, launch_cu_hold(h->aux, h->reserve_cus, hold_flag, 100.0, h->hold_lds_kb));
            HIPCHK(h, hipStreamWaitEvent(h->front, h->ev_aux, 0));
            HIPCHK(h, hipStreamWaitEvent(h->encs, h->ev_aux, 0));
        }
        // the encoder: on its own stream, behind the decoder that last read this parity's `memory` (the call two back) and
        // behind the encoder before it (stream order: the encoder's scratch is one set)
        if (enc_ahead_cfg) {
            if (h->dec_done_pending[parity]) HIPCHK(h, hipStreamWaitEvent(h->encs, h->ev_dec_done[parity], 0));
            // ... which is enough only if the previous call ran encoder-ahead too.  A call in the other form (the decoder
            // form was switched in between: tts_set_option, or tts_wait_host / check_status after a decoder timeout) ran its
            // encoder AND decoder on `front`, in the one encoder scratch and in memory.even: behind its decoder, the last
            // thing recorded on that stream
            if (h->last_enc_ahead == 0 && h->dec_done_pending[parity ^ 1])
                HIPCHK(h, hipStreamWaitEvent(h->encs, h->ev_dec_done[parity ^ 1], 0));
            // The HOST waits for the gap (the call returns at most ~2.5 calls ahead of the device: back-pressure), and the
            // encoder is enqueued into an idle queue.  As a stream wait, enqueued two calls early, the barrier packet sat at
            // the head of the third queue through a whole Griffin-Lim phase, and every kernel boundary of that phase took
            // ~18 us longer (13.16 against 12.67 ms per call on one box, whatever the queue's priority).
            if (h->gap_pending[parity]) HIPCHK(h, hipEventSynchronize(h->ev_gap[parity]));
            h->stream = h->encs;
        } else {
            // the encoder in front of its decoder on the front stream (one `memory` buffer): behind whatever encoders and
            // decoders of earlier calls are still on the encoder / front streams (the front stream's own order covers the latter)
            for (int i = 0; i < 2; ++i)
                if (h->enc_ready_pending[i]) HIPCHK(h, hipStreamWaitEvent(h->front, h->ev_enc_ready[i], 0));
            h->stream = h->front;
        }
    } else if (h->encs) {
        // an unpipelined call runs its encoder and decoder on the main stream in the same scratch: behind whatever the
        // pipelined calls before it still have on the encoder and front streams
        for (int i = 0; i < 2; ++i) {
            if (h->enc_ready_pending[i]) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_enc_ready[i], 0));
            if (h->dec_done_pending[i]) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_dec_done[i], 0));
        }
    }
    if (pipelined) h->last_enc_ahead = enc_ahead_cfg ? 1 : 0;
    if (h->input_event) HIPCHK(h, hipStreamWaitEvent(h->stream, h->input_event, 0));   // (tts_synthesize_host: the ids' upload)
    rc = tts_encoder_forward(h, ids, B, Ts, memory);
    if (!rc && h->enc_done_event) HIPCHK(h, hipEventRecord(h->enc_done_event, h->stream));
    if (!rc && pipelined && keys_ahead) {
        ProfScope ps(h, ST_ENCODER, 1);
        rc = attention_keys(h, memory, B, Ts, keys_ahead);
    }
    if (pipelined) {
        if (!rc && enc_ahead_cfg) {
            HIPCHK(h, hipEventRecord(h->ev_enc_ready[parity], h->encs));
            h->enc_ready_pending[parity] = true;
            HIPCHK(h, hipStreamWaitEvent(h->front, h->ev_enc_ready[parity], 0));
        }
        h->stream = h->front;
    }
    h->cur_hold_flag = hold_flag;
    h->cur_cu_budget = (pipelined && h->reserve_cus > 0) ? h->reserve_cus : 0;
    // nothing in flight on the main stream: no post-net, no Griffin-Lim runs beside this call's decoder (the first call of a
    // burst) -- the weight-stationary decoder may then spread over twice the compute units (decoder_impl; the same bits)
    h->dec_chip_idle = pipelined && hipStreamQuery(main_stream) == hipSuccess;
    h->defer_projection = pipelined;
    h->defer_parity = parity;
    h->has_pending_proj = false;
    h->pre_keys = (pipelined && keys_ahead) ? keys_ahead : nullptr;
    if (!rc) rc = tts_decoder_forward(h, memory, B, Ts, sp->n_steps, mel, align_out);
    h->pre_keys = nullptr;
    if (!rc && pipelined) {
        HIPCHK(h, hipEventRecord(h->ev_dec_done[parity], h->front));
        h->dec_done_pending[parity] = true;
    }
    h->defer_projection = false;
    h->cur_hold_flag = nullptr;
    h->cur_cu_budget = 0;
    h->dec_chip_idle = false;
    h->stream = main_stream;
    if (!rc && !pipelined && h->front) {
        if (!h->ev_serial_done) HIPCHK(h, hipEventCreateWithFlags(&h->ev_serial_done, hipEventDisableTiming));
        HIPCHK(h, hipEventRecord(h->ev_serial_done, h->stream));
        h->serial_pending = true;
    }
    if (rc) {
        if (hold_flag) hipMemsetAsync(hold_flag, 1, sizeof(int), h->front);
        return rc;
    }
    // (a seeded start with iterations needs no initial codes at all: gl_run)
    const bool phase_on_front = gl_streaming && pipelined && sp->n_iter >= 0 && (init_phase != nullptr || sp->n_iter == 0);
    if (pipelined) {
        if (hold_flag) HIPCHK(h, hipMemsetAsync(hold_flag, 1, sizeof(int), h->front));   // release the held CUs
        if (phase_on_front) {
            // this parity's buffers were last used by the Griffin-Lim of the call two back
            if (h->gl_pending[parity]) HIPCHK(h, hipStreamWaitEvent(h->front, h->ev_gl_done[parity], 0));
            HIPCHK(h, launch_phase_init(h->front, init_phase, sp->seed, phase_pair[0], B, F, T, FP));
        }
        HIPCHK(h, hipEventRecord(h->ev_front_done, h->front));
        h->front_pending = true;
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_front_done, 0));
    }
    if (h->has_pending_proj) {   // the decoder's output projection, on the main stream (see defer_projection)
        h->has_pending_proj = false;
        if ((rc = run_single(h, h->pending_proj))) return rc;
    }
    if (h->encs) {   // the gap between two Griffin-Lim phases opens: the encoder of the next call of this parity may run
        HIPCHK(h, hipEventRecord(h->ev_gap[parity], h->stream));
        h->gap_pending[parity] = true;
    }
    int* db_flag = nullptr;
    if (denorm_can_assert(sp->ref_db, sp->max_db) && (rc = denorm_flag_arm(h, &db_flag))) return rc;
    if ((rc = postnet_impl(h, mel, B, T, linear, magi, sp->ref_db, sp->max_db, sp->power, db_flag))) return rc;
    if (db_flag && (rc = denorm_flag_read(h))) return rc;   // as the reference: no waveform for such a spectrogram
    if (h->front) {   // (also for an unpipelined call between pipelined ones: its buffers are the same ones)
        HIPCHK(h, hipEventRecord(h->ev_post_done[parity], h->stream));
        h->post_pending[parity] = true;
    }
    const int wide_from = (pipelined && gl_streaming && !h->deterministic) ? gl_wide_from(h, B, Ts, sp->n_steps, T, sp->n_iter) : -1;
    h->gl_wide_used[parity] = wide_from >= 0;
    if (gl_streaming)
        rc = gl_run(h, magi, init_phase, sp->seed, B, T, sp->n_iter, sp->win_length, sp->hop_length, c.n_fft, wav, nullptr,
                    sp->peak_normalize != 0, pipelined, phase_pair, phase_on_front, wide_from);
    else
        rc = gl_run_generic(h, magi, init_phase, sp->seed, B, T, sp->n_iter, sp->win_length, sp->hop_length, c.n_fft, wav, nullptr,
                            sp->peak_normalize != 0);
    if (h->front && !rc) {
        HIPCHK(h, hipEventRecord(h->ev_gl_done[parity], h->stream));
        h->gl_pending[parity] = true;
    }
    return rc;
}


// Host-memory form of tts_synthesize (see sstts_hip.h): uploads and downloads on copy streams, ordered by events, so that
// consecutive calls overlap exactly like calls on device-resident buffers.
int tts_synthesize_host(tts_handle_t h, const int32_t* ids_host, int B, int Ts, const tts_synth_params_t* sp, int* ticket) {
    DeviceScope dev_scope(h);
    int rc = check_ready(h);
    if (rc) return rc;
    if (!ids_host || !sp || !ticket || B < 1 || Ts < 1 || sp->n_steps < 1)
        return fail(h, TTS_ERR_INVALID, "synthesize_host: bad arguments");
    auto& io = h->hio;
    const int T = sp->n_steps * h->cfg.reduction;
    const size_t ids_bytes = (size_t)B * Ts * sizeof(int32_t);
    const size_t n_wav = (size_t)B * sp->hop_length * (size_t)(T - 1);
    const bool want_lin = (sp->host_outputs & TTS_HOST_LINEAR) != 0, want_ali = (sp->host_outputs & TTS_HOST_ALIGNMENTS) != 0;
    const size_t n_lin = want_lin ? (size_t)B * T * (size_t)(1 + h->cfg.n_fft / 2) : 0;
    const size_t n_ali = want_ali ? (size_t)sp->n_steps * B * Ts : 0;
    if (!io.in) {
        // The copy streams get the LOWEST priority: streams of one priority share a few hardware queues in creation order
        // (whatever else the process has created counts), and a copy stream that lands on the main stream's queue holds the
        // main stream's kernels behind its event waits and its 70 MB download.  Nothing else in the library uses this level.
        int prio_least = 0, prio_greatest = 0;
        HIPCHK(h, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        HIPCHK(h, hipStreamCreateWithPriority(&io.in, hipStreamNonBlocking, prio_least));
        HIPCHK(h, hipStreamCreateWithPriority(&io.out, hipStreamNonBlocking, prio_least));
        for (int i = 0; i < 3; ++i) {
            HIPCHK(h, hipEventCreateWithFlags(&io.ev_h2d[i], hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&io.ev_enc[i], hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&io.ev_ready[i], hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&io.ev_d2h[i], hipEventDisableTiming));
        }
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&io.status_pinned), 6 * sizeof(int), hipHostMallocDefault));
    }
    if (ids_bytes > io.ids_bytes || n_wav * sizeof(float) > io.wav_bytes || n_lin * sizeof(float) > io.lin_bytes ||
        n_ali * sizeof(float) > io.ali_bytes) {
        // growing the buffers: nothing of an earlier call may be in flight
        if ((rc = sync_all(h))) return rc;
        HIPCHK(h, hipStreamSynchronize(io.in));
        HIPCHK(h, hipStreamSynchronize(io.out));
        for (int i = 0; i < 3; ++i) {
            if (ids_bytes > io.ids_bytes) {
                if (io.ids_pinned[i]) HIPCHK(h, hipHostFree(io.ids_pinned[i]));
                if (io.ids_dev[i]) HIPCHK(h, hipFree(io.ids_dev[i]));
                HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&io.ids_pinned[i]), ids_bytes, hipHostMallocDefault));
                HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&io.ids_dev[i]), ids_bytes));
            }
            if (n_wav * sizeof(float) > io.wav_bytes) {
                if (io.wav_pinned[i]) HIPCHK(h, hipHostFree(io.wav_pinned[i]));
                if (io.wav_dev[i]) HIPCHK(h, hipFree(io.wav_dev[i]));
                HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&io.wav_pinned[i]), n_wav * sizeof(float), hipHostMallocDefault));
                HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&io.wav_dev[i]), n_wav * sizeof(float)));
            }
            if (n_lin * sizeof(float) > io.lin_bytes) {
                if (io.lin_pinned[i]) HIPCHK(h, hipHostFree(io.lin_pinned[i]));
                if (io.lin_dev[i]) HIPCHK(h, hipFree(io.lin_dev[i]));
                HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&io.lin_pinned[i]), n_lin * sizeof(float), hipHostMallocDefault));
                HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&io.lin_dev[i]), n_lin * sizeof(float)));
            }
            if (n_ali * sizeof(float) > io.ali_bytes) {
                if (io.ali_pinned[i]) HIPCHK(h, hipHostFree(io.ali_pinned[i]));
                if (io.ali_dev[i]) HIPCHK(h, hipFree(io.ali_dev[i]));
                HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&io.ali_pinned[i]), n_ali * sizeof(float), hipHostMallocDefault));
                HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&io.ali_dev[i]), n_ali * sizeof(float)));
            }
            io.d2h_pending[i] = io.enc_pending[i] = false;
        }
        io.ids_bytes = std::max(io.ids_bytes, ids_bytes);
        io.wav_bytes = std::max(io.wav_bytes, n_wav * sizeof(float));
        io.lin_bytes = std::max(io.lin_bytes, n_lin * sizeof(float));
        io.ali_bytes = std::max(io.ali_bytes, n_ali * sizeof(float));
    }
    const int t = io.tickets++;
    const int par = t % 3;   // (buffer set of this call; the device-side pipeline keeps its own parity)
    // the pinned staging buffer and the device copy of the ids were last used by the call three back
    if (io.d2h_pending[par]) HIPCHK(h, hipEventSynchronize(io.ev_h2d[par]));
    std::memcpy(io.ids_pinned[par], ids_host, ids_bytes);
    if (io.enc_pending[par]) HIPCHK(h, hipStreamWaitEvent(io.in, io.ev_enc[par], 0));
    HIPCHK(h, hipMemcpyAsync(io.ids_dev[par], io.ids_pinned[par], ids_bytes, hipMemcpyHostToDevice, io.in));
    HIPCHK(h, hipEventRecord(io.ev_h2d[par], io.in));
    // the waveform buffer of this set is free once the download of the call three back has left it
    if (io.d2h_pending[par]) HIPCHK(h, hipStreamWaitEvent(h->stream, io.ev_d2h[par], 0));
    h->input_event = io.ev_h2d[par];
    h->enc_done_event = io.ev_enc[par];
    // (the optional outputs of this set were last read by the download of the call three back: same event as the waveforms)
    rc = tts_synthesize(h, io.ids_dev[par], B, Ts, sp, nullptr, io.wav_dev[par], nullptr, want_ali ? io.ali_dev[par] : nullptr,
                        want_lin ? io.lin_dev[par] : nullptr);
    h->input_event = nullptr;
    h->enc_done_event = nullptr;
    if (rc) return rc;
    io.enc_pending[par] = true;
    HIPCHK(h, hipEventRecord(io.ev_ready[par], h->stream));
    HIPCHK(h, hipStreamWaitEvent(io.out, io.ev_ready[par], 0));
    HIPCHK(h, hipMemcpyAsync(io.wav_pinned[par], io.wav_dev[par], n_wav * sizeof(float), hipMemcpyDeviceToHost, io.out));
    if (want_lin) HIPCHK(h, hipMemcpyAsync(io.lin_pinned[par], io.lin_dev[par], n_lin * sizeof(float), hipMemcpyDeviceToHost, io.out));
    if (want_ali) HIPCHK(h, hipMemcpyAsync(io.ali_pinned[par], io.ali_dev[par], n_ali * sizeof(float), hipMemcpyDeviceToHost, io.out));
    io.n_lin[par] = n_lin;
    io.n_ali[par] = n_ali;
    // the sticky status words of the persistent kernels travel with the waveforms (tts_wait_host must not wait for
    // anything but this call: a stream synchronisation there would wait for the NEXT call's download as well)
    io.status_pinned[2 * par] = io.status_pinned[2 * par + 1] = 0;
    io.failed[par] = false;   // (the set is reused: the ticket that failed can no longer be waited on)
    if (h->pd_used && h->pd_sync)
        HIPCHK(h, hipMemcpyAsync(&io.status_pinned[2 * par + 1], h->pd_sync + 64 * h->pd_clusters + 1, sizeof(int),
                                 hipMemcpyDeviceToHost, io.out));
    HIPCHK(h, hipEventRecord(io.ev_d2h[par], io.out));
    io.d2h_pending[par] = true;
    io.n_floats[par] = n_wav;
    *ticket = t;
    return TTS_OK;
}

int tts_wait_host(tts_handle_t h, int ticket, const float** wav_host, size_t* n_floats) {
    DeviceScope dev_scope(h);
    if (!h || !wav_host) return TTS_ERR_INVALID;
    auto& io = h->hio;
    if (ticket < 0 || ticket >= io.tickets || ticket < io.tickets - 3)
        return fail(h, TTS_ERR_INVALID, "wait_host: this ticket's buffer has been handed to a later call (at most three calls in flight)");
    const int par = ticket % 3;
    HIPCHK(h, hipEventSynchronize(io.ev_d2h[par]));
    // the download is behind everything the call launched: a timed-out persistent kernel must not pass for a result
    if (io.status_pinned[2 * par + 1] || io.failed[par]) {
        // what check_status does at a synchronisation, on the first report: the sticky device word is cleared (behind the
        // downloads already queued: a call in flight behind this one may still be reported once, conservatively), the
        // handle leaves the persistent path by itself and stops carrying the word along.  The buffer set stays marked:
        // a second wait on this ticket (tts_wait_host after a failed tts_wait_host_outputs) must not hand out its waveforms
        if (!io.failed[par]) {
            io.failed[par] = true;
            io.status_pinned[2 * par + 1] = 0;
            if (h->pd_sync) HIPCHK(h, hipMemsetAsync(h->pd_sync + 64 * h->pd_clusters + 1, 0, sizeof(int), io.out));
            h->persistent_decoder = 0;
            h->pd_used = false;
        }
        return fail(h, TTS_ERR_HIP,
                    "persistent decoder: a workgroup waited for its cluster longer than the bound (not all "
                    "workgroups were co-resident); the outputs of that call are invalid -- the handle has "
                    "switched to the launch-per-layer path (tts_set_option(h, \"persistent_decoder\", 1) switches back)");
    }
    *wav_host = io.wav_pinned[par];
    if (n_floats) *n_floats = io.n_floats[par];
    return TTS_OK;
}

int tts_wait_host_outputs(tts_handle_t h, int ticket, const float** linear_host, size_t* n_linear, const float** align_host,
                          size_t* n_align) {
    DeviceScope dev_scope(h);
    if (!h) return TTS_ERR_INVALID;
    const float* wav = nullptr;
    const int rc = tts_wait_host(h, ticket, &wav, nullptr);   // same event, same checks (ticket range, decoder status)
    if (rc) return rc;
    auto& io = h->hio;
    const int par = ticket % 3;
    if (linear_host) *linear_host = io.n_lin[par] ? io.lin_pinned[par] : nullptr;
    if (n_linear) *n_linear = io.n_lin[par];
    if (align_host) *align_host = io.n_ali[par] ? io.ali_pinned[par] : nullptr;
    if (n_align) *n_align = io.n_ali[par];
    return TTS_OK;
}

int tts_decoder_kernel_choice(tts_handle_t h, int B, int Ts, int pipelined) {
    if (!h || B < 1 || Ts < 1) return TTS_ERR_INVALID;
    if (!h->finalized) return fail(h, TTS_ERR_NOT_LOADED, "decoder_kernel_choice: weights not finalised");
    return pd_choice(h, B, Ts, pipelined ? h->reserve_cus : h->n_cus_dev, pipelined != 0);
}

int tts_debug_workspace(tts_handle_t h, const char* name, void** dptr, size_t* bytes) {
    DeviceScope dev_scope(h);
    if (!h || !name) return TTS_ERR_INVALID;
    auto it = h->ws.find(name);
    if (it == h->ws.end()) return fail(h, TTS_ERR_INVALID, std::string("no workspace buffer ") + name);
    if (dptr) *dptr = it->second.p;
    if (bytes) *bytes = it->second.bytes;
    return TTS_OK;
}

// Host-only view of the Griffin-Lim work-item planner (no GPU needed): classes[4][2] = {frames per run, runs
// per utterance} in execution order, *max_item_frames = the chunk size the runs are processed in; returns the
// number of classes or a negative status.
int tts_debug_gl_plan(int T, int B, int win_length, int hop_length, int n_workers, int* classes, int* max_item_frames) {
    if (T < 1 || B < 1 || win_length < 2 || win_length > TTS_GL_NFFT || hop_length < 1 || n_workers < 1 || !classes)
        return TTS_ERR_INVALID;
    GlParams p;
    std::memset(&p, 0, sizeof(p));
    p.T = T; p.B = B; p.win = win_length; p.hop = hop_length;
    p.ncol = (win_length + hop_length - 1) / hop_length;
    const int ring = gl_stream_ring_frames(win_length, hop_length);
    if (p.ncol > 8 || ring < 1) return TTS_ERR_UNSUPPORTED;
    int n_stage = 3;   // the handle's default launch form (option "gl_pair")
    while (n_stage > 1 && gl_stream_ring_frames(win_length, hop_length, n_stage) <= 0) --n_stage;
    gl_plan_stream(p, n_workers, n_stage);
    if (max_item_frames) *max_item_frames = ring;
    for (int k = 0; k < GL_MAX_CLASSES; ++k) {
        classes[2 * k] = p.cls_C[k];
        classes[2 * k + 1] = p.cls_n[k];
    }
    return p.n_classes;
}

// Diagnostic: one launch of the GEMM kernel, C[M][N] = conv(A)[M][ktaps*Cin] . Wt[N][K]^T (device pointers),
// optionally with the max-pool loader; for tools/gemm_bench.py.
int tts_debug_gemm(tts_handle_t h, const float* A, const float* Wt, float* C, int M, int N, int Cin, int ktaps, int T,
                   int pool) {
    DeviceScope dev_scope(h);
    if (!h || !A || !Wt || !C || M < 1 || N < 1 || Cin < 4 || (Cin & 3) || ktaps < 1 || T < 1 || M % T) return TTS_ERR_INVALID;
    GemmGroup g = conv_group(A, Cin, ktaps, T, Wt, nullptr, nullptr, nullptr, C, N, 0, M, N, ACT_NONE, pool);
    {   // the caller's weights: their image is made again on every call (outside the timed span)
        int rc = gemm_attach_image(h, g, true);
        if (rc) return rc;
    }
    ProfScope ps(h, ST_DEBUG_GEMM, 1);
    const int slices = gemm_splitk_slices(g.K);   // same rule as the CBHG projections
    if (slices > 1) {
        WS(h, "debug.splitk", float, (size_t)slices * M * N, part);
        HIPCHK(h, launch_gemm_splitk(h->stream, g, slices, part, h->gemm_ps));
        return TTS_OK;
    }
    GemmBatch b;
    std::memset(&b, 0, sizeof(b));
    b.g[0] = g;
    b.ps = h->gemm_ps;
    HIPCHK(h, launch_gemm(h->stream, b, 1));
    return TTS_OK;
}

// Diagnostic: occupy `n_wgs` workgroup slots of `lds_kb` KB LDS each for `ms` milliseconds on a private
// stream (to study how the other kernels behave on a partially occupied GPU).  Not part of the product path.
int tts_debug_hold(tts_handle_t h, int n_wgs, int lds_kb, double ms) {
    DeviceScope dev_scope(h);
    if (!h || n_wgs < 1 || lds_kb < 1 || lds_kb > 160) return TTS_ERR_INVALID;
    if (!h->debug_hooks) return fail(h, TTS_ERR_INVALID, "tts_debug_hold: a diagnostic; set the option \"debug_hooks\" to 1 on this handle first");
    static hipStream_t dbg = nullptr;
    static int* never = nullptr;
    static std::mutex dbg_mutex;
    std::lock_guard<std::mutex> lock(dbg_mutex);
    if (!dbg) {
        HIPCHK(h, hipStreamCreateWithFlags(&dbg, hipStreamNonBlocking));
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&never), sizeof(int)));
        HIPCHK(h, hipMemset(never, 0, sizeof(int)));
    }
    HIPCHK(h, cu_hold_configure());
    HIPCHK(h, launch_cu_hold(dbg, n_wgs, never, ms, lds_kb));
    return TTS_OK;
}

int tts_profile_reset(tts_handle_t h) {
    DeviceScope dev_scope(h);
    if (!h) return TTS_ERR_INVALID;
    prof_collect(h);
    for (int i = 0; i < ST_COUNT; ++i) {
        h->prof_ms[i] = 0;
        h->prof_launches[i] = 0;
    }
    return TTS_OK;
}

int tts_profile_get(tts_handle_t h, const char* stage, float* ms_total, int64_t* launches) {
    DeviceScope dev_scope(h);
    if (!h || !stage) return TTS_ERR_INVALID;
    prof_collect(h);
    for (int i = 0; i < ST_COUNT; ++i)
        if (!std::strcmp(stage, kStageNames[i])) {
            if (ms_total) *ms_total = (float)h->prof_ms[i];
            if (launches) *launches = h->prof_launches[i];
            return TTS_OK;
        }
    return fail(h, TTS_ERR_INVALID, std::string("unknown stage ") + stage);
}

int tts_device_info(tts_handle_t h, char uuid_hex[33], int* n_compute_units) {
    DeviceScope dev_scope(h);
    if (!h) return TTS_ERR_INVALID;
    if (uuid_hex) {
        hipUUID id;
        HIPCHK(h, hipDeviceGetUuid(&id, h->device));
        static const char* hex = "0123456789abcdef";
        for (int i = 0; i < 16; ++i) {
            uuid_hex[2 * i] = hex[((unsigned char)id.bytes[i]) >> 4];
            uuid_hex[2 * i + 1] = hex[((unsigned char)id.bytes[i]) & 15];
        }
        uuid_hex[32] = 0;
    }
    if (n_compute_units) *n_compute_units = h->n_cus_dev;
    return TTS_OK;
}

}  // extern "C"
