// C ABI, part 1: the handle -- lifecycle, streams and options, the grow-only workspace, weights (manifest, packing into the
// arena), profiling spans, allocation and copies.  include/sstts_hip.h is the interface; api_internal.h what the parts share.
#include "api_internal.h"

namespace tts_api {
thread_local std::string g_create_error;
const char* const kStageNames[ST_COUNT] = {"encoder", "decoder", "postnet", "denorm", "gl_iter", "gl_final", "debug_gemm"};
}  // namespace tts_api

namespace tts_api {


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
int gemm_attach_image(tts_handle_t h, GemmGroup& g, bool refresh) {
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


// Is the HIP runtime this PROCESS runs on at least the one the library was built with (major.minor)?  A host that loaded another
// ROCm's libamdhip64 first (import torch: PyTorch bundles its own) serves the library with that one -- same soname.
bool graph_runtime_ok(int* have) {
    int v = 0;
    if (hipRuntimeGetVersion(&v) != hipSuccess) v = 0;
    if (have) *have = v;
    return v / 100000 >= HIP_VERSION / 100000;   // HIP_VERSION = major * 10^7 + minor * 10^5 + patch
}

}  // namespace tts_api

// ======================================================================================== C ABI
extern "C" {


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
                        "PyTorch bundles); hipGraph replays of the decoder are wrong there (csrc/api_internal.h, `use_graph`) -- the "
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


int tts_debug_workspace(tts_handle_t h, const char* name, void** dptr, size_t* bytes) {
    DeviceScope dev_scope(h);
    if (!h || !name) return TTS_ERR_INVALID;
    auto it = h->ws.find(name);
    if (it == h->ws.end()) return fail(h, TTS_ERR_INVALID, std::string("no workspace buffer ") + name);
    if (dptr) *dptr = it->second.p;
    if (bytes) *bytes = it->second.bytes;
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
