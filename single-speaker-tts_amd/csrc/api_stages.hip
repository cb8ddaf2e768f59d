// C ABI, part 2: the stages -- what tts_encoder_forward / tts_decoder_forward / tts_postnet_forward / tts_denorm_power /
// tts_griffin_lim / tts_stft... enqueue (CBHG, the decoder forms and their choice, Griffin-Lim planning and launches, the
// general power-of-two path), and the diagnostic entry points.
#include "api_internal.h"

namespace tts_api {


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


int device_cus(tts_handle_t h) { return h->gl.n_cus; }


int gl_fp(int n_fft);

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
int gl_fp(int n_fft) { return ((n_fft / 2 + 1) + 31) & ~31; }   // padded row length (TTS_GL_FP for 2048)
// The streaming kernel is specialised to the model's configuration; everything else takes the general kernels.
bool gl_is_streaming(int n_fft, int win, int hop) { return n_fft == TTS_GL_NFFT && gl_stream_instantiated(win, hop); }

// periodic hann (scipy get_window('hann', win, fftbins=True)), float64 then float32
void hann_window(int win, std::vector<double>& wd, std::vector<float>& wf) {
    wd.resize(win);
    wf.resize(win);
    for (int i = 0; i < win; ++i) {
        wd[i] = 0.5 - 0.5 * std::cos(2.0 * M_PI * i / win);
        wf[i] = (float)wd[i];
    }
}

// librosa window_sumsquare (float32 buffer, sequential += of the padded squared window) as its RECIPROCAL where librosa's
// istft divides (wss > tiny(float32)), 1 elsewhere
void recip_window_sumsquare(const std::vector<double>& wd, int n_fft, int hop, int T, std::vector<float>& wss) {
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
    if (!gl_is_streaming(n_fft, win, hop)) return fail(h, TTS_ERR_UNSUPPORTED, "griffin_lim: the streaming kernel is instantiated for 1102 / 275 and 800 / 200 at n_fft 2048 only");
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
           int win, int hop, int n_fft, float* wav, float* mse, bool peak_normalize,
           bool under_reservation, float2* const* phase_pair, bool phase_ready,
           int wide_from) {
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
    // (the cut decides who does which frames, never the waveform's bits: every sample is summed over the frames that cover it in
    //  ascending order whatever run they lie in -- tests/test_gpu_audio.py::test_griffin_lim_bits_do_not_depend_on_the_cut)
    HIPCHK(h, gl_plan_stream(p, n_cus - held > 16 ? n_cus - held : n_cus, per_launch, h->debug_hooks ? h->gl_runs : 0,
                             h->debug_hooks ? h->gl_run_len : 0, h->stream));
    // wide_from >= 0 (the pipelined tts_synthesize, see gl_wide_from there): launches from that index on are cut for ALL
    // compute units -- the second stream's decoder has left its share by then.  A second cut of the same frames.
    GlParams pw = p;
    const bool two_cuts = held > 0 && wide_from >= 0 && n_cus - held > 16 &&
                          !(h->debug_hooks && (h->gl_runs || h->gl_run_len));
    if (two_cuts) HIPCHK(h, gl_plan_stream(pw, n_cus, per_launch, 0, 0, h->stream));
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
            if (getenv("GL_TIMELINE_WGS"))   // every workgroup: block, XCC, first item, runs, start, end
                for (int w = 0; w < 512; ++w)
                    if (host[2 * w]) {
                        const unsigned long long m = host[1536 + w];
                        fprintf(stderr, "wg %3d xcc %d item %4d runs %d start %.1f end %.1f\n", w, (int)((m >> 48) & 0xf), (int)(m & 0xffffffffu),
                                (int)((m >> 32) & 0xffff), (host[2 * w] - t0) * 0.01, (host[2 * w + 1] - t0) * 0.01);
                    }
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


// ---------------------------------------------------------------------------------------- stages
// A stage entry point called by the USER (not by tts_synthesize) runs on the main stream in the one set of enc.* / dec.*
// workspaces that the pipelined calls use on the encoder and front streams: it starts behind whatever those streams still
// hold, and the next pipelined call's encoder and decoder start behind it (ev_serial_done, as for an unpipelined
// tts_synthesize).  Stream order alone covers the post-net (main stream on both sides).
int standalone_begin(tts_handle_t h) {
    if (h->in_synthesize || !h->encs) return TTS_OK;
    for (int i = 0; i < 2; ++i) {
        if (h->enc_ready_pending[i]) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_enc_ready[i], 0));
        if (h->dec_done_pending[i]) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_dec_done[i], 0));
    }
    return TTS_OK;
}

int standalone_end(tts_handle_t h) {
    if (h->in_synthesize || !h->front) return TTS_OK;
    if (!h->ev_serial_done) HIPCHK(h, hipEventCreateWithFlags(&h->ev_serial_done, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(h->ev_serial_done, h->stream));
    h->serial_pending = true;
    return TTS_OK;
}

int encoder_impl(tts_handle_t h, const int32_t* ids, int B, int Ts, float* memory) {
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
int pd_kernel_for(tts_handle_t h, int B, int Ts, int budget) {
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
int pd_choice(tts_handle_t h, int B, int Ts, int budget, bool pipelined) {
    if (h->persistent_decoder <= 0) return 0;
    const int k = pd_kernel_for(h, B, Ts, budget);
    if (h->persistent_decoder >= 2) return k;
    if (k == 2) return 2;
    if (k == 1) return (pipelined && B > 48) ? 1 : 0;
    return 0;
}


// keys = memory_layer(memory), no bias (LuongAttention, reference tacotron/model.py:205-223; the values stay the raw memory)
int attention_keys(tts_handle_t h, const float* memory, int B, int Ts, float* keys) {
    const int A = h->cfg.n_attention_units, mem = 2 * h->cfg.n_gru_units;
    return run_single(h, dense_group(memory, mem, h->mem_wt, nullptr, keys, A, B * Ts, A, mem, ACT_NONE));
}

int decoder_impl(tts_handle_t h, const float* memory, int B, int Ts, int n_steps, float* mel, float* alignments) {
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
                                     h->cur_hold_flag, c.force_cudnn, h->debug_hooks ? h->pd_debug_delay : 0, rows, clusters, sc.p_hist,
                                     sc.err_flag));
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
int postnet_impl(tts_handle_t h, const float* mel, int B, int T, float* linear, float* mag, float ref_db,
                        float max_db, float power, int* db_flag) {
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


// reference audio/conversion.py:47-49: decibel_to_magnitude raises AssertionError when some dB value is below
// -100.  The lowest value inv_normalize_decibel can produce is ref - (|ref| + |max|) (clip(x) == 0): with the
// reference's constants (6.02, 99.89) that is -93.87 dB, so the assertion cannot fire and nothing is checked.
// Constants that allow it get the data-dependent check the reference makes: the de-normalising kernels raise
// a device flag, which the caller reads back (one stream synchronisation, only in that configuration).
bool denorm_can_assert(float ref_db, float max_db) {
    return ref_db - (std::fabs(ref_db) + std::fabs(max_db)) < -100.0f;
}

int denorm_flag_arm(tts_handle_t h, int** flag) {
    if (!h->an.flag) HIPCHK(h, hipMalloc(&h->an.flag, sizeof(int)));
    HIPCHK(h, hipMemsetAsync(h->an.flag, 0, sizeof(int), h->stream));
    *flag = h->an.flag;
    return TTS_OK;
}

int denorm_flag_read(tts_handle_t h) {
    int flag = 0;
    HIPCHK(h, hipMemcpyAsync(&flag, h->an.flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (flag)
        return fail(h, TTS_ERR_DB_RANGE,
                    "\"conversion.decibel_to_magnitude\" was asked to convert a dB value smaller -100 dB.");
    return TTS_OK;
}

}  // namespace tts_api

// ======================================================================================== C ABI
extern "C" {

int tts_encoder_forward(tts_handle_t h, const int32_t* ids, int B, int Ts, float* memory) {
    DeviceScope dev_scope(h);
    int rc = check_ready(h);
    if (rc) return rc;
    if (!ids || !memory || B < 1 || Ts < 1) return fail(h, TTS_ERR_INVALID, "encoder_forward: bad arguments");
    if ((rc = standalone_begin(h))) return rc;
    if ((rc = encoder_impl(h, ids, B, Ts, memory))) return rc;
    return standalone_end(h);
}

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


int tts_postnet_forward(tts_handle_t h, const float* mel, int B, int T, float* linear) {
    DeviceScope dev_scope(h);
    if (!linear) return fail(h, TTS_ERR_INVALID, "postnet_forward: bad arguments");
    return postnet_impl(h, mel, B, T, linear, nullptr, 0.f, 0.f, 1.f);
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


// Host-only view of the Griffin-Lim work-item planner (no GPU needed): items[n][4] = {utterance, first frame, frames, slot word}
// in the order the workgroups draw them (at most max_items are written), *ring_frames = frames the kernel's LDS ring holds;
// returns the number of items or a negative status.
int tts_debug_gl_plan(int T, int B, int win_length, int hop_length, int n_workers, int* items, int max_items, int* ring_frames) {
    if (T < 1 || B < 1 || win_length < 2 || win_length > TTS_GL_NFFT || hop_length < 1 || n_workers < 1 || !items || max_items < 0)
        return TTS_ERR_INVALID;
    const int ring = gl_stream_ring_frames(win_length, hop_length);
    if ((win_length + hop_length - 1) / hop_length > 8 || ring < 1) return TTS_ERR_UNSUPPORTED;
    int n_stage = 3;   // the handle's default launch form (option "gl_pair")
    while (n_stage > 1 && gl_stream_ring_frames(win_length, hop_length, n_stage) <= 0) --n_stage;
    std::vector<int4> v;
    int slots = 0;
    const int n = gl_plan_items(T, B, win_length, hop_length, n_workers, n_stage, 0, 0, &v, &slots);
    if (ring_frames) *ring_frames = ring;
    for (int k = 0; k < n && k < max_items; ++k) {
        items[4 * k] = v[k].x; items[4 * k + 1] = v[k].y; items[4 * k + 2] = v[k].z; items[4 * k + 3] = v[k].w;
    }
    return n;
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

}  // extern "C"
