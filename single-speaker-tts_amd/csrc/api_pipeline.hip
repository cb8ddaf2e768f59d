// C ABI, part 3: the call pipeline -- tts_synthesize (three calls in flight on the main, front and encoder streams, events per
// call parity, the wide Griffin-Lim launches), its host-memory form and the tickets of tts_wait_host.
#include "api_internal.h"

namespace tts_api {


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
int gl_wide_from(tts_handle_t h, int B, int Ts, int n_steps, int T, int n_iter) {
    if (h->gl_wide == -2 || h->reserve_cus <= 0) return -1;
    const int pd = pd_choice(h, B, Ts, h->reserve_cus, true);
    if (pd == 0) return -1;                    // launch-per-layer decoder: sleeper workgroups hold the units through the whole phase
    if (h->gl_wide >= 0) return h->gl_wide;    // (tools: an explicit launch index)
    if (pd != 2) return -1;                    // the streamed-weights decoder outlasts Griffin-Lim
    const int per_launch = h->gl_pair < 1 ? 1 : (h->gl_pair > 3 ? 3 : h->gl_pair);
    // The constants were measured on 256 compute units with 32 reserved (224 for Griffin-Lim) at T_s = 150: a launch scales
    // with the units Griffin-Lim really has, a decoder step with the memory length through its attention phase (8.9 of 44.5 us
    // at T_s = 150: keys and values of the whole memory per step, decoder_ws.hip).  On a device of another size the model is
    // not trusted at all: no wide launches there (they only ever cost time, never bits, but a wrong guess costs 0.5 ms a launch).
    if (h->n_cus_dev != 256 || h->n_cus_dev - h->reserve_cus < 16) return -1;
    const double launch_ms = 3.125e-6 * (double)B * T * per_launch * 224.0 / (double)(h->n_cus_dev - h->reserve_cus);
    const double step_ms = (0.0356 + 0.0089 * (double)Ts / 150.0) * (h->cfg.force_cudnn ? 0.038 / 0.045 : 1.0);   // (seven hand-offs per step instead of ten)
    const double dec_ms = step_ms * n_steps + 0.1;
    const int n_launches = (n_iter + per_launch - 1) / per_launch;
    const int from = (int)std::ceil((dec_ms + 0.5 * launch_ms + 0.1) / launch_ms);
    return from <= n_launches ? from : -1;
}

}  // namespace tts_api

// ======================================================================================== C ABI
extern "C" {


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
    const int wide_from = (pipelined && gl_streaming) ? gl_wide_from(h, B, Ts, sp->n_steps, T, sp->n_iter) : -1;
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

}  // extern "C"
