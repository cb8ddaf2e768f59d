"""GPU: BASELINE.json's full-size configurations -- direct oracle comparison where the oracle
finishes in seconds, size-independent properties (shard invariance, softmax normalisation,
determinism, round trips) at the sizes where it does not."""
import numpy as np
import pytest

from conftest import pkg, rel_l2
from oracle import audio_oracle as A
from oracle import tacotron_oracle as O

pytestmark = pytest.mark.gpu
N_FFT, WIN, HOP = 2048, 1102, 275


def bench_ids(B, Ts, seed):
    rng = np.random.default_rng(seed)
    ids = np.zeros((B, Ts), np.int32)
    for b in range(B):
        L = int(np.clip(round(rng.normal(100, 30)), 20, Ts - 1))
        ids[b, :L] = rng.integers(2, 39, L)
        ids[b, L] = 1
    return ids


def test_config2_encoder_32x150(engine, hparams, weights64):
    ids = bench_ids(32, 150, 1234)
    ref = O.encoder(ids, weights64, hparams)
    got = engine.encoder_forward(ids).to_host()
    e = rel_l2(got, ref)
    print('config 2 encoder 32x150 rel-L2', e)
    assert e < 1e-3


def test_config3_decoder_200_steps_b64(engine, hparams, weights64):
    """200 autoregressive steps: the fp32 error must not grow past the 1e-3 parity bar."""
    rng = np.random.default_rng(7)
    memory = (rng.standard_normal((64, 150, 256)) * 0.5).astype(np.float32)
    ref_mel, ref_al = O.decoder(memory.astype(np.float64), weights64, hparams)
    mel, al = engine.decoder_forward(memory, 200)
    mel, al = mel.to_host(), al.to_host()
    assert mel.shape == (64, 200, 400) and al.shape == (200, 64, 150)
    e = rel_l2(mel, ref_mel)
    e_last = rel_l2(mel[:, -1], ref_mel[:, -1])
    print('config 3 decoder B=64 S=200: mel rel-L2 {:.3e} (last step {:.3e}), align max-abs {:.3e}'.format(
        e, e_last, float(np.abs(al - ref_al).max())))
    assert e < 1e-3 and e_last < 1e-3
    assert np.abs(al - ref_al).max() < 1e-4
    assert np.allclose(al.sum(-1), 1.0, atol=1e-5)


def test_shard_invariance(engine):
    """Utterance i gives bit-identical results alone, in a shard of 8 and in the batch of 64:
    the property that makes the multi-GPU utterance sharding exact."""
    ids = bench_ids(64, 150, 99)
    mem = engine.encoder_forward(ids).to_host()
    mem8 = engine.encoder_forward(ids[16:24]).to_host()
    assert np.array_equal(mem[16:24], mem8)
    mel, _ = engine.decoder_forward(mem, 12, want_alignments=False)
    mel8, _ = engine.decoder_forward(mem[16:24], 12, want_alignments=False)
    assert np.array_equal(mel.to_host()[16:24], mel8.to_host())
    m = mel.to_host().reshape(64, -1, 80)
    lin = engine.postnet_forward(m).to_host()
    lin1 = engine.postnet_forward(m[17:18]).to_host()
    assert np.array_equal(lin[17:18], lin1)


def test_config4_postnet_full_length(engine, hparams, weights64):
    rng = np.random.default_rng(11)
    mel = rng.random((4, 1000, 80)).astype(np.float32)
    ref = O.post_process(mel.astype(np.float64), weights64, hparams)
    got = engine.postnet_forward(mel).to_host()
    e = rel_l2(got, ref)
    print('config 4 post-net T=1000 rel-L2', e)
    assert got.shape == (4, 1000, 1025) and e < 1e-3


def test_config4_griffin_lim_full_length(engine):
    rng = np.random.default_rng(42)
    T = 1000
    n = HOP * (T - 1)
    t = np.arange(n) / 22050.0
    y = (0.3 * np.sin(2 * np.pi * 200 * t * (1 + 0.2 * np.sin(2 * np.pi * 0.7 * t))) + 0.02 * rng.standard_normal(n)).astype(np.float32)
    mag1 = np.abs(A.stft(y, N_FFT, HOP, WIN)).astype(np.float32)
    assert mag1.shape == (1025, 1000)
    init = rng.random((1, 1025, T)).astype(np.float32)
    ref_wav, ref_mse = A.griffin_lim_v2(mag1, WIN, HOP, N_FFT, 3, init_phase=init[0])
    wav, mse = engine.griffin_lim(mag1[None], 3, WIN, HOP, N_FFT, init_phase=init)
    assert wav.shape == (1, 274725)
    e = rel_l2(wav.to_host()[0], ref_wav)
    print('config 4 GL T=1000, 3 iterations: wav rel-L2 {:.3e}, mse {} vs {}'.format(e, mse.to_host()[0], ref_mse))
    assert e < 1e-3 and abs(mse.to_host()[0] - ref_mse) < 1e-3 * ref_mse
    # batch of 64 copies: no cross-utterance coupling.  Rows of one batch are bitwise identical; against the
    # single run only the partition into work items (hence the overlap-add summation order) may differ
    big = engine.to_device(np.broadcast_to(mag1, (64,) + mag1.shape).copy())
    init64 = engine.to_device(np.broadcast_to(init[0], (64,) + init[0].shape).copy())
    w64, m64 = engine.griffin_lim(big, 3, WIN, HOP, N_FFT, init_phase=init64)
    w64 = w64.to_host()
    assert np.array_equal(w64[63], w64[0]) and np.array_equal(w64[31], w64[0])
    assert rel_l2(w64[0], wav.to_host()[0]) < 1e-5
    w64b, _ = engine.griffin_lim(big, 3, WIN, HOP, N_FFT, init_phase=init64)
    assert np.array_equal(w64b.to_host(), w64)          # run-to-run bitwise reproducible
    # mse decreases with iterations; round trip: |STFT(iSTFT(S))| of a consistent S returns S
    _, m10 = engine.griffin_lim(mag1[None], 10, WIN, HOP, N_FFT, init_phase=init)
    assert m10.to_host()[0] < mse.to_host()[0]


def test_end_to_end_synthesize_matches_staged(engine, hparams):
    ids = bench_ids(3, 40, 5)
    init = np.random.default_rng(1).random((3, 1025, 50)).astype(np.float32)
    out = engine.synthesize(ids, 10, 6.02, 99.89, 1.3, 4, WIN, HOP, init_phase=init, peak_normalize=True,
                            want_mel=True, want_alignments=True, want_linear=True)
    mem = engine.encoder_forward(ids)
    mel, al = engine.decoder_forward(mem, 10)
    lin = engine.postnet_forward(mel.to_host().reshape(3, 50, 80))
    mag = engine.denorm_power(lin, 6.02, 99.89, 1.3)
    wav, _ = engine.griffin_lim(mag, 4, WIN, HOP, N_FFT, init_phase=init, want_mse=False)   # same kernel variant
    wav = engine.peak_normalize(wav)
    assert np.array_equal(out['mel'].to_host().reshape(3, 10, 400), mel.to_host())
    assert np.array_equal(out['linear'].to_host(), lin.to_host())
    assert np.array_equal(out['alignments'].to_host(), al.to_host())
    # the fused path de-normalises in the Dense epilogue and peak-normalises in the last iSTFT: same
    # arithmetic, different instruction selection -> equal to rounding, not bitwise
    assert rel_l2(out['wav'].to_host(), wav.to_host()) < 1e-4
    assert np.abs(out['wav'].to_host()).max(axis=1).tolist() == [1.0, 1.0, 1.0]


def test_pipelined_calls_equal_sequential(engine):
    """Back-to-back tts_synthesize calls overlap (decoder of call k+1 under Griffin-Lim of call k,
    shared scratch, CU reservation): results must be bit-identical to fully serialised calls."""
    # one shape throughout: the first call of a new (B, Ts, n_steps) always runs unpipelined (it sizes the workspaces)
    batches = [bench_ids(4, 30, 40 + i) for i in range(4)]
    inits = [np.random.default_rng(i).random((4, 1025, 40)).astype(np.float32) for i in range(4)]

    def run(pipeline):
        engine.set_option('pipeline', pipeline)
        dev_ids = [engine.to_device(b) for b in batches]
        dev_init = [engine.to_device(x) for x in inits]
        outs = [engine.synthesize(dev_ids[i], 8, 6.02, 99.89, 1.3, 6, WIN, HOP, init_phase=dev_init[i],
                                  want_mel=True, want_alignments=True, want_linear=True) for i in range(4)]
        engine.synchronize()
        return [{k: v.to_host() for k, v in o.items()} for o in outs]

    # the pipelined call runs the decoder as the persistent kernel by default, the serialised one as the
    # launch-per-layer graph (equal to rounding only): compare like with like, both ways
    try:
        pip = {}
        for pd in (0, 2):
            engine.set_option('persistent_decoder', pd)
            seq = run(0)
            pip[pd] = run(1)
            for a, b in zip(seq, pip[pd]):
                for k in a:
                    assert np.array_equal(a[k], b[k]), (pd, k)
        # the default (round 5): the weight-stationary persistent kernel under the call pipeline at every batch size
        engine.set_option('persistent_decoder', 1)
        dflt = run(1)
        for a, b in zip(dflt, pip[2]):
            for k in a:
                assert np.array_equal(a[k], b[k]), ('default', k)
    finally:
        engine.set_option('pipeline', 1)
        engine.set_option('persistent_decoder', 1)


def test_encoder_ahead_equals_encoder_in_front(engine):
    """Round 4: under the call pipeline with the persistent decoder the encoder of a call runs on a stream of its own, one
    inter-Griffin-Lim gap ahead of its decoder (`memory` double-buffered by call parity, the host waiting for the gap).  The
    option enc_stream = 0 restores the round-3 order (encoder in front of its decoder on the front stream): same bits, for
    calls of changing content, with an unpipelined call of another shape and a stand-alone encoder call in between."""
    batches = [bench_ids(5, 21, 70 + i) for i in range(7)]
    other = bench_ids(2, 9, 99)

    def run(enc_stream):
        engine.set_option('enc_stream', enc_stream)
        # every upload before the first call: a host copy between two calls is a null-stream copy, which would serialise
        # exactly the overlaps this test is about (the stand-alone encoder call against the next pipelined encoder)
        dev = [engine.to_device(b) for b in batches]
        dev_other = engine.to_device(other)
        outs = []
        for i in range(len(batches)):
            outs.append(engine.synthesize(dev[i], 6, 6.02, 99.89, 1.3, 4, WIN, HOP, seed=500 + i,
                                          want_mel=True, want_alignments=True, want_linear=True))
            if i == 3:   # a call of another shape (unpipelined: it sizes nothing new the second time round, but breaks the rhythm)
                outs.append(engine.synthesize(dev_other, 6, 6.02, 99.89, 1.3, 4, WIN, HOP, seed=9, want_mel=True))
            if i == 4:   # the encoder's scratch is shared with stand-alone calls (ordered by an event since round 5)
                engine.encoder_forward(dev_other)
        engine.synchronize()
        return [{k: v.to_host() for k, v in o.items() if v is not None} for o in outs]

    try:
        engine.set_option('persistent_decoder', 2)   # (the encoder only runs ahead beside the persistent decoder)
        run(1)                                        # shapes known: the second round is pipelined from its second call on
        ahead = run(1)
        front = run(0)
        assert len(ahead) == len(front) == len(batches) + 1
        for i, (a, b) in enumerate(zip(ahead, front)):
            for k in a:
                assert np.array_equal(a[k], b[k]), (i, k)
            assert np.isfinite(a['wav']).all() and np.abs(a['wav']).max() > 0
    finally:
        engine.set_option('enc_stream', 1)
        engine.set_option('persistent_decoder', 1)


def test_decoder_form_switched_between_pipelined_calls(engine):
    """The decoder form decides whether a pipelined call's encoder runs ahead on its own stream (persistent decoder) or in
    front of its decoder on the front stream (launch-per-layer): switching the form between back-to-back pipelined calls
    of one shape must not let the two orders meet in the one encoder scratch / `memory` buffer.  Same bits as serial calls."""
    batches = [bench_ids(5, 21, 170 + i) for i in range(8)]
    forms = [0, 0, 2, 2, 0, 2, 2, 0]

    def run(pipeline):
        engine.set_option('pipeline', pipeline)
        dev = [engine.to_device(b) for b in batches]
        outs = []
        for i, d in enumerate(dev):
            engine.set_option('persistent_decoder', forms[i])
            outs.append(engine.synthesize(d, 6, 6.02, 99.89, 1.3, 4, WIN, HOP, seed=700 + i, want_mel=True, want_linear=True))
        engine.synchronize()
        return [{k: v.to_host() for k, v in o.items() if v is not None} for o in outs]

    try:
        run(1)   # shapes known
        seq = run(0)
        pip = run(1)
        # (the two decoder forms differ in the last bits, so call i is compared with call i of the same form)
        bad = [(i, forms[i], k) for i, (a, b) in enumerate(zip(seq, pip)) for k in a if not np.array_equal(a[k], b[k])]
        assert not bad, bad
    finally:
        engine.set_option('pipeline', 1)
        engine.set_option('persistent_decoder', 1)


def test_pipelined_calls_with_changing_shapes_equal_sequential(engine):
    """Shapes change between back-to-back calls, so unpipelined calls (the first of every new shape) alternate with
    pipelined ones, a stand-alone Griffin-Lim call sits in the middle, and calls of both parities use the phasor and
    mel buffers from either stream (the initial phasors of a pipelined call are written on the front stream): every
    waveform and spectrogram must equal the fully serialised run bit for bit."""
    shapes = [(4, 30, 8), (4, 30, 8), (4, 30, 8), (3, 21, 6), (3, 21, 6), (4, 30, 8), (4, 30, 8), (3, 21, 6), (3, 21, 6), (3, 21, 6)]
    batches = [bench_ids(B, Ts, 70 + i) for i, (B, Ts, S) in enumerate(shapes)]
    inits = [np.random.default_rng(90 + i).random((B, 1025, 5 * S)).astype(np.float32) for i, (B, Ts, S) in enumerate(shapes)]
    gl_mag = (np.random.default_rng(7).random((2, 1025, 30), dtype=np.float32) ** 3) * 4
    gl_init = np.random.default_rng(8).random((2, 1025, 30), dtype=np.float32)

    def run(pipeline):
        engine.set_option('pipeline', pipeline)
        dev_ids = [engine.to_device(b) for b in batches]
        dev_init = [engine.to_device(x) for x in inits]
        d_mag, d_gi = engine.to_device(gl_mag), engine.to_device(gl_init)
        outs, solo = [], None
        for i, (B, Ts, S) in enumerate(shapes):
            outs.append(engine.synthesize(dev_ids[i], S, 6.02, 99.89, 1.3, 5, WIN, HOP, init_phase=dev_init[i],
                                          want_mel=True, want_linear=True))
            if i == 5:   # between two pipelined calls
                solo, _ = engine.griffin_lim(d_mag, 4, WIN, HOP, 2048, init_phase=d_gi, want_mse=False)
        engine.synchronize()
        return [{k: v.to_host() for k, v in o.items() if v is not None} for o in outs], solo.to_host()

    try:
        for pd in (0, 2):
            engine.set_option('persistent_decoder', pd)
            seq, seq_solo = run(0)
            pip, pip_solo = run(1)
            assert np.array_equal(seq_solo, pip_solo), pd
            for i, (a, b) in enumerate(zip(seq, pip)):
                for k in a:
                    assert np.array_equal(a[k], b[k]), (pd, i, k)
    finally:
        engine.set_option('pipeline', 1)
        engine.set_option('persistent_decoder', 1)


def test_pipelining_on_an_adopted_stream(engine):
    """tts_set_stream + pipeline = 2: calls on a caller's stream (a plain hipStream_t here, what
    torch.cuda.current_stream().cuda_stream is) overlap like those on the library's own, with identical results;
    pipeline = 1 on an adopted stream stays serialised."""
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    stream = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(stream)) == 0
    batches = [bench_ids(4, 30, 60 + i) for i in range(4)]
    inits = [np.random.default_rng(10 + i).random((4, 1025, 40)).astype(np.float32) for i in range(4)]
    dev_ids = [engine.to_device(b) for b in batches]
    dev_init = [engine.to_device(x) for x in inits]

    def run():
        outs = [engine.synthesize(dev_ids[i], 8, 6.02, 99.89, 1.3, 6, WIN, HOP, init_phase=dev_init[i], want_mel=True)
                for i in range(4)]
        engine.synchronize()
        return [{k: v.to_host() for k, v in o.items() if v is not None} for o in outs]

    try:
        # (one decoder form throughout: by default a pipelined call takes the persistent kernel and a serial one the
        #  launch-per-layer decoder, which agree to rounding only)
        engine.set_option('persistent_decoder', 2)
        engine.set_option('pipeline', 0)
        ref = run()
        engine.set_stream(stream.value)
        for mode in (1, 2):
            engine.set_option('pipeline', mode)
            got = run()
            for a, b in zip(ref, got):
                for k in a:
                    assert np.array_equal(a[k], b[k]), (mode, k)
    finally:
        engine.set_stream(None)
        engine.set_option('pipeline', 1)
        engine.set_option('persistent_decoder', 1)
        hip.hipStreamDestroy(stream)


@pytest.mark.parametrize('B,Ts,steps,n_iter', [(6, 25, 40, 7), (64, 150, 200, 60), (1, 30, 50, 10)])
def test_a_calls_bits_do_not_depend_on_what_ran_before(engine, B, Ts, steps, n_iter):
    """The outputs of a call -- WAVEFORM included -- are the same bits whether the call ran pipelined behind other calls (its
    Griffin-Lim beside the next call's decoder on all but `reserve_cus` compute units, its last launches cut for the whole
    chip) or alone in a drained pipeline (one cut for the whole chip): the cut of the frames into runs does not reach the
    bits (tests/test_gpu_audio.py::test_griffin_lim_bits_do_not_depend_on_the_cut), every decoder path runs one kernel form's
    arithmetic.  At the bench's full size too (its pipelined calls take both cuts)."""
    batches = [bench_ids(B, Ts, 300 + i) for i in range(3)]

    def run(pipeline):
        engine.set_option('pipeline', pipeline)
        dev = [engine.to_device(b) for b in batches]
        outs = [engine.synthesize(d, steps, 6.02, 99.89, 1.3, n_iter, WIN, HOP, seed=40 + i, want_mel=True, want_linear=True, want_alignments=True)
                for i, d in enumerate(dev)]
        engine.synchronize()
        res = [{k: v.to_host() for k, v in o.items() if v is not None} for o in outs]
        for o in outs:
            for v in o.values():
                if v is not None:
                    v.free()
        for d in dev:
            d.free()
        return res

    try:
        run(1)   # shapes known
        seq = run(0)
        pip = run(1)
        for i, (a, b) in enumerate(zip(seq, pip)):
            for k in a:
                assert np.array_equal(a[k], b[k]), (i, k)
        assert np.isfinite(seq[0]['wav']).all() and np.abs(seq[0]['wav']).max() > 0
    finally:
        engine.set_option('pipeline', 1)
