"""GPU parity: Griffin-Lim / de-normalisation / peak-normalise kernels vs the numpy oracle.

Tolerances: one iteration from identical phases: rel-L2 <= 1e-4 on the
waveform; after n iterations the oracle's own convergence measures (mse, spectral
convergence) within 1 %, not sample equality.
"""
import numpy as np
import pytest

from conftest import pkg, rel_l2
from oracle import audio_oracle as A

pytestmark = pytest.mark.gpu

N_FFT, WIN, HOP = 2048, 1102, 275


def synth_mag(rng, B, T):
    """magnitude spectrograms of band-limited noise + tones, (B, 1025, T) float32."""
    out = []
    for b in range(B):
        n = HOP * (T - 1)
        t = np.arange(n) / 22050.0
        y = 0.3 * np.sin(2 * np.pi * (220 + 40 * b) * t) + 0.1 * rng.standard_normal(n)
        out.append(np.abs(A.stft(y.astype(np.float32), N_FFT, HOP, WIN)).astype(np.float32))
    return np.stack(out)


@pytest.mark.parametrize('B,T,n_iter', [(1, 12, 0), (2, 12, 1), (2, 40, 1), (1, 70, 3)])
def test_griffin_lim_few_iterations(engine, B, T, n_iter):
    rng = np.random.default_rng(10 * T + n_iter)
    mag = synth_mag(rng, B, T)
    assert mag.shape == (B, 1025, T)
    init = rng.random(mag.shape).astype(np.float32)
    wav, mse = engine.griffin_lim(mag, n_iter, WIN, HOP, N_FFT, init_phase=init)
    wav, mse = wav.to_host(), mse.to_host()
    for b in range(B):
        ref_wav, ref_mse = A.griffin_lim_v2(mag[b], WIN, HOP, N_FFT, n_iter, init_phase=init[b])
        e = rel_l2(wav[b], ref_wav)
        print('GL B={} T={} it={} b={}: wav rel-L2 {:.3e} mse {} vs {}'.format(B, T, n_iter, b, e, mse[b], ref_mse))
        assert wav[b].shape == ref_wav.shape
        assert e < 1e-4 * max(1, n_iter)   # one iteration from identical phases <= 1e-4
        if n_iter > 0:
            assert abs(mse[b] - ref_mse) <= 1e-3 * abs(ref_mse) + 1e-9


def test_griffin_lim_convergence_matches_oracle(engine):
    rng = np.random.default_rng(3)
    B, T, n_iter = 2, 48, 30
    mag = synth_mag(rng, B, T)
    init = rng.random(mag.shape).astype(np.float32)
    wav, mse = engine.griffin_lim(mag, n_iter, WIN, HOP, N_FFT, init_phase=init)
    wav, mse = wav.to_host(), mse.to_host()
    for b in range(B):
        ref_wav, ref_mse = A.griffin_lim_v2(mag[b], WIN, HOP, N_FFT, n_iter, init_phase=init[b])
        sc = lambda w: np.linalg.norm(np.abs(A.stft(w, N_FFT, HOP, WIN)) - mag[b]) / np.linalg.norm(mag[b])
        print('GL 30 it b={}: mse {} vs {}, spectral convergence {} vs {}'.format(b, mse[b], ref_mse, sc(wav[b]),
                                                                                    sc(ref_wav)))
        assert abs(mse[b] - ref_mse) <= 0.01 * abs(ref_mse)
        assert abs(sc(wav[b]) - sc(ref_wav)) <= 0.01 * sc(ref_wav)


def test_griffin_lim_seeded_is_deterministic(engine):
    rng = np.random.default_rng(4)
    mag = engine.to_device(synth_mag(rng, 2, 20))
    w0, _ = engine.griffin_lim(mag, 2, WIN, HOP, N_FFT, seed=1234)
    w1, _ = engine.griffin_lim(mag, 2, WIN, HOP, N_FFT, seed=1234)
    w2, _ = engine.griffin_lim(mag, 2, WIN, HOP, N_FFT, seed=99)
    assert np.array_equal(w0.to_host(), w1.to_host())
    assert not np.array_equal(w0.to_host(), w2.to_host())
    assert np.isfinite(w0.to_host()).all()


def test_griffin_lim_work_counter_ring_wraps(engine):
    """The work counters of the launches are slots of a 256-entry ring that is zeroed once; every launch zeroes the slot of the
    launch before it (csrc/api_stages.hip, gl_run).  More launches than slots on one handle: the result of a call does not depend
    on where in the ring it falls (a slot that was not cleared would hand a launch no work: frames left unwritten)."""
    rng = np.random.default_rng(14)
    mag = engine.to_device(synth_mag(rng, 2, 24))
    first, first_mse = engine.griffin_lim(mag, 4, WIN, HOP, N_FFT, seed=7)   # 3 + 1 iterations and the final iSTFT: 3 launches
    first, first_mse = first.to_host(), first_mse.to_host()
    for k in range(100):                                                     # 300 launches: the ring wraps
        w, m = engine.griffin_lim(mag, 4, WIN, HOP, N_FFT, seed=7)
        if k % 20 == 19:
            assert np.array_equal(w.to_host(), first) and np.array_equal(m.to_host(), first_mse), k
    # another launch form in between (one iteration per launch, no mse) and back
    w1, _ = engine.griffin_lim(mag, 1, WIN, HOP, N_FFT, seed=7, want_mse=False)
    assert np.isfinite(w1.to_host()).all()
    w, m = engine.griffin_lim(mag, 4, WIN, HOP, N_FFT, seed=7)
    assert np.array_equal(w.to_host(), first) and np.array_equal(m.to_host(), first_mse)


def test_griffin_lim_zero_bins(engine):
    # zero magnitude everywhere -> zero signal; angle(0) = 0 must not produce NaNs
    mag = np.zeros((1, 1025, 12), np.float32)
    wav, mse = engine.griffin_lim(mag, 2, WIN, HOP, N_FFT, seed=5)
    assert np.array_equal(wav.to_host(), np.zeros((1, HOP * 11), np.float32))
    assert mse.to_host()[0] == 0.0


def test_denorm_power(engine):
    rng = np.random.default_rng(6)
    lin = (rng.random((2, 9, 1025)) * 1.4 - 0.2).astype(np.float32)   # exercises both clip sides
    mag = engine.denorm_power(lin, 6.02, 99.89, 1.3).to_host()
    for b in range(2):
        ref = A.linear_to_magnitude(lin[b], 6.02, 99.89, 1.3)
        assert mag[b].shape == ref.shape == (1025, 9)
        assert rel_l2(mag[b], ref) < 1e-5
        assert np.allclose(mag[b], ref, rtol=2e-5, atol=0)


def test_denorm_db_assertion(engine):
    lin = np.zeros((1, 4, 1025), np.float32)
    with pytest.raises(AssertionError):
        engine.denorm_power(lin, -50.0, 99.89, 1.3)


def test_peak_normalize(engine):
    rng = np.random.default_rng(8)
    wav = (rng.standard_normal((3, 5000)) * 0.01).astype(np.float32)
    wav[2] = 0.0
    d = engine.to_device(wav)
    got = engine.peak_normalize(d).to_host()
    for b in range(3):
        assert np.array_equal(got[b], A.peak_normalize(wav[b]))


@pytest.mark.parametrize('per_launch', [1, 2, 3])
@pytest.mark.parametrize('B,T,n_iter,want_mse', [(2, 40, 6, False), (1, 70, 7, True), (3, 151, 5, True), (2, 9, 4, False)])
def test_griffin_lim_iterations_per_launch(engine, per_launch, B, T, n_iter, want_mse):
    """gl_stream_kernel runs 1, 2 or 3 iterations per launch (the spectrum goes from one iteration to the next in
    registers, normalised to |S| e^{i phi} without the 32-bit phasor code in between): every split of n_iter into
    launches -- with the mse the last iteration is always a launch of its own -- against the oracle.  Runs shorter
    than the stages' lead (T = 9), runs that wrap the rings several times (T = 151), both utterance ends in one run."""
    rng = np.random.default_rng(1000 * T + n_iter)
    mag = synth_mag(rng, B, T)
    init = rng.random(mag.shape).astype(np.float32)
    engine.set_option('gl_pair', per_launch)
    try:
        wav, mse = engine.griffin_lim(mag, n_iter, WIN, HOP, N_FFT, init_phase=init, want_mse=want_mse)
        wav = wav.to_host()
        mse = mse.to_host() if want_mse else None
    finally:
        engine.set_option('gl_pair', 3)
    for b in range(B):
        ref_wav, ref_mse = A.griffin_lim_v2(mag[b], WIN, HOP, N_FFT, n_iter, init_phase=init[b])
        e = rel_l2(wav[b], ref_wav)
        print('GL {} per launch, B={} T={} it={} b={}: wav rel-L2 {:.3e}'.format(per_launch, B, T, n_iter, b, e))
        assert e < 1e-4 * n_iter
        if want_mse:
            assert abs(mse[b] - ref_mse) <= 1e-3 * abs(ref_mse) + 1e-9


def seed_u(seed, B, F, T):
    """numpy restatement of gl_seed_phasor's angle (csrc/griffin_lim.hip): u = 24 bits of lowbias32(index ^ seed mix),
    index = (b F + f) T + t in the reference's (B, F, T) layout"""
    idx = np.arange(B * F * T, dtype=np.uint64)
    lo, hi = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32)
    s_lo, s_hi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over='ignore'):
        x = lo ^ (hi * np.uint32(0x9E3779B9)) ^ s_lo ^ np.uint32((int(s_hi) * 0x85EBCA6B) & 0xFFFFFFFF)
        x ^= x >> np.uint32(16); x *= np.uint32(0x7FEB352D); x ^= x >> np.uint32(15); x *= np.uint32(0x846CA68B); x ^= x >> np.uint32(16)
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).reshape(B, F, T)


@pytest.mark.parametrize('per_launch,n_iter,want_mse', [(1, 1, False), (2, 2, False), (3, 3, False), (3, 7, False), (3, 4, True), (3, 0, False)])
@pytest.mark.parametrize('seed', [0, 5, (7 << 32) + 12345])
def test_griffin_lim_seeded_start(engine, per_launch, n_iter, want_mse, seed):
    """With no initial-phase array the first launch of the iteration draws every bin's phasor from the seed itself (the
    SEEDED instantiations of gl_stream_kernel for 1, 2 and 3 iterations per launch and the mse form; with no iteration
    the codes come from phase_init_kernel): the same waveform, to the v_sin / v_cos error, as an explicit array holding
    the numpy restatement of those draws, and as the oracle started from that array."""
    B, T = 3, 41
    rng = np.random.default_rng(77)
    mag = synth_mag(rng, B, T)
    u = seed_u(seed, B, mag.shape[1], T)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02
    engine.set_option('gl_pair', per_launch)
    try:
        w_seed, m_seed = engine.griffin_lim(mag, n_iter, WIN, HOP, N_FFT, seed=seed, want_mse=want_mse)
        w_expl, m_expl = engine.griffin_lim(mag, n_iter, WIN, HOP, N_FFT, init_phase=u, want_mse=want_mse)
        w_again, _ = engine.griffin_lim(mag, n_iter, WIN, HOP, N_FFT, seed=seed, want_mse=want_mse)
    finally:
        engine.set_option('gl_pair', 3)
    w_seed, w_expl = w_seed.to_host(), w_expl.to_host()
    assert np.array_equal(w_seed, w_again.to_host())
    # the two starts differ by the v_sin / v_cos error of the in-kernel draw (~1e-6 per phasor) against sincospif + phasor code;
    # an iteration carries such a difference on and can grow it (7 iterations: up to 2.6e-4 on one of the seeds)
    tol = 4e-5 * max(1, n_iter)
    assert rel_l2(w_seed, w_expl) < tol, rel_l2(w_seed, w_expl)
    for b in range(B):
        ref_wav, ref_mse = A.griffin_lim_v2(mag[b], WIN, HOP, N_FFT, n_iter, init_phase=u[b])
        assert rel_l2(w_seed[b], ref_wav) < 1e-4 * max(1, n_iter)
        if want_mse:
            assert abs(m_seed.to_host()[b] - ref_mse) <= 1e-3 * abs(ref_mse) + 1e-9


@pytest.mark.parametrize('run_len', [8, 16, 40, 104, 296])
@pytest.mark.parametrize('per_launch,n_iter,want_mse', [(1, 2, True), (3, 4, False)])
def test_griffin_lim_forced_run_cuts(engine, run_len, per_launch, n_iter, want_mse):
    """Every cut of the utterances into runs gives the same waveform (the same BITS even:
    test_griffin_lim_bits_do_not_depend_on_the_cut; here against the oracle): forced run lengths from one round of the waves to longer than the utterance -- many runs per
    workgroup, runs too short to draw the next item late (fewer than four rounds: drawn at the start) next to runs that draw
    it 24 indices before their end, remainders of one class."""
    import os
    B, T = 5, 151
    rng = np.random.default_rng(run_len)
    mag = synth_mag(rng, B, T)
    init = rng.random(mag.shape).astype(np.float32)
    engine.set_option('gl_pair', per_launch)
    engine.set_option('debug_hooks', 1)
    engine.set_option('gl_run_len', run_len)
    try:
        wav, mse = engine.griffin_lim(mag, n_iter, WIN, HOP, N_FFT, init_phase=init, want_mse=want_mse)
        wav = wav.to_host()
        mse = mse.to_host() if want_mse else None
    finally:
        engine.set_option('gl_run_len', 0)
        engine.set_option('debug_hooks', 0)
        engine.set_option('gl_pair', 3)
    for b in range(B):
        ref_wav, ref_mse = A.griffin_lim_v2(mag[b], WIN, HOP, N_FFT, n_iter, init_phase=init[b])
        assert rel_l2(wav[b], ref_wav) < 1e-4 * n_iter
        if want_mse:
            assert abs(mse[b] - ref_mse) <= 1e-3 * abs(ref_mse) + 1e-9


def test_griffin_lim_bits_do_not_depend_on_the_cut(engine):
    """The waveform is the same BITS whatever runs the utterances are cut into and however many workgroups draw them: every
    sample is summed over the frames that cover it in ascending order, in whatever run they lie, and a run's halo recomputes
    its neighbours' frames exactly.  (Until round 6 the cut was believed to be part of the bits; it is what lets a pipelined
    call use one cut beside the decoder and another on the whole chip without its result depending on either.)"""
    B, T = 8, 1000
    rng = np.random.default_rng(77)
    mag = engine.to_device((rng.random((B, 1025, T), dtype=np.float32) ** 4) * 10)
    init = engine.to_device(rng.random((B, 1025, T), dtype=np.float32))
    engine.set_option('debug_hooks', 1)
    outs = {}
    try:
        for runs, rl, workers in ((0, 0, 0), (3, 0, 0), (5, 0, 0), (0, 296, 0), (0, 104, 0), (0, 56, 0), (0, 0, 224), (0, 0, 100), (0, 0, 17)):
            engine.set_option('gl_runs', runs)
            engine.set_option('gl_run_len', rl)
            engine.set_option('gl_workers', workers)
            wav, _ = engine.griffin_lim(mag, 12, WIN, HOP, N_FFT, init_phase=init, want_mse=False)
            engine.synchronize()
            outs[(runs, rl, workers)] = wav.to_host().copy()
            wav.free()
    finally:
        for k in ('gl_runs', 'gl_run_len', 'gl_workers', 'debug_hooks'):
            engine.set_option(k, 0)
        mag.free(); init.free()
    ref = outs[(0, 0, 0)]
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0
    for k, v in outs.items():
        assert np.array_equal(ref.view(np.uint32), v.view(np.uint32)), k


def test_griffin_lim_utterances_cut_into_different_numbers_of_runs(engine, weights):
    """The frames are dealt to the workgroups across utterance ends (gl_plan_items), so utterances are not all cut into the same
    number of runs (T = 120, B = 7 on 17 workgroups: six utterances in 3 runs, one in 4): the per-run partial results -- the
    squared error of the last iteration, the peak of the final iSTFT -- have one slot per run, and the last run of an utterance
    zeroes the slots it does not have.  The mse against the oracle and the default cut; the peak-normalised waveform of
    tts_synthesize bit for bit against the default cut."""
    B, T, n_iter = 7, 120, 4
    rng = np.random.default_rng(120)
    mag = synth_mag(rng, B, T)
    init = rng.random(mag.shape).astype(np.float32)
    ids = np.zeros((B, 20), np.int32)
    for b in range(B):
        ids[b, :12 + b] = rng.integers(2, 39, 12 + b)
        ids[b, 12 + b] = 1
    res = {}
    engine.set_option('debug_hooks', 1)
    engine.set_option('pipeline', 0)
    try:
        for workers in (0, 17):
            engine.set_option('gl_workers', workers)
            wav, mse = engine.griffin_lim(mag, n_iter, WIN, HOP, N_FFT, init_phase=init, want_mse=True)
            out = engine.synthesize(ids, T // 5, 6.02, 99.89, 1.3, n_iter, WIN, HOP, seed=9, peak_normalize=True)
            engine.synchronize()
            res[workers] = (wav.to_host().copy(), mse.to_host().copy(), out['wav'].to_host().copy())
    finally:
        engine.set_option('gl_workers', 0)
        engine.set_option('debug_hooks', 0)
        engine.set_option('pipeline', 1)
    wav0, mse0, syn0 = res[0]
    wav1, mse1, syn1 = res[17]
    assert np.array_equal(wav0.view(np.uint32), wav1.view(np.uint32))
    assert np.allclose(mse0, mse1, rtol=1e-5, atol=0)
    assert np.array_equal(syn0.view(np.uint32), syn1.view(np.uint32))
    assert np.allclose(np.abs(syn1).max(axis=1), 1.0, atol=1e-6)
    for b in (0, B - 1):
        _, ref_mse = A.griffin_lim_v2(mag[b], WIN, HOP, N_FFT, n_iter, init_phase=init[b])
        assert abs(mse1[b] - ref_mse) <= 1e-3 * abs(ref_mse) + 1e-9


# ---- every power-of-two n_fft / window / hop on the audio surface (csrc/griffin_lim_generic.hip): the reference passes
# n_fft, win_length and hop_length as arguments (audio/synthesis.py:5-40, 43-125; audio/features.py:5-86, 116-145)
@pytest.mark.parametrize('n_fft,win,hop,B,T', [
    (1024, 800, 200, 3, 60),
    (4096, 2400, 600, 2, 40),
    (512, 512, 128, 2, 50),        # win == n_fft, hop = win / 4 (librosa's defaults)
    (256, 200, 50, 1, 45),
    (2048, 1200, 300, 2, 40),      # the model's n_fft with another window / hop: the general kernels as well
    (2048, 800, 200, 3, 60),       # 50 / 12.5 ms at 16 kHz: the streaming kernel's second instantiation (round 6)
    (2048, 800, 200, 1, 260),      # ... several laps of its LDS ring
    (2048, 2048, 512, 1, 30),
])
def test_griffin_lim_other_sizes_one_iteration(engine, n_fft, win, hop, B, T):
    """One iteration and the final iSTFT from identical phases against the oracle, sample by sample (<= 1e-4 of the peak),
    with the mse of the iteration."""
    rng = np.random.default_rng(n_fft + win)
    F = 1 + n_fft // 2
    mag = ((rng.random((B, F, T)) ** 4) * 10).astype(np.float32)
    init = rng.random((B, F, T)).astype(np.float32)
    wav, mse = engine.griffin_lim(mag, 1, win, hop, n_fft, init_phase=init, want_mse=True)
    wav, mse = wav.to_host(), mse.to_host()
    assert wav.shape == (B, hop * (T - 1))
    for b in range(B):
        ref_wav, ref_mse = A.griffin_lim_v2(mag[b], win, hop, n_fft, 1, init_phase=init[b])
        assert np.abs(wav[b] - ref_wav).max() <= 1e-4 * max(1e-6, np.abs(ref_wav).max()), (n_fft, b)
        assert rel_l2(wav[b], ref_wav) < 2e-5
        assert abs(mse[b] - ref_mse) <= 1e-4 * ref_mse
    # no iteration at all: the iSTFT of the initial estimate
    wav0, _ = engine.griffin_lim(mag, 0, win, hop, n_fft, init_phase=init, want_mse=False)
    ref0, _ = A.griffin_lim_v2(mag[0], win, hop, n_fft, 0, init_phase=init[0])
    assert rel_l2(wav0.to_host()[0], ref0) < 2e-5


@pytest.mark.parametrize('n_fft,win,hop', [(1024, 800, 200), (4096, 2400, 600), (2048, 800, 200)])
def test_griffin_lim_other_sizes_30_iterations(engine, n_fft, win, hop):
    """30 iterations: the mse (the reference's convergence measure, audio/synthesis.py:115) within 1 % of the oracle's and the
    same spectral convergence of the result; a seeded start is reproducible."""
    B, T, n_iter = 2, 50, 30
    rng = np.random.default_rng(n_fft)
    F = 1 + n_fft // 2
    mag = ((rng.random((B, F, T)) ** 4) * 10).astype(np.float32)
    init = rng.random((B, F, T)).astype(np.float32)
    wav, mse = engine.griffin_lim(mag, n_iter, win, hop, n_fft, init_phase=init, want_mse=True)
    wav, mse = wav.to_host(), mse.to_host()
    for b in range(B):
        ref_wav, ref_mse = A.griffin_lim_v2(mag[b], win, hop, n_fft, n_iter, init_phase=init[b])
        assert abs(mse[b] - ref_mse) <= 0.01 * ref_mse, (mse[b], ref_mse)
        sc = lambda y: np.linalg.norm(np.abs(A.stft(y, n_fft, hop, win)) - mag[b]) / np.linalg.norm(mag[b])   # noqa: E731
        assert abs(sc(wav[b]) - sc(ref_wav)) <= 0.01 * sc(ref_wav)
    w1, _ = engine.griffin_lim(mag, 3, win, hop, n_fft, seed=77, want_mse=False)
    w2, _ = engine.griffin_lim(mag, 3, win, hop, n_fft, seed=77, want_mse=False)
    w3, _ = engine.griffin_lim(mag, 3, win, hop, n_fft, seed=78, want_mse=False)
    assert np.array_equal(w1.to_host(), w2.to_host()) and not np.array_equal(w1.to_host(), w3.to_host())


@pytest.mark.parametrize('n_fft,win,hop', [(1024, 800, 200), (4096, 2400, 600), (512, 400, 100)])
def test_stft_and_mel_other_sizes(engine, n_fft, win, hop):
    """tts_stft / tts_stft_magnitude / tts_mel_spectrogram at other transform sizes against the oracle (librosa.stft,
    HTK mel basis with Slaney normalisation: audio/features.py:5-86, 116-145)."""
    rng = np.random.default_rng(n_fft)
    n = hop * 37
    y = (0.3 * np.sin(2 * np.pi * 330 * np.arange(n) / 22050) + 0.05 * rng.standard_normal(n)).astype(np.float32)
    Fm = pkg('audio.features')
    S = Fm.linear_scale_spectrogram(y, n_fft, hop, win, engine=engine)
    ref = A.stft(y, n_fft, hop, win)
    assert S.dtype == np.complex64 and S.shape == ref.shape == (1 + n_fft // 2, 38)
    assert np.linalg.norm(S - ref) / np.linalg.norm(ref) < 1e-5
    mel = Fm.mel_scale_spectrogram(y, n_fft, 22050, 80, 0, 8000, hop, win, 1.0, engine=engine)
    rmel, _ = A.mel_scale_spectrogram(y, n_fft, 22050, 80, 0, 8000, hop, win, 1.0)
    assert mel.shape == rmel.shape == (80, 38) and rel_l2(mel, rmel) < 1e-5
    p2 = engine.stft_magnitude(y[None], n_fft, win, hop, 2.0).to_host()[0]
    assert rel_l2(p2, np.abs(ref) ** 2) < 1e-5


@pytest.mark.parametrize('per_launch', [1, 2, 3])
def test_streaming_kernel_second_window_iterations_per_launch(engine, per_launch):
    """The streaming kernel's 800 / 200 instantiation (n_fft 2048: the reference's 50 ms / 12.5 ms at 16 kHz) with one, two and
    three iterations per launch, runs cut by the planner for a batch that does not fit one run per workgroup: 7 iterations
    against the oracle through the mse and sample-wise within the bound of test_griffin_lim_few_iterations."""
    B, T, n_iter, win, hop = 3, 150, 7, 800, 200
    rng = np.random.default_rng(800 + per_launch)
    mag = ((rng.random((B, 1025, T)) ** 4) * 10).astype(np.float32)
    init = rng.random((B, 1025, T)).astype(np.float32)
    try:
        engine.set_option('gl_pair', per_launch)
        wav, mse = engine.griffin_lim(mag, n_iter, win, hop, N_FFT, init_phase=init, want_mse=True)
    finally:
        engine.set_option('gl_pair', 3)
    wav, mse = wav.to_host(), mse.to_host()
    for b in range(B):
        ref_wav, ref_mse = A.griffin_lim_v2(mag[b], win, hop, N_FFT, n_iter, init_phase=init[b])
        assert rel_l2(wav[b], ref_wav) < 1e-4 * n_iter, (per_launch, b)
        assert abs(mse[b] - ref_mse) <= 1e-3 * ref_mse
