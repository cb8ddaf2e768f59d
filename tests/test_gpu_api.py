"""GPU: the reference-shaped Python surface (tacotron.model / tacotron.inference / audio.*), the
analysis features, golden fixtures and C-ABI behaviour, all through libsstts_hip.so."""
import os

import numpy as np
import pytest

from conftest import pkg, rel_l2
from oracle import audio_oracle as A

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
N_FFT, WIN, HOP = 2048, 1102, 275


def test_manifest_matches_python(engine, hparams):
    W = pkg('tacotron.weights')
    m = W.manifest(hparams)
    assert engine.manifest() == [(k, tuple(v)) for k, v in m.items()]


def test_golden_network_fixture(engine):
    g = np.load(os.path.join(GOLD, 'network_small.npz'))
    S = int(g['n_steps'])
    mem = engine.encoder_forward(g['ids'])
    assert rel_l2(mem.to_host(), g['memory']) < 1e-3
    mel, al = engine.decoder_forward(mem, S)
    assert rel_l2(mel.to_host(), g['reduced_mel']) < 1e-3
    assert np.abs(al.to_host() - g['alignments']).max() < 1e-4
    lin = engine.postnet_forward(mel.to_host().reshape(2, -1, 80))
    assert rel_l2(lin.to_host(), g['linear']) < 1e-3


def test_golden_griffin_lim_fixture(engine):
    g = np.load(os.path.join(GOLD, 'griffin_lim_small.npz'))
    wav, mse = engine.griffin_lim(g['mag'][None], int(g['n_iter']), WIN, HOP, N_FFT, init_phase=g['init_phase'][None])
    assert rel_l2(wav.to_host()[0], g['wav']) < 1e-4
    assert abs(mse.to_host()[0] - g['mse']) < 1e-3 * g['mse']
    for n, key in ((0, 'wav_n0'), (1, 'wav_n1')):
        w, _ = engine.griffin_lim(g['mag'][None], n, WIN, HOP, N_FFT, init_phase=g['init_phase'][None])
        assert rel_l2(w.to_host()[0], g[key]) < 1e-4, key
    mag = engine.denorm_power(g['linear'][None], 6.02, 99.89, 1.3).to_host()[0]
    assert rel_l2(mag, g['linear_mag']) < 1e-5


def test_weights_required(hparams):
    sstts = pkg()
    eng = sstts.Engine(hparams)
    with pytest.raises(sstts.TtsError) as e:
        eng.encoder_forward(np.zeros((1, 4), np.int32))
    assert e.value.code == -2
    with pytest.raises(sstts.TtsError):
        eng.load_weights({})
    eng.close()


def test_unsupported_configs_are_rejected(hparams):
    import copy
    sstts = pkg()
    hp = copy.deepcopy(hparams)
    hp.encoder.n_gru_units = 64
    with pytest.raises(sstts.TtsError) as e:
        sstts.Engine(hp)
    assert e.value.code == -5
    eng = sstts.Engine(hparams)
    with pytest.raises(sstts.TtsError) as e:   # n_fft must be a power of two between 256 and 4096
        eng.griffin_lim(np.ones((1, 501, 20), np.float32), 1, 400, 100, 1000)
    assert e.value.code == -5
    with pytest.raises(sstts.TtsError) as e:
        eng.griffin_lim(np.ones((1, 4097, 20), np.float32), 1, 400, 100, 8192)
    assert e.value.code == -5
    eng.close()


def test_device_info(engine):
    """tts_device_info: 32 hex digits of the device's UUID and its compute-unit count (what bench.py compares across ranks)."""
    uuid, cus = engine.device_info()
    assert len(uuid) == 32 and all(c in '0123456789abcdef' for c in uuid) and cus >= 64
    assert engine.device_info() == (uuid, cus)


def test_stft_and_mel_features(engine):
    rng = np.random.default_rng(0)
    F = pkg('audio.features')
    y = (0.3 * np.sin(2 * np.pi * 440 * np.arange(HOP * 30) / 22050) + 0.05 * rng.standard_normal(HOP * 30)).astype(np.float32)
    S = F.linear_scale_spectrogram(y, N_FFT, HOP, WIN, engine=engine)
    ref = A.stft(y, N_FFT, HOP, WIN)
    assert S.dtype == np.complex64 and S.shape == ref.shape == (1025, 31)
    assert np.linalg.norm(S - ref) / np.linalg.norm(ref) < 1e-5
    mel = F.mel_scale_spectrogram(y, N_FFT, 22050, 80, 0, 8000, HOP, WIN, 1.0, engine=engine)
    rmel, rlin = A.mel_scale_spectrogram(y, N_FFT, 22050, 80, 0, 8000, HOP, WIN, 1.0)
    assert mel.shape == rmel.shape == (80, 31)
    assert rel_l2(mel, rmel) < 1e-5
    lin2 = engine.stft_magnitude(y[None], N_FFT, WIN, HOP, 2.0).to_host()[0]
    assert rel_l2(lin2, np.abs(ref) ** 2) < 1e-5
    # default hop = win // 4, default win = n_fft (librosa defaults kept by the reference wrapper)
    S2 = F.linear_scale_spectrogram(y, N_FFT, engine=engine)
    assert S2.shape == (1025, 1 + len(y) // 512)
    assert np.linalg.norm(S2 - A.stft(y, N_FFT, 512, N_FFT)) / np.linalg.norm(S2) < 1e-5


def test_conversion_module(engine):
    C = pkg('audio.conversion')
    rng = np.random.default_rng(1)
    x = rng.random((7, 5)).astype(np.float32) * 3
    assert np.allclose(C.magnitude_to_decibel(x, engine=engine), A.magnitude_to_decibel(x), rtol=1e-5, atol=1e-4)
    assert C.magnitude_to_decibel(np.zeros(3, np.float32), engine=engine).tolist() == [-100.0] * 3
    db = (rng.random((4, 6)).astype(np.float32) - 0.7) * 100
    assert np.allclose(C.decibel_to_magnitude(db, engine=engine), A.decibel_to_magnitude(db), rtol=2e-5)
    with pytest.raises(AssertionError):
        C.decibel_to_magnitude(np.array([0.0, -100.5], np.float32), engine=engine)
    n = rng.random(11).astype(np.float32) * 1.4 - 0.2
    assert np.allclose(C.inv_normalize_decibel(n, 6.02, 99.89, engine=engine), A.inv_normalize_decibel(n, 6.02, 99.89), atol=1e-4)
    assert np.allclose(C.normalize_decibel(db, 6.02, 99.89, engine=engine), A.normalize_decibel(db, 6.02, 99.89), atol=1e-6)
    assert C.ms_to_samples(50.0, 22050) == 1102 and C.ms_to_samples(12.5, 22050) == 275
    assert C.samples_to_ms(22050, 22050) == 1000


def test_synthesis_module_single_and_batched(engine):
    S = pkg('audio.synthesis')
    g = np.load(os.path.join(GOLD, 'griffin_lim_small.npz'))
    wav = S.spectrogram_to_wav(g['mag'], WIN, HOP, N_FFT, 2, init_phase=g['init_phase'], engine=engine)
    assert wav.dtype == np.float32 and wav.shape == g['wav'].shape
    assert rel_l2(wav, g['wav']) < 1e-4
    sig, mse = S.griffin_lim_v2(np.stack([g['mag'], g['mag']]), WIN, HOP, N_FFT, 2,
                                init_phase=np.stack([g['init_phase']] * 2), engine=engine)
    assert sig.shape == (2,) + g['wav'].shape and np.array_equal(sig[0], sig[1])
    unseeded = S.spectrogram_to_wav(g['mag'], WIN, HOP, N_FFT, 1, engine=engine)
    assert np.isfinite(unseeded).all()


def test_save_wav_float32_peak_normalised(engine, tmp_path):
    io = pkg('audio.io')
    wav = (np.random.default_rng(2).standard_normal(3000) * 0.1).astype(np.float32)
    p = str(tmp_path / '1.wav')
    io.save_wav(p, wav, 22050, True, engine=engine)
    from scipy.io import wavfile
    sr, data = wavfile.read(p)
    assert sr == 22050 and data.dtype == np.float32
    assert np.array_equal(data, A.peak_normalize(wav))


def test_tacotron_facade_and_inference(weights, hparams, tmp_path):
    M = pkg('tacotron.model')
    I = pkg('tacotron.inference')
    P = pkg('tacotron.params')
    with pytest.raises(NotImplementedError):
        M.Tacotron(M.Tacotron.model_placeholders(), M.Mode.TRAIN)
    ph = M.Tacotron.model_placeholders()
    assert set(ph) == {'ph_sentences', 'ph_sentence_length', 'ph_mel_specs', 'ph_lin_specs', 'ph_time_frames'}
    model = M.Tacotron(inputs=ph, mode=M.Mode.PREDICT, weights=weights)
    ids = np.array([I.pad_sentence(np.array([5, 9, 1]), 6), [4, 5, 6, 7, 8, 1]], dtype=np.int32)
    assert ids[0].tolist() == [5, 9, 1, 0, 0, 0]
    lin, mel, red, al = model.run([model.output_linear_spec, model.output_mel_spec, model.reduced_output_mel_spec,
                                   model.alignment_history], {model.inp_sentences: ids}, n_steps=3)
    assert lin.shape == (2, 15, 1025) and mel.shape == (2, 15, 80) and red.shape == (2, 3, 400) and al.shape == (3, 2, 6)
    assert np.array_equal(red.reshape(2, 15, 80), mel)
    P.inference_params.synthesis_dir = str(tmp_path)
    specs = I.inference(model, ids, n_steps=3)
    assert len(specs) == 2 and specs[0].shape == (1025, 15) and specs[0].dtype == np.float32
    ref = A.decibel_to_magnitude(A.inv_normalize_decibel(lin[0].T, 6.02, 99.89))
    assert rel_l2(specs[0], ref) < 1e-5
    a = np.load(tmp_path / 'alignments.npz')['alignments']
    s = np.load(tmp_path / 'linear-spectrogram.npz')['linear_spec']
    assert a.shape == (2, 6, 3) and s.shape == (1, 1025, 15, 1)           # reference model.py:552-598 layouts
    assert np.array_equal(a, np.transpose(al, (1, 2, 0))) and np.array_equal(s[0, :, :, 0], lin[0].T)
    wavs = I.synthesize_batch(model, ids, n_steps=3, n_iter=2, seed=3)
    assert wavs.shape == (2, 275 * 14) and np.isfinite(wavs).all()
    model.engine.close()


def test_synthesize_sentences_writes_numbered_wavs(weights, tmp_path):
    I = pkg('tacotron.inference')
    P = pkg('tacotron.params')
    P.model_params.decoder.maximum_iterations = 20       # 4 decoder steps keep the test small
    P.model_params.reconstruction_iterations = 2
    try:
        with pytest.raises(NotADirectoryError):
            I.synthesize_sentences(['hello'], weights, out_dir=str(tmp_path / 'missing'))
        wavs = I.synthesize_sentences(['Hello world.', 'Mr. Smith said hi!'], weights, out_dir=str(tmp_path))
    finally:
        P.model_params.decoder.maximum_iterations = 1000
        P.model_params.reconstruction_iterations = 50
    assert sorted(os.listdir(tmp_path)) == ['1.wav', '2.wav']
    from scipy.io import wavfile
    sr, d = wavfile.read(tmp_path / '2.wav')
    assert sr == 22050 and d.dtype == np.float32 and d.shape == wavs[1].shape == (275 * 19,)
    assert np.isclose(np.abs(d).max(), 1.0)


def test_serve_helpers(weights, tmp_path):
    S = pkg('tacotron.serve')
    P = pkg('tacotron.params')
    LJ = pkg('datasets.lj_speech').LJSpeechDatasetHelper
    ds = LJ('/nonexistent', dict(P.dataset_params.vocabulary_dict), False)
    ids = S.pre_process_sentences(['Hello there.', 'Hi'], ds)
    assert ids.dtype == np.int32 and ids.shape == (2, 12) and ids[1].tolist()[:3] == [9, 4, 1] and ids[1, 3:].sum() == 0
    P.model_params.decoder.maximum_iterations = 20
    P.model_params.reconstruction_iterations = 2
    try:
        gen = S.serve(iter([['Hello there.', 'Hi'], ['One more.']]), weights)
        first = next(gen)
        second = next(gen)
    finally:
        P.model_params.decoder.maximum_iterations = 1000
        P.model_params.reconstruction_iterations = 50
    assert len(first) == 2 and len(second) == 1 and first[0].shape == (275 * 19,) and np.isfinite(first[1]).all()


@pytest.mark.parametrize('form', ['grucell', 'cudnn_canonical', 'cudnn_opaque'])
def test_weights_through_the_checkpoint_importer_match_the_oracle(tmp_path, form):
    """bundle on disk -> load_checkpoint (checksums verified) -> tts_set_weight -> HIP forward pass, compared with the
    ORACLE run on the original weight dictionary (not with a second HIP run): the GRUCell layout, and force_cudnn
    checkpoints that store the CBHG bi-GRUs the way CudnnGRUSaveable does or as the raw opaque cuDNN buffer
    (reference tacotron/inference.py:44-55,71; layers.py:560-577).  Two data shards, several restarts per block."""
    import copy
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tf_bundle_writer import write_tensor_bundle
    from oracle import tacotron_oracle as O
    from conftest import rel_l2
    C = pkg('tacotron.checkpoint')
    W = pkg('tacotron.weights')
    hp = copy.deepcopy(pkg('tacotron.params').ModelParams())
    cudnn = form != 'grucell'
    hp.force_cudnn = cudnn
    weights = W.synthetic_weights(5, hp)
    U, H = hp.encoder.n_highway_units, hp.encoder.n_gru_units
    ck = {}
    for k, v in weights.items():
        cb_gru = None
        for scope in ('encoder', 'post_process'):
            for d in ('fw', 'bw'):
                pre = '{}/gru/{}/gru_cell_{}/'.format(scope, d, d)
                if cudnn and k.startswith(pre):
                    cb_gru = (scope, d, k[len(pre):])
        if cb_gru and form == 'cudnn_canonical':
            k = '{}/gru/cudnn_gru/stack_bidirectional_rnn/cell_0/bidirectional_rnn/{}/cudnn_compatible_gru_cell/{}'.format(*cb_gru)
        elif cb_gru:
            continue                                   # goes into the opaque buffer below
        ck[k] = v
    if form == 'cudnn_opaque':
        for scope in ('encoder', 'post_process'):
            ws, bs = [], []
            for d in ('fw', 'bw'):
                pre = '{}/gru/{}/gru_cell_{}/'.format(scope, d, d)
                gk, gb = weights[pre + 'gates/kernel'], weights[pre + 'gates/bias']
                wi = [gk[:U, :H].T, gk[:U, H:].T, weights[pre + 'candidate/input_projection/kernel'].T]
                wr = [gk[U:, :H].T, gk[U:, H:].T, weights[pre + 'candidate/hidden_projection/kernel'].T]
                ws += [np.concatenate([m.reshape(-1) for m in wi]), np.concatenate([m.reshape(-1) for m in wr])]
                bs += [np.concatenate([gb[:H], gb[H:], weights[pre + 'candidate/input_projection/bias']]),
                       np.concatenate([np.zeros(2 * H, np.float32), weights[pre + 'candidate/hidden_projection/bias']])]
            ck['{}/gru/cudnn_gru/opaque_kernel'.format(scope)] = np.concatenate(ws + bs).astype(np.float32)
    ck['global_step'] = np.array(215000, dtype=np.int64)
    ck['dense/kernel/Adam'] = np.zeros_like(weights['dense/kernel'])
    run = tmp_path / 'run'
    run.mkdir()
    write_tensor_bundle(str(run / 'model.ckpt-215000'), ck, block_entries=16, num_shards=2, restart_interval=4, crc_fn=C.crc32c)
    (run / 'checkpoint').write_text('model_checkpoint_path: "model.ckpt-215000"\n')
    loaded = C.load_checkpoint(str(run), hp)
    B, Ts, S = 2, 11, 6
    ids = np.random.default_rng(0).integers(2, 39, (B, Ts)).astype(np.int32)
    ids[:, -1] = 1
    w64 = {k: v.astype(np.float64) for k, v in weights.items()}
    ref_mem = O.encoder(ids, w64, hp)
    ref_mel, ref_al = O.decoder(ref_mem, w64, hp, n_steps=S)
    ref_lin = O.post_process(ref_mel.reshape(B, -1, hp.n_mels), w64, hp)
    eng = pkg().Engine(hp)
    eng.load_weights(loaded)
    mem = eng.encoder_forward(ids)
    mel, al = eng.decoder_forward(mem, S)
    lin = eng.postnet_forward(mel.to_host().reshape(B, S * hp.reduction, hp.n_mels))
    errs = dict(memory=rel_l2(mem.to_host().reshape(-1), ref_mem.reshape(-1)), mel=rel_l2(mel.to_host().reshape(-1), ref_mel.reshape(-1)),
                align=float(np.abs(al.to_host().reshape(-1) - ref_al.reshape(-1)).max()), linear=rel_l2(lin.to_host().reshape(-1), ref_lin.reshape(-1)))
    eng.close()
    print('importer ({}) vs oracle: {}'.format(form, errs))
    assert errs['memory'] < 1e-4 and errs['mel'] < 1e-3 and errs['align'] < 1e-4 and errs['linear'] < 1e-3, errs


def test_host_facade_two_calls_in_flight_equals_serial_calls(engine, hparams):
    """tacotron.inference.synthesize_stream (host ids in, host waveforms out, three batches in flight: the upload of batch
    k + 1 and the download of batch k - 1 overlap batch k) gives, bit for bit, what one serialised device call per batch
    gives -- for batches of changing content and a changing sentence length (the second shape is new: unpipelined once)."""
    Inf = pkg('tacotron.inference')
    Tm = pkg('tacotron.model')
    P = pkg('tacotron.params')
    model = Tm.Tacotron(inputs=Tm.Tacotron.model_placeholders(), mode=Tm.Mode.PREDICT, engine=engine, hparams=hparams)
    rng = np.random.default_rng(12)
    shapes = [(5, 17), (5, 17), (5, 17), (3, 9), (3, 9), (5, 17)]
    batches = []
    for B, Ts in shapes:
        ids = rng.integers(2, 39, (B, Ts)).astype(np.int32)
        ids[:, -1] = 1
        batches.append(ids)
    S, n_iter = 8, 4
    loader = P.dataset_params.dataset_loader
    args = (S, loader.mel_mag_ref_db, loader.mel_mag_max_db, hparams.magnitude_power, n_iter, 1102, 275)
    reset = np.full((2, 5), 2, np.int32)   # a call of another shape: both sequences then start unpipelined

    engine.synthesize(reset, *args, seed=1)
    got = [w.copy() for w in Inf.synthesize_stream(model, iter(batches), n_steps=S, n_iter=n_iter, seed=100, peak_normalize=True)]
    assert len(got) == len(batches)
    # ... and with the linear spectrograms and the alignments of every call travelling to host memory behind its waveforms
    # (tts_synth_params_t::host_outputs, tts_wait_host_outputs): what the reference's inference() hands back (inference.py:75-101)
    engine.synthesize(reset, *args, seed=1)
    got_x = [(w.copy(), l.copy(), a.copy()) for w, l, a in
             Inf.synthesize_stream(model, iter(batches), n_steps=S, n_iter=n_iter, seed=100, peak_normalize=True,
                                   want_linear=True, want_alignments=True)]
    assert len(got_x) == len(batches)
    # the same calls on device-resident ids, each waited for before the next is made
    engine.synthesize(reset, *args, seed=1)
    for k, ids in enumerate(batches):
        ref = engine.synthesize(engine.to_device(ids), *args, seed=100 + k, peak_normalize=True, want_linear=True, want_alignments=True)
        want = ref['wav'].to_host()
        assert got[k].shape == want.shape == (ids.shape[0], 275 * (S * hparams.reduction - 1))
        assert np.isfinite(got[k]).all() and np.abs(got[k]).max() > 0
        assert np.array_equal(got[k], want), k
        w, lin, ali = got_x[k]
        assert np.array_equal(w, want), k
        assert lin.shape == (ids.shape[0], S * hparams.reduction, 1025) and np.array_equal(lin, ref['linear'].to_host()), k
        assert ali.shape == (S, ids.shape[0], ids.shape[1]) and np.array_equal(ali, ref['alignments'].to_host()), k
    # inference_stream: per batch what the reference's inference() returns (de-normalised (1025, T) magnitudes) and the waveforms
    engine.synthesize(reset, *args, seed=1)
    for k, (specs, wavs) in enumerate(Inf.inference_stream(model, iter(batches[:2]), n_steps=S, n_iter=n_iter, seed=100)):
        one = Inf.inference(model, batches[k], n_steps=S)
        assert len(specs) == len(one) == batches[k].shape[0]
        for a, b in zip(specs, one):
            assert a.shape == b.shape == (1025, S * hparams.reduction) and np.allclose(a, b, rtol=1e-5, atol=0)
        assert wavs.shape[0] == batches[k].shape[0]
    # tickets of calls whose buffers have been handed on are refused
    sstts = pkg()
    with pytest.raises(sstts.TtsError):
        engine.wait_host(0)


def test_host_facade_keeps_the_call_pipeline():
    """Throughput guard: through host memory (tts_synthesize_host / tts_wait_host, three calls in flight) a batch takes what it
    takes in a device-resident loop -- the copy streams must not end up serialising the front and the main stream (they did
    once: streams of one priority share a few hardware queues).  Bench shape, `bench.py --through-facade` in a process of
    its own: a process that has imported torch runs the library on torch's bundled HIP runtime, where the same loop is
    about 50 % slower through host memory, and pytest's collection imports torch.  The bound is loose
    (the failure mode is a factor of two)."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '8', '--warmup', '3', '--through-facade',
                          '--no-cpu-baseline'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    rec = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith('{')][-1])
    t_dev, t_host = rec['ms_per_step'], rec['facade_ms_per_step']
    print('per batch: device-resident loop {:.2f} ms, through host memory {:.2f} ms'.format(t_dev, t_host))
    assert t_host < 1.3 * t_dev, (t_dev, t_host)
