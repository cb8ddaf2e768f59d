"""GPU: BASELINE.json's configurations at their FULL sizes against the oracle.

  config 1  one LJ-Speech sentence, text -> waveform through the tacotron.inference mirror: 200 decoder steps,
            1000 frames, 50 Griffin-Lim iterations (the reference's defaults, params/model.py:48,108)
  config 4  64 utterances x 1000 frames x 60 Griffin-Lim iterations: oracle on two rows, duplicated rows
            bit-equal, error decreasing with the iteration count
  plus a Griffin-Lim run on the one model output the reference ships
  (visualization/data/ljspeech/v1.1/post-processing/ljspeech-linear-spec-post-215k.npz, copied as data).

Griffin-Lim is compared the way SURVEY.md 8(d) prescribes: sample-wise for one iteration from identical phases,
through the reference's own `mse` (audio/synthesis.py:112) and the spectral convergence after many iterations."""
import os

import numpy as np
import pytest

from conftest import pkg, rel_l2
from oracle import audio_oracle as A
from oracle import tacotron_oracle as O

pytestmark = pytest.mark.gpu
N_FFT, WIN, HOP, SR = 2048, 1102, 275, 22050
REF_DB, MAX_DB, POWER = 6.02, 99.89, 1.3
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def spectral_convergence(wav, mag):
    est = np.abs(A.stft(np.asarray(wav, np.float32), N_FFT, HOP, WIN)).astype(np.float64)
    return float(np.linalg.norm(est - mag) / np.linalg.norm(mag))


def test_config1_single_sentence_text_to_waveform_full_size(hparams, weights, weights64):
    """'Single LJ-Speech sentence end-to-end': the reference's __main__ path (tacotron/inference.py:130-200)
    through the drop-in modules -- text normalisation, ids, padding, network, de-normalisation, ** 1.3,
    50 Griffin-Lim iterations -- at the reference's sizes, compared with the fp64 oracle stage by stage."""
    T = pkg('tacotron.model')
    I = pkg('tacotron.inference')
    LJ = pkg('datasets.lj_speech')
    P = pkg('tacotron.params')
    dataset = LJ.LJSpeechDatasetHelper(dataset_folder=P.dataset_params.dataset_folder,
                                       char_dict=P.dataset_params.vocabulary_dict, fill_dict=False)
    text = ['Printing, in the only sense with which we are at present concerned, differs from most arts.']
    id_seqs, lengths = dataset.process_sentences(text)
    ids = np.array([I.pad_sentence(np.frombuffer(s, dtype=np.int32), max(lengths)) for s in id_seqs], dtype=np.int32)
    assert ids.shape[0] == 1 and ids[0, -1] == 1          # EOS appended
    model = T.Tacotron(inputs=T.Tacotron.model_placeholders(), mode=T.Mode.PREDICT, weights=weights)
    n_steps = model.n_steps()
    assert n_steps == 200 and hparams.reconstruction_iterations == 50
    frames = n_steps * hparams.reduction
    init = np.random.default_rng(2024).random((1, 1025, frames)).astype(np.float32)
    eng = model.engine
    out = eng.synthesize(ids, n_steps, REF_DB, MAX_DB, POWER, 50, WIN, HOP, init_phase=init, peak_normalize=False,
                         want_mel=True, want_linear=True, want_alignments=True)
    ref = O.tacotron_predict(ids, weights64, hparams, n_steps=n_steps)
    mel, lin = out['mel'].to_host(), out['linear'].to_host()
    assert mel.shape == (1, 1000, 80) and lin.shape == (1, 1000, 1025)
    e_mel, e_lin = rel_l2(mel, ref['mel']), rel_l2(lin, ref['linear'])
    e_al = float(np.abs(out['alignments'].to_host() - ref['alignments']).max())
    # what inference() returns per utterance: the (1025, T) magnitude (tacotron/inference.py:94-101)
    mags = I.inference(model, ids)
    mag_ref = A.linear_to_magnitude(ref['linear'][0].astype(np.float32), REF_DB, MAX_DB, 1.0)
    e_mag = rel_l2(mags[0], mag_ref)
    # Griffin-Lim, 50 iterations from the same initial phases, on the oracle's magnitude ** 1.3
    mag_pow = A.linear_to_magnitude(ref['linear'][0].astype(np.float32), REF_DB, MAX_DB, POWER)
    ref_wav, ref_mse = A.griffin_lim_v2(mag_pow, WIN, HOP, N_FFT, 50, init_phase=init[0])
    wav_gl, mse_gl = eng.griffin_lim(mag_pow[None], 50, WIN, HOP, N_FFT, init_phase=init)
    mse_gl = float(mse_gl.to_host()[0])
    wav = out['wav'].to_host()[0]
    assert wav.shape == (HOP * (frames - 1),) and np.isfinite(wav).all()
    sc_ref, sc_hip = spectral_convergence(ref_wav, mag_pow), spectral_convergence(wav, mag_pow)
    sc_gl = spectral_convergence(wav_gl.to_host()[0], mag_pow)
    print('config 1: mel {:.2e} linear {:.2e} align {:.2e} magnitude {:.2e}; GL(50) mse {:.6g} vs {:.6g}, spectral '
          'convergence e2e {:.5f} / staged {:.5f} vs oracle {:.5f}'.format(e_mel, e_lin, e_al, e_mag, mse_gl, ref_mse,
                                                                         sc_hip, sc_gl, sc_ref))
    assert e_mel < 1e-3 and e_lin < 1e-3 and e_al < 1e-4 and e_mag < 1e-3
    assert abs(mse_gl - ref_mse) <= 0.01 * ref_mse
    assert abs(sc_gl - sc_ref) <= 0.01 * sc_ref and abs(sc_hip - sc_ref) <= 0.01 * sc_ref
    eng.close()


def test_config4_griffin_lim_b64_t1000_60_iterations(engine):
    """'Post-net CBHG + 1025-bin linear spec + 60-iter Griffin-Lim, batch=64': the Griffin-Lim leg at full size.
    62 different spectrograms + two duplicates; the oracle follows two rows for all 60 iterations."""
    rng = np.random.default_rng(7)
    B, T = 64, 1000
    n = HOP * (T - 1)
    t = np.arange(n) / SR
    sig = np.empty((B, n), np.float32)
    for b in range(B - 2):
        f0 = 90.0 + 6.0 * b
        sig[b] = (0.3 * np.sin(2 * np.pi * f0 * t * (1 + 0.1 * np.sin(2 * np.pi * (0.3 + 0.01 * b) * t))) +
                  0.1 * np.sin(2 * np.pi * 3.1 * f0 * t) + 0.02 * rng.standard_normal(n))
    sig[62], sig[63] = sig[0], sig[1]
    mag = engine.stft_magnitude(sig, N_FFT, WIN, HOP).to_host()          # (64, 1025, 1000): inputs only
    assert mag.shape == (B, 1025, T)
    init = rng.random((B, 1025, T)).astype(np.float32)
    init[62], init[63] = init[0], init[1]
    d_mag, d_init = engine.to_device(mag), engine.to_device(init)
    wav60, mse60 = engine.griffin_lim(d_mag, 60, WIN, HOP, N_FFT, init_phase=d_init)
    wav60, mse60 = wav60.to_host(), mse60.to_host()
    assert wav60.shape == (B, n) and np.isfinite(wav60).all()
    # no cross-utterance coupling: duplicated rows are bit-identical
    assert np.array_equal(wav60[62], wav60[0]) and np.array_equal(wav60[63], wav60[1])
    assert mse60[62] == mse60[0] and mse60[63] == mse60[1]
    # the error of the estimate decreases with the iteration count (every row)
    _, mse20 = engine.griffin_lim(d_mag, 20, WIN, HOP, N_FFT, init_phase=d_init)
    _, mse40 = engine.griffin_lim(d_mag, 40, WIN, HOP, N_FFT, init_phase=d_init)
    mse20, mse40 = mse20.to_host(), mse40.to_host()
    assert np.all(mse40 < mse20) and np.all(mse60 < mse40)
    # oracle on two rows, all 60 iterations
    for b in (0, 1):
        ref_wav, ref_mse = A.griffin_lim_v2(mag[b], WIN, HOP, N_FFT, 60, init_phase=init[b])
        sc_ref, sc_hip = spectral_convergence(ref_wav, mag[b]), spectral_convergence(wav60[b], mag[b])
        print('config 4 row {}: mse {:.6g} vs oracle {:.6g}; spectral convergence {:.5f} vs {:.5f}'.format(
            b, mse60[b], ref_mse, sc_hip, sc_ref))
        assert abs(mse60[b] - ref_mse) <= 0.01 * ref_mse
        assert abs(sc_hip - sc_ref) <= 0.01 * sc_ref
    # run-to-run bitwise reproducible at full size
    wav60b, _ = engine.griffin_lim(d_mag, 60, WIN, HOP, N_FFT, init_phase=d_init)
    assert np.array_equal(wav60b.to_host(), wav60)


def test_griffin_lim_on_the_references_shipped_spectrogram(engine):
    """The only model output the reference ships: a (1, 1025, 1000, 1) linear spectrogram dump written by
    Tacotron.summary (tacotron/model.py:573-596) after 215k training steps.  Used as a magnitude spectrogram with
    the dynamic range of a trained model (6e-9 ... 0.94): one iteration sample-wise, 30 iterations through mse."""
    spec = np.load(os.path.join(GOLDEN, 'reference_linear_spec_post_215k.npz'))['linear_spec']
    assert spec.shape == (1, 1025, 1000, 1) and spec.dtype == np.float32
    mag = np.ascontiguousarray(spec[0, :, :, 0])
    init = np.random.default_rng(5).random((1, 1025, 1000)).astype(np.float32)
    ref1, mse1 = A.griffin_lim_v2(mag, WIN, HOP, N_FFT, 1, init_phase=init[0])
    wav1, m1 = engine.griffin_lim(mag[None], 1, WIN, HOP, N_FFT, init_phase=init)
    e1 = rel_l2(wav1.to_host()[0], ref1)
    ref30, mse30 = A.griffin_lim_v2(mag, WIN, HOP, N_FFT, 30, init_phase=init[0])
    wav30, m30 = engine.griffin_lim(mag[None], 30, WIN, HOP, N_FFT, init_phase=init)
    m1, m30 = float(m1.to_host()[0]), float(m30.to_host()[0])
    sc_ref, sc_hip = spectral_convergence(ref30, mag), spectral_convergence(wav30.to_host()[0], mag)
    print('reference spectrogram: 1 iteration wav rel-L2 {:.2e}, mse {:.6g} vs {:.6g}; 30 iterations mse {:.6g} vs '
          '{:.6g}, spectral convergence {:.5f} vs {:.5f}'.format(e1, m1, mse1, m30, mse30, sc_hip, sc_ref))
    assert e1 < 1e-4 and abs(m1 - mse1) <= 1e-3 * mse1
    assert abs(m30 - mse30) <= 0.01 * mse30 and abs(sc_hip - sc_ref) <= 0.01 * sc_ref
