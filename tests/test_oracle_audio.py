"""CPU: the numpy audio oracle against its golden vectors, torch.stft/istft and closed forms."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import audio_oracle as A

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
N_FFT, WIN, HOP = 2048, 1102, 275


def test_ms_to_samples_reference_values():
    # reference audio/conversion.py:136 with params/model.py:13-24
    assert A.ms_to_samples(50.0, 22050) == 1102
    assert A.ms_to_samples(12.5, 22050) == 275


def test_db_identities():
    ref, mx = 6.02, 99.89
    x = np.linspace(0, 1, 11)
    assert np.allclose(A.normalize_decibel(A.inv_normalize_decibel(x, ref, mx), ref, mx), x)
    assert np.isclose(A.inv_normalize_decibel(np.array([1.0]), ref, mx)[0], ref)
    assert np.isclose(A.inv_normalize_decibel(np.array([0.0]), ref, mx)[0], ref - (abs(ref) + abs(mx)))
    assert np.allclose(A.decibel_to_magnitude(A.magnitude_to_decibel(np.array([1e-3, 1.0, 7.0]))), [1e-3, 1.0, 7.0])
    assert A.magnitude_to_decibel(np.array([0.0]))[0] == -100.0
    with pytest.raises(AssertionError):
        A.decibel_to_magnitude(np.array([-100.5]))
    lin = np.array([[-.5, 0., .5, 1., 1.5]], np.float32)
    m = A.linear_to_magnitude(lin, ref, mx, 1.3)
    assert m.dtype == np.float32 and m.shape == (5, 1)
    assert m[0, 0] == m[1, 0] and m[3, 0] == m[4, 0]           # clipped both sides


def test_window_and_sumsquare():
    w = A.hann_periodic(WIN)
    assert w[0] == 0.0 and np.isclose(w[WIN // 2], 1.0)
    assert np.allclose(w[1:], w[1:][::-1])                      # periodic hann symmetry
    T = 9
    wss = A.window_sumsquare(T, HOP, WIN, N_FFT)
    assert wss.dtype == np.float32 and wss.shape == (N_FFT + HOP * (T - 1),)
    ref = np.zeros(wss.shape[0])
    wp = A.pad_center(w ** 2, N_FFT)
    for i in range(T):
        ref[i * HOP:i * HOP + N_FFT] += wp
    assert np.allclose(wss, ref, rtol=1e-6)
    assert (N_FFT - WIN) // 2 == 473


def test_stft_matches_torch():
    rng = np.random.default_rng(0)
    y = rng.standard_normal(HOP * 20).astype(np.float32)
    S = A.stft(y, N_FFT, HOP, WIN)
    assert S.dtype == np.complex64 and S.shape == (1025, 21)     # 1 + len // hop frames
    St = torch.stft(torch.tensor(y, dtype=torch.float64), N_FFT, HOP, WIN,
                    window=torch.hann_window(WIN, periodic=True, dtype=torch.float64), center=True,
                    pad_mode='reflect', return_complex=True).numpy()
    assert np.linalg.norm(S - St) / np.linalg.norm(St) < 1e-6


def test_istft_matches_torch_and_inverts_stft():
    rng = np.random.default_rng(1)
    y = rng.standard_normal(HOP * 16).astype(np.float32)
    S = A.stft(y, N_FFT, HOP, WIN)
    yi = A.istft(S, HOP, WIN)
    assert yi.dtype == np.float32 and yi.shape == (HOP * 16,)    # hop * (T - 1)
    assert np.abs(yi - y)[WIN:-WIN].max() < 1e-5
    yt = torch.istft(torch.tensor(S.astype(np.complex128)), N_FFT, HOP, WIN,
                     window=torch.hann_window(WIN, periodic=True, dtype=torch.float64), center=True,
                     length=HOP * 16).numpy()
    assert np.abs(yi - yt)[8:-8].max() < 1e-5


def test_istft_ignores_imaginary_dc_and_nyquist():
    rng = np.random.default_rng(2)
    S = (rng.standard_normal((1025, 6)) + 1j * rng.standard_normal((1025, 6))).astype(np.complex64)
    S2 = S.copy()
    S2[0] = S2[0].real
    S2[-1] = S2[-1].real
    assert np.array_equal(A.istft(S, HOP, WIN), A.istft(S2, HOP, WIN))


def test_griffin_lim_golden_and_convergence():
    g = np.load(os.path.join(GOLD, 'griffin_lim_small.npz'))
    hist = []
    wav, mse = A.griffin_lim_v2(g['mag'], WIN, HOP, N_FFT, int(g['n_iter']), init_phase=g['init_phase'], history=hist)
    assert wav.dtype == np.float32
    assert rel_l2(wav, g['wav']) < 1e-6 and abs(mse - g['mse']) < 1e-6 * g['mse']
    assert rel_l2(hist[0]['signal'], g['wav_n0']) < 1e-6       # the signal of iteration 1 is istft of the initial phases
    assert rel_l2(hist[1]['signal'], g['wav_n1']) < 1e-6
    assert np.allclose(np.abs(hist[0]['angles']), 1.0, atol=1e-6)
    _, mse10 = A.griffin_lim_v2(g['mag'], WIN, HOP, N_FFT, 10, init_phase=g['init_phase'])
    assert mse10 < mse                                            # reconstruction error decreases
    assert rel_l2(A.linear_to_magnitude(g['linear'], 6.02, 99.89, 1.3), g['linear_mag']) < 1e-7
    assert np.array_equal(A.peak_normalize(g['wav']), g['peak_norm'])
    assert np.abs(g['peak_norm']).max() == 1.0


def test_zero_spectrum_gives_unit_phase_and_silence():
    wav, mse = A.griffin_lim_v2(np.zeros((1025, 6), np.float32), WIN, HOP, N_FFT, 2, init_phase=np.zeros((1025, 6)))
    assert np.array_equal(wav, np.zeros(HOP * 5, np.float32)) and mse == 0
    assert np.array_equal(A.peak_normalize(np.zeros(4)), np.zeros(4, np.float32))


def test_mel_filterbank_htk_slaney():
    M = A.mel_filterbank(22050, 2048, 80, 0, 8000)
    assert M.shape == (80, 1025) and (M >= 0).all()
    freqs = np.linspace(0, 11025, 1025)
    mel_f = A.mel_to_hz_htk(np.linspace(A.hz_to_mel_htk(0), A.hz_to_mel_htk(8000), 82))
    assert np.isclose(A.hz_to_mel_htk(1000.0), 2595 * np.log10(1 + 1000 / 700))
    assert (M[:, freqs > 8000] == 0).all()
    # Slaney area normalisation: each triangle integrates to ~1 (discretisation error at low bins)
    area = (M * (freqs[1] - freqs[0])).sum(1)
    assert np.allclose(area[20:], 1.0, rtol=0.05)
    peak = M.argmax(1)
    assert (np.diff(peak) > 0).all()
    assert np.allclose(freqs[peak], mel_f[1:-1], atol=freqs[1])
