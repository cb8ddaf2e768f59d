"""CPU oracle for the audio side of the inference path (numpy restatement).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

PARITY UNPINNED: the STFT / iSTFT / mel-filterbank / WAV-normalise arithmetic
lives in librosa (requirements.txt:2 of the reference, ``>= 0.6.1``; the
``librosa.output.write_wav`` call at audio/io.py:53 bounds it to < 0.8), which is
absent from /root/reference and not installable here; the reference holds no
golden vectors for it.  The functions below restate librosa-0.6.x's published
algorithms (stft / istft / window_sumsquare / filters.mel / util.normalize) at the
reference's call sites, including its dtype behaviour (float64 FFT, float32
overlap-add buffer, complex64 STFT matrix).  Closed-form identities and an
independent torch.stft/istft formulation cross-check them in tests/.

Reference call sites: audio/synthesis.py:5-125, audio/conversion.py:5-136,
audio/features.py:5-145, audio/io.py:33-53, tacotron/inference.py:93-101,170-182.
"""
import numpy as np


# --------------------------------------------------------------------------- conversion
def magnitude_to_decibel(mag):
    """reference audio/conversion.py:5-29."""
    return 20.0 * np.log10(np.maximum(1e-5, mag))


def decibel_to_magnitude(mag_db):
    """reference audio/conversion.py:32-53 (raises on dB < -100)."""
    if (mag_db < -100.0).any():
        raise AssertionError('"conversion.decibel_to_magnitude" was asked to convert a dB value '
                             'smaller -100 dB.')
    return np.power(10.0, mag_db / 20.0)


def normalize_decibel(db, ref_db, max_db):
    """reference audio/conversion.py:56-78."""
    return np.clip(1.0 + (db - ref_db) / (abs(ref_db) + abs(max_db)), 0.0, 1.0)


def inv_normalize_decibel(norm_db, ref_db, max_db):
    """reference audio/conversion.py:81-102."""
    return ((np.clip(norm_db, 0.0, 1.0) - 1.0) * (abs(ref_db) + abs(max_db))) + ref_db


def ms_to_samples(ms, sampling_rate):
    """reference audio/conversion.py:122-136."""
    return int((ms / 1000) * sampling_rate)


def linear_to_magnitude(linear, ref_db, max_db, power):
    """reference tacotron/inference.py:93-101 + :175 for ONE utterance.

    linear (T,F) float32 network output -> (F,T) float32 magnitude ** power.  float32
    array (.) python scalar stays float32 in numpy, as in the reference."""
    spec = np.asarray(linear)
    db = inv_normalize_decibel(spec.T, ref_db, max_db)
    mag = decibel_to_magnitude(db)
    return np.power(mag, power)


# --------------------------------------------------------------------------- librosa-0.6 pieces
def hann_periodic(win_length):
    """scipy.signal.get_window('hann', M, fftbins=True) [librosa-0.6 filters.get_window]."""
    n = np.arange(win_length, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)


def pad_center(data, size):
    """librosa.util.pad_center for 1-D data: left pad (size-n)//2."""
    n = data.shape[0]
    lpad = int((size - n) // 2)
    return np.pad(data, (lpad, int(size - n - lpad)), mode='constant')


def window_sumsquare(n_frames, hop_length, win_length, n_fft, dtype=np.float32):
    """librosa.filters.window_sumsquare(window='hann', norm=None): sequential accumulation
    of the padded squared window into a ``dtype`` buffer (S11)."""
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, dtype=dtype)
    win_sq = pad_center(hann_periodic(win_length) ** 2, n_fft)
    for i in range(n_frames):
        sample = i * hop_length
        x[sample:min(n, sample + n_fft)] += win_sq[:max(0, min(n_fft, n - sample))]
    return x


def stft(y, n_fft, hop_length, win_length, dtype=np.complex64):
    """librosa.stft(window='hann', center=True, pad_mode='reflect') (S10).

    Window padded centrally to n_fft (float64), reflect padding by n_fft//2, frames at hop,
    float64 FFT of window*frame, first 1+n_fft/2 bins, cast to complex64.  Returns (F, n_frames)."""
    fft_window = pad_center(hann_periodic(win_length), n_fft)
    y = np.asarray(y)
    yp = np.pad(y, int(n_fft // 2), mode='reflect')
    n_frames = 1 + int((len(yp) - n_fft) / hop_length)
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    frames = yp[idx]                                  # (n_fft, n_frames), dtype of y
    spec = np.fft.rfft(fft_window[:, None] * frames, axis=0)  # float64 math
    return spec.astype(dtype)


def istft(stft_matrix, hop_length, win_length, dtype=np.float32):
    """librosa.istft(window='hann', center=True, length=None) (S11).

    Per frame: hermitian extension -> ifft (imaginary parts of the DC and Nyquist bins drop
    out of .real) -> * padded window -> overlap-add into a ``dtype`` buffer (each += rounds
    to dtype, as ``y[a:b] = y[a:b] + ytmp`` does) -> divide by window_sumsquare where it
    exceeds tiny -> trim n_fft//2 each side."""
    n_fft = 2 * (stft_matrix.shape[0] - 1)
    ifft_window = pad_center(hann_periodic(win_length), n_fft)
    n_frames = stft_matrix.shape[1]
    y = np.zeros(n_fft + hop_length * (n_frames - 1), dtype=dtype)
    frames = np.fft.irfft(np.asarray(stft_matrix, dtype=np.complex128), n=n_fft, axis=0)
    frames = ifft_window[:, None] * frames            # float64
    for i in range(n_frames):
        sample = i * hop_length
        y[sample:sample + n_fft] = y[sample:sample + n_fft] + frames[:, i]
    wss = window_sumsquare(n_frames, hop_length, win_length, n_fft, dtype=dtype)
    nz = wss > np.finfo(wss.dtype).tiny
    y[nz] /= wss[nz]
    return y[int(n_fft // 2):-int(n_fft // 2)]


def griffin_lim_v2(spectrogram, win_length, hop_length, n_fft, n_iter, init_phase=None, rng=None,
                   history=None):
    """reference audio/synthesis.py:43-125 (S12) with the random initial phase INJECTED.

    ``init_phase``: array of U[0,1) numbers shaped like ``spectrogram`` (what
    ``np.random.rand(*spectrogram.shape)`` returns at synthesis.py:85), or None to draw from
    ``rng``.  Returns (signal float32, mse)."""
    spectrogram = np.asarray(spectrogram)
    if init_phase is None:
        rng = rng or np.random.default_rng()
        init_phase = rng.random(spectrogram.shape)
    mse = None
    angles = np.exp(2j * np.pi * np.asarray(init_phase, dtype=np.float64))
    mag = np.abs(spectrogram).astype(np.complex128)
    for _ in range(n_iter):
        full = mag * angles
        estimated_signal = istft(full, hop_length, win_length)
        estimated_stft = stft(estimated_signal, n_fft, hop_length, win_length)
        # exp(1j*angle(z)) evaluated in complex64 like numpy does for a complex64 input.
        ang = np.angle(estimated_stft)
        angles = (np.cos(ang) + 1j * np.sin(ang)).astype(np.complex64)
        mse = np.square(np.abs(spectrogram) - np.abs(estimated_stft)).mean()
        if history is not None:
            history.append(dict(signal=estimated_signal, angles=angles, mse=mse))
    full = mag * angles
    estimated_signal = istft(full, hop_length, win_length)
    return estimated_signal, mse


def spectrogram_to_wav(mag, win_length, hop_length, n_fft, n_iter, init_phase=None, rng=None):
    """reference audio/synthesis.py:5-40."""
    wav, _ = griffin_lim_v2(mag, win_length, hop_length, n_fft, n_iter, init_phase, rng)
    return wav.astype(np.float32)


def peak_normalize(wav):
    """librosa.util.normalize(y, norm=inf) as used by librosa.output.write_wav(norm=True)
    (reference audio/io.py:53): divide by max|y| unless it is below tiny."""
    wav = np.asarray(wav, dtype=np.float32)
    length = np.max(np.abs(wav))
    if length < np.finfo(np.float32).tiny:
        length = np.float32(1.0)
    return wav / length


# --------------------------------------------------------------------------- analysis features
def hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(htk=True, norm=1) [librosa-0.6]; reference audio/features.py:75-80."""
    if fmax is None:
        fmax = float(sr) / 2
    weights = np.zeros((n_mels, int(1 + n_fft // 2)))
    fftfreqs = np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    mel_f = mel_to_hz_htk(np.linspace(hz_to_mel_htk(fmin), hz_to_mel_htk(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def linear_scale_spectrogram(wav, n_fft, hop_length, win_length):
    """reference audio/features.py:116-145."""
    return stft(wav, n_fft, hop_length, win_length)


def mel_scale_spectrogram(wav, n_fft, sampling_rate, n_mels, fmin, fmax, hop_length, win_length, power):
    """reference audio/features.py:5-86.  Returns (mel (n_mels,t), linear (F,t))."""
    mag_spec = np.abs(stft(wav, n_fft, hop_length, win_length))
    linear_spec = mag_spec ** power
    mel_basis = mel_filterbank(sampling_rate, n_fft, n_mels, fmin, fmax)
    return np.dot(mel_basis, linear_spec), linear_spec
