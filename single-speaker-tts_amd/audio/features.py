"""Mirror of reference audio/features.py (analysis side): STFT and HTK-mel spectrograms on the GPU."""
import numpy as np

from . import default_engine


def linear_scale_spectrogram(wav, n_fft, hop_length=None, win_length=None, engine=None):
    """librosa.stft(wav, n_fft, hop_length, win_length): complex64 (1 + n_fft/2, t)
    (reference audio/features.py:116-145)."""
    eng = engine or default_engine()
    win_length = win_length or n_fft
    hop_length = hop_length or int(win_length // 4)
    wav = np.asarray(wav, dtype=np.float32)
    return eng.stft(wav[None], n_fft, win_length, hop_length).to_host()[0]


def mel_scale_spectrogram(wav, n_fft, sampling_rate, n_mels, fmin, fmax, hop_length, win_length, power, engine=None):
    """reference audio/features.py:5-86: mel (n_mels, t) of abs(stft) ** power, HTK mel scale."""
    eng = engine or default_engine()
    wav = np.asarray(wav, dtype=np.float32)
    lin = eng.stft_magnitude(wav[None], n_fft, win_length, hop_length, power)
    return eng.mel_spectrogram(lin, n_fft, sampling_rate, n_mels, fmin, fmax).to_host()[0]


def calculate_mfccs(mel_spec, sampling_rate, n_mfcc):
    raise NotImplementedError('calculate_mfccs is unused by the reference (audio/features.py:89-113) and out of scope')
