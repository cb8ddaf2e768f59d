"""Mirror of reference audio/synthesis.py: Griffin-Lim on the GPU.

``spectrogram_to_wav`` (:5-40) and ``griffin_lim_v2`` (:43-125) keep their signatures; two
optional keyword arguments are added because the reference draws its initial phase from the
unseeded global ``np.random`` (:85): ``init_phase`` injects those U[0,1) numbers, ``seed`` draws
them on the device.  A leading batch axis (B, F, T) reconstructs B utterances in one call."""
import numpy as np

from . import default_engine


def griffin_lim_v2(spectrogram, win_length, hop_length, n_fft, n_iter, init_phase=None, seed=None, engine=None):
    """Returns (audio float32 (n,) or (B,n), mse float32)."""
    eng = engine or default_engine()
    spec = np.asarray(spectrogram, dtype=np.float32)
    single = spec.ndim == 2
    if single:
        spec = spec[None]
        if init_phase is not None:
            init_phase = np.asarray(init_phase, dtype=np.float32)[None]
    if seed is None and init_phase is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1))   # unseeded, like the reference
    wav, mse = eng.griffin_lim(spec, n_iter, win_length, hop_length, n_fft, init_phase=init_phase, seed=seed or 0)
    wav, mse = wav.to_host(), mse.to_host()
    if n_iter == 0:
        return (wav[0], None) if single else (wav, None)
    return (wav[0], mse[0]) if single else (wav, mse)


def spectrogram_to_wav(mag, win_length, hop_length, n_fft, n_iter, init_phase=None, seed=None, engine=None):
    """reference audio/synthesis.py:5-40."""
    wav, _ = griffin_lim_v2(mag, win_length=win_length, hop_length=hop_length, n_fft=n_fft, n_iter=n_iter,
                            init_phase=init_phase, seed=seed, engine=engine)
    return wav.astype(np.float32)
