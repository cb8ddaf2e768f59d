"""Mirror of reference audio/conversion.py (same names, arguments and error behaviour).

Array conversions execute on the GPU through ``tts_db_convert``; results are float32 (the
reference's float32 inputs stay float32 as well).  The scalar helpers are host arithmetic."""
import numpy as np

from . import default_engine


def _convert(x, mode, ref_db=0.0, max_db=0.0, engine=None):
    x = np.asarray(x)
    out = (engine or default_engine()).db_convert(x.reshape(-1), mode, ref_db, max_db)
    return out.reshape(x.shape)


def magnitude_to_decibel(mag, engine=None):
    """20 * log10(max(1e-5, mag))   (reference audio/conversion.py:5-29)."""
    return _convert(mag, 0, engine=engine)


def decibel_to_magnitude(mag_db, engine=None):
    """power(10, mag_db / 20); AssertionError below -100 dB (reference audio/conversion.py:32-53)."""
    return _convert(mag_db, 1, engine=engine)


def normalize_decibel(db, ref_db, max_db, engine=None):
    """clip(1 + (db - ref_db) / (|ref_db| + |max_db|), 0, 1)   (reference audio/conversion.py:56-78)."""
    return _convert(db, 2, ref_db, max_db, engine)


def inv_normalize_decibel(norm_db, ref_db, max_db, engine=None):
    """(clip(norm_db, 0, 1) - 1) * (|ref_db| + |max_db|) + ref_db   (reference audio/conversion.py:81-102)."""
    return _convert(norm_db, 3, ref_db, max_db, engine)


def samples_to_ms(samples, sampling_rate):
    """reference audio/conversion.py:105-119."""
    return (samples / sampling_rate) * 1000


def ms_to_samples(ms, sampling_rate):
    """reference audio/conversion.py:122-136."""
    return int((ms / 1000) * sampling_rate)


def get_duration(wav, sr):
    """reference audio/conversion.py:139-152 (librosa.core.get_duration of a time series)."""
    return float(np.asarray(wav).shape[-1]) / float(sr)
