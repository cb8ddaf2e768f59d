"""Mirror of reference audio/io.py ``save_wav`` (:33-53): float32 WAV, optional peak normalisation.

librosa.output.write_wav(path, y.astype(float32), sr, norm=True) = util.normalize(y, norm=inf)
then scipy.io.wavfile.write of float32 samples (WAVE_FORMAT_IEEE_FLOAT).  The normalisation runs
on the GPU (tts_peak_normalize); the container is written with the standard library."""
import struct

import numpy as np

from . import default_engine


def _write_float32_wav(path, data, sampling_rate):
    data = np.ascontiguousarray(data, dtype='<f4')
    if data.ndim == 2:          # (2, n) stereo as the reference documents -> interleave
        channels = data.shape[0]
        data = np.ascontiguousarray(data.T)
    else:
        channels = 1
    nbytes = data.nbytes
    with open(path, 'wb') as f:
        # RIFF / fmt (IEEE float, 18-byte fmt chunk) / fact / data -- what scipy.io.wavfile emits
        fmt = struct.pack('<HHIIHHH', 3, channels, sampling_rate, sampling_rate * channels * 4, channels * 4, 32, 0)
        fact = struct.pack('<I', data.shape[0])
        riff_size = 4 + (8 + len(fmt)) + (8 + len(fact)) + (8 + nbytes)
        f.write(b'RIFF' + struct.pack('<I', riff_size) + b'WAVE')
        f.write(b'fmt ' + struct.pack('<I', len(fmt)) + fmt)
        f.write(b'fact' + struct.pack('<I', len(fact)) + fact)
        f.write(b'data' + struct.pack('<I', nbytes))
        f.write(data.tobytes())


def save_wav(wav_path, wav, sampling_rate, norm=False, engine=None):
    """reference audio/io.py:33-53."""
    wav = np.asarray(wav).astype(np.float32)
    if norm:
        eng = engine or default_engine()
        flat = wav.reshape(1, -1)            # inf-norm over the whole array (axis=None)
        wav = eng.peak_normalize(eng.to_device(flat)).to_host().reshape(wav.shape)
    _write_float32_wav(wav_path, wav, sampling_rate)


def load_wav(wav_path, sampling_rate=None, offset=0.0, duration=None):
    raise NotImplementedError('load_wav (librosa resampling loader) is outside the inference path')
