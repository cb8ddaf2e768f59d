#!/usr/bin/env python3
"""Analysis-STFT micro benchmark: 64 x 274725 samples -> 64 x 1000 frames."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sstts = importlib.import_module('single-speaker-tts_amd')
eng = sstts.Engine()
rng = np.random.default_rng(0)
wav = eng.to_device(rng.standard_normal((64, 274725)).astype(np.float32))
out = eng.stft_magnitude(wav, 2048, 1102, 275, 1.0)
eng.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    eng.lib.tts_stft_magnitude(eng.handle, wav.ptr, 64, 274725, 2048, 1102, 275, 1.0, out.ptr)
eng.synchronize()
print('stft_magnitude 64x1000 frames: %.1f us per call (includes the (B,T,F)->(B,F,T) transpose)' % ((time.perf_counter() - t0) / n * 1e6))
