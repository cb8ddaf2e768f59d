#!/usr/bin/env python3
"""Griffin-Lim iteration time across (n_fft, win, hop): the two streaming instantiations and the general kernels
(griffin_lim_generic.hip), B = 64, T = 1000 frames, 30 iterations after a 3-iteration warm-up."""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sstts = importlib.import_module('single-speaker-tts_amd')
eng = sstts.Engine()
rng = np.random.default_rng(0)
for n_fft, win, hop in ((2048, 1102, 275), (2048, 800, 200), (2048, 1200, 300), (1024, 800, 200), (1024, 551, 137), (4096, 2400, 600), (512, 400, 100)):
    B, T = 64, 1000
    F = 1 + n_fft // 2
    mag = eng.to_device((rng.random((B, F, T), dtype=np.float32) ** 4) * 10)
    init = eng.to_device(rng.random((B, F, T), dtype=np.float32))
    eng.griffin_lim(mag, 3, win, hop, n_fft, init_phase=init, want_mse=False)
    eng.synchronize()
    t0 = time.perf_counter()
    eng.griffin_lim(mag, 30, win, hop, n_fft, init_phase=init, want_mse=False)
    eng.synchronize()
    dt = (time.perf_counter() - t0)
    print('n_fft {:4d} win {:4d} hop {:3d}: {:7.1f} us per iteration (B = 64, T = 1000), {:.1f} ps per bin and iteration'.format(n_fft, win, hop, dt / 30 * 1e6, dt / 30 * 1e12 / (B * F * T)), flush=True)
    mag.free(); init.free()
