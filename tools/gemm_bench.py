#!/usr/bin/env python3
"""GEMM / conv1d kernel micro benchmark at the shapes of the two CBHG stacks (tts_debug_gemm)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sstts = importlib.import_module('single-speaker-tts_amd')
eng = sstts.Engine()
if len(sys.argv) > 1:
    eng.set_option('gemm_presplit', int(sys.argv[1]))   # 0: the weights split in the kernel, tile by tile (round 4)
if len(sys.argv) > 2:
    eng.set_option('gemm_ps', int(sys.argv[2]))         # 1: producer / consumer form (512 threads), 0: 256 threads (round 4)
rng = np.random.default_rng(0)
shapes = [  # name, M, N, Cin, ktaps, T, pool
    ('dense 64000x256x3072', 64000, 256, 3072, 1, 1000, 0),
    ('post proj1 conv3 1024->256 +pool', 64000, 256, 1024, 3, 1000, 1),
    ('post proj1 conv3 1024->256', 64000, 256, 1024, 3, 1000, 0),
    ('final dense 256->1025', 64000, 1025, 256, 1, 1000, 0),
    ('dense 64000x1024x256', 64000, 1024, 256, 1, 1000, 0),
    ('dense 64000x128x128', 64000, 128, 128, 1, 1000, 0),
    ('enc proj1 conv3 2048->128 +pool', 9600, 128, 2048, 3, 150, 1),
    ('dense 8192x8192x1024', 8192, 8192, 1024, 1, 8192, 0),
]
for name, M, N, Cin, kt, T, pool in shapes:
    K = Cin * kt
    A = eng.to_device(rng.standard_normal((M + 8, Cin), dtype=np.float32))
    W = eng.to_device(rng.standard_normal((N, K), dtype=np.float32) * 0.05)
    C = eng.empty((M, N))
    def run():
        eng._check(eng.lib.tts_debug_gemm(eng.handle, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, Cin, kt, T, pool))
    run(); eng.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / n
    # the fastest single launch (HIP events of the library): what a micro-benchmark's best-of-n reports; the average of
    # back-to-back launches above is lower because the clock drops under sustained matrix load
    eng.set_option('profile', 1)
    best = 1e9
    for _ in range(n):
        eng.profile_reset()
        run()
        eng.synchronize()
        ms, cnt = eng.profile_get('debug_gemm')
        best = min(best, ms / max(1, cnt))
    # ... and the average of ten back-to-back launches by the same events (the wall-clock figure above includes the
    # pack kernel that tts_debug_gemm runs on the caller's weights before every launch)
    eng.profile_reset()
    for _ in range(10):
        run()
    eng.synchronize()
    ms, cnt = eng.profile_get('debug_gemm')
    avg = ms / max(1, cnt)
    eng.set_option('profile', 0)
    print('%-36s %8.1f us  %6.1f TFLOP/s by events over 10 launches  (wall %8.1f us; fastest launch %8.1f us  %6.1f)' % (
        name, avg * 1e3, 2.0 * M * N * K / (avg * 1e-3) / 1e12, dt * 1e6, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12), flush=True)
    A.free(); W.free(); C.free()
