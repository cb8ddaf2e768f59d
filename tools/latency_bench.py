#!/usr/bin/env python3
"""End-to-end latency of one tts_synthesize call (sequential, no call pipelining) for small batches."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sstts = importlib.import_module('single-speaker-tts_amd')
W = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
eng = sstts.Engine()
eng.load_weights(W.synthetic_weights(0))
eng.set_option('pipeline', 0)
if len(sys.argv) > 1:
    eng.set_option('persistent_decoder', int(sys.argv[1]))   # 2: the persistent decoder also for unpipelined calls
rng = np.random.default_rng(0)
for B in (1, 2, 4, 8, 16, 32, 64):
    ids = rng.integers(2, 39, (B, 150)).astype(np.int32)
    ids[:, -1] = 1
    d_ids = eng.to_device(ids)
    kw = dict(n_steps=200, ref_db=6.02, max_db=99.89, power=1.3, n_iter=60, win_length=1102, hop_length=275, seed=1)
    out = eng.synthesize(d_ids, **kw)
    eng.synchronize()
    eng.set_option('profile', 1)
    eng.profile_reset()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        eng.synthesize(d_ids, wav=out['wav'], **kw)
        eng.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    st = {s: eng.profile_get(s)[0] / n for s in ('encoder', 'decoder', 'postnet', 'gl_iter', 'gl_final')}
    eng.set_option('profile', 0)
    print('B=%2d: %.2f ms per call (12.46 s of audio each)  %s' % (B, dt, {k: round(v, 2) for k, v in st.items()}), flush=True)
