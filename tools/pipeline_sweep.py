#!/usr/bin/env python3
"""Throughput of back-to-back tts_synthesize calls (call pipelining on) over the batch size, with the decoder as the
persistent kernel or as the launch-per-layer graph: where does the persistent form pay?"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sstts = importlib.import_module('single-speaker-tts_amd')
W = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
eng = sstts.Engine()
eng.load_weights(W.synthetic_weights(0))
rng = np.random.default_rng(0)
for B in [int(x) for x in os.environ.get('SWEEP_B', '1,4,8,16,32,48,64').split(',')]:
    ids = rng.integers(2, 39, (B, 150)).astype(np.int32)
    ids[:, -1] = 1
    d_ids = eng.to_device(ids)
    kw = dict(n_steps=200, ref_db=6.02, max_db=99.89, power=1.3, n_iter=60, win_length=1102, hop_length=275, seed=1)
    res = []
    for pd in (0, 2):
        eng.set_option('persistent_decoder', pd)
        out = eng.synthesize(d_ids, **kw)
        for _ in range(3):
            eng.synthesize(d_ids, wav=out['wav'], **kw)
        eng.synchronize()
        n = 16
        t0 = time.perf_counter()
        for _ in range(n):
            eng.synthesize(d_ids, wav=out['wav'], **kw)
        eng.synchronize()
        res.append((time.perf_counter() - t0) / n * 1e3)
    print('B=%2d: %.2f ms per call launch-per-layer, %.2f ms persistent' % (B, res[0], res[1]), flush=True)
eng.set_option('persistent_decoder', 1)
