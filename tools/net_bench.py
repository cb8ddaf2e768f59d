#!/usr/bin/env python3
"""Encoder / post-net stage micro benchmark at the bench shapes (B=64)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sstts = importlib.import_module('single-speaker-tts_amd')
W = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
eng = sstts.Engine()
eng.load_weights(W.synthetic_weights(0))
rng = np.random.default_rng(0)
ids = eng.to_device(rng.integers(2, 39, (64, 150)).astype(np.int32))
mel = eng.to_device(rng.random((64, 1000, 80), dtype=np.float32))
mem = eng.encoder_forward(ids)
lin = eng.postnet_forward(mel)
eng.set_option('profile', 1)
n = 5
for fused in (0, 1, 0, 1):   # the CBHG tail as the GEMM chain / as one launch (csrc/cbhg_tail.hip)
    eng.set_option('fused_tail', fused)
    eng.encoder_forward(ids, out=mem)
    eng.postnet_forward(mel, out=lin)
    eng.profile_reset()
    for _ in range(n):
        eng.encoder_forward(ids, out=mem)
        eng.postnet_forward(mel, out=lin)
    print('fused_tail %d: encoder %.3f ms, postnet %.3f ms' % (fused, eng.profile_get('encoder')[0] / n, eng.profile_get('postnet')[0] / n))
