#!/usr/bin/env python3
"""Decoder loop on a partially occupied GPU: `hold` workgroups of `lds` KB keep CUs busy (no memory traffic)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sstts = importlib.import_module('single-speaker-tts_amd')
W = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
eng = sstts.Engine()
eng.load_weights(W.synthetic_weights(0))
rng = np.random.default_rng(0)
mem = eng.to_device((rng.standard_normal((64, 150, 256)) * 0.5).astype(np.float32))
mel, al = eng.decoder_forward(mem, 200)
eng.synchronize()
for hold, lds in [(0, 0), (224, 152), (228, 152), (248, 100)]:
    if hold:
        eng.set_option('debug_hooks', 1)
        eng._check(eng.lib.tts_debug_hold(eng.handle, hold, lds, 80.0))
        time.sleep(0.005)
    t0 = time.perf_counter()
    eng.decoder_forward(mem, 200, mel=mel, alignments=al)
    eng.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print('hold %3d x %3d KB: decoder 200 steps %.2f ms' % (hold, lds, dt), flush=True)
    time.sleep(0.15)
