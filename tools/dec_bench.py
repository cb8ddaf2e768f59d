#!/usr/bin/env python3
"""Decoder-only micro benchmark (B=64, Ts=150, 200 steps)."""
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
graph = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng.set_option('use_graph', graph)
mel, al = eng.decoder_forward(mem, 200)
eng.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    eng.decoder_forward(mem, 200, mel=mel, alignments=al)
eng.synchronize()
print('decoder 200 steps: %.2f ms (graph=%d)' % ((time.perf_counter() - t0) / n * 1e3, graph))
