#!/usr/bin/env python3
"""Decoder-only micro benchmark (B=64, Ts=150, 200 steps): the launch-per-layer decoder against the weight-stationary kernel with 32 and
with 16 utterances per cluster (arguments: B, then any of `persistent` (no launch-per-layer run), `rows32` / `rows16` (one cluster
size only), `streamed` (decoder_persistent.hip), `cudnn`, `graph`)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sstts = importlib.import_module('single-speaker-tts_amd')
W = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
if 'cudnn' in sys.argv[2:]:   # CudnnCompatibleGRUCell arithmetic (the reference's shipped default, force_cudnn=True)
    import copy
    P = importlib.import_module('single-speaker-tts_amd.tacotron.params')
    hp = copy.deepcopy(P.ModelParams())
    hp.force_cudnn = True
    eng = sstts.Engine(hp)
    eng.load_weights(W.synthetic_weights(0, hp))
else:
    eng = sstts.Engine()
    eng.load_weights(W.synthetic_weights(0))
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mem = eng.to_device((rng.standard_normal((B, 150, 256)) * 0.5).astype(np.float32))
out = {}
FORMS = (2,) if 'persistent' in sys.argv[2:] else (0, 2)   # (under --pmc the launch-per-layer form is 12000 dispatches)
ROWS = (32,) if 'rows32' in sys.argv[2:] else ((16,) if 'rows16' in sys.argv[2:] else ((0,) if 'streamed' in sys.argv[2:] else (32, 16)))
eng.set_option('debug_hooks', 1)
if 'streamed' in sys.argv[2:]:
    eng.set_option('pd_ws', 0)   # decoder_persistent.hip instead of the weight-stationary kernel
if 'graph' in sys.argv[2:]:
    eng.set_option('use_graph', 1)
for pd, rows in [(pd, r) for pd in FORMS for r in (ROWS if pd else (0,))]:
    eng.set_option('persistent_decoder', pd)
    eng.set_option('pd_rows', rows)
    mel, al = eng.decoder_forward(mem, 200)
    eng.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        eng.decoder_forward(mem, 200, mel=mel, alignments=al)
    eng.synchronize()
    print('decoder 200 steps, B=%d: %.2f ms (%s)' % (B, (time.perf_counter() - t0) / n * 1e3,
                                                     ('persistent kernel, weights streamed' if 'streamed' in sys.argv[2:] else 'persistent kernel, weight-stationary, %d utterances per cluster' % rows) if pd else ('launch per layer, hipGraph' if 'graph' in sys.argv[2:] else 'launch per layer')), flush=True)
    out[pd] = (mel.to_host().astype(np.float64), al.to_host().astype(np.float64))
for name, i in ((('mel', 0), ('alignments', 1)) if len(FORMS) == 2 else ()):
    a, b = out[0][i], out[2][i]
    print('%s: persistent vs launch path rel-L2 %.3g, max abs %.3g' % (name, np.linalg.norm(a - b) / np.linalg.norm(a), np.abs(a - b).max()))
