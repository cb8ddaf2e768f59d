#!/usr/bin/env python3
"""Every stage of the library beside MFMA GEMM launches of a second handle, compared bit for bit with its quiet result:
    python tools/stage_beside_gemm.py REPS [stage ...]      (tests/test_gpu_neighbours.py is the short form)
Round 6: the general Griffin-Lim kernels built WITH packed-f32 instructions fail this 106 of 120 times (profiles/
r06_experiment_packed_f32_beside_mfma.txt); select such a build with SSTTS_HIP_LIB."""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sstts = importlib.import_module('single-speaker-tts_amd')
W = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
from test_gpu_round5 import bench_ids
eng = sstts.Engine(); eng.load_weights(W.synthetic_weights(0))
eng2 = sstts.Engine(); eng2.load_weights(W.synthetic_weights(0))
rng = np.random.default_rng(0)
F = 1025
x = eng2.to_device(rng.standard_normal((9600, 256)).astype(np.float32)); w = eng2.to_device(rng.standard_normal((256, 256)).astype(np.float32)); c = eng2.empty((9600, 256))
def aggr():
    for _ in range(30):
        eng2._check(eng2.lib.tts_debug_gemm(eng2.handle, x.data_ptr(), w.data_ptr(), c.data_ptr(), 9600, 256, 256, 1, 150, 0))
ids = eng.to_device(bench_ids(16, 60, 5))
mem = eng.to_device((rng.standard_normal((16, 60, 256)) * 0.5).astype(np.float32))
melb = eng.to_device(rng.standard_normal((16, 100, 80)).astype(np.float32) * 0.1)
magg = eng.to_device((rng.random((16, F, 100), dtype=np.float32) ** 4) * 10)
mags = eng.to_device((rng.random((16, F, 200), dtype=np.float32) ** 4) * 10)
mag1k = eng.to_device((rng.random((16, 513, 100), dtype=np.float32) ** 4) * 10)
def first(r): return r[0] if isinstance(r, tuple) else r
victims = {
    'encoder': lambda: eng.encoder_forward(ids),
    'decoder_launch': lambda: (eng.set_option('persistent_decoder', 0), first(eng.decoder_forward(mem, 10)))[1],
    'decoder_streamed': lambda: (eng.set_option('persistent_decoder', 2), eng.set_option('debug_hooks', 1), eng.set_option('pd_ws', 0), first(eng.decoder_forward(mem, 10)), eng.set_option('pd_ws', 1))[3],
    'decoder_ws': lambda: (eng.set_option('persistent_decoder', 2), first(eng.decoder_forward(mem, 10)))[1],
    'postnet': lambda: first(eng.postnet_forward(melb)),
    'gl_general_2048': lambda: first(eng.griffin_lim(magg, 2, 1200, 300, 2048, seed=3, want_mse=False)),
    'gl_general_1024': lambda: first(eng.griffin_lim(mag1k, 2, 800, 200, 1024, seed=3, want_mse=False)),
    'gl_stream': lambda: first(eng.griffin_lim(mags, 3, 1102, 275, 2048, seed=3, want_mse=False)),
}
for name in (sys.argv[2:] or list(victims)):
    v = victims[name]
    ref = v(); eng.synchronize(); ref = ref.to_host().copy()
    bad = n = 0
    for rep in range(int(sys.argv[1])):
        aggr()
        outs = [v() for _ in range(2)]
        eng.synchronize(); eng2.synchronize()
        for o in outs:
            n += 1
            if not np.array_equal(o.to_host(), ref, equal_nan=True): bad += 1
    print(name, 'mismatches', bad, 'of', n, flush=True)
