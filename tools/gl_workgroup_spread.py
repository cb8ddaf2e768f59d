#!/usr/bin/env python3
"""When the workgroups of a Griffin-Lim launch end (a -DGL_TIMELINE build: bash tools/build_timeline.sh, SSTTS_HIP_LIB=tools/bin/lib_tl.so;
GL_TIMELINE_WGS=1 lists every workgroup with its XCD): the spread between the mean and the last end is what the cut leaves idle.
    python tools/gl_workgroup_spread.py [forced run length ...]"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sstts = importlib.import_module('single-speaker-tts_amd')
eng = sstts.Engine()
eng.set_option('debug_hooks', 1)
rng = np.random.default_rng(0)
B, T, F = 64, 1000, 1025
mag = eng.to_device((rng.random((B, F, T), dtype=np.float32) ** 4) * 10)
init = eng.to_device(rng.random((B, F, T), dtype=np.float32))
for workers in (224, 256):
    eng.set_option('gl_workers', workers)
    for rl in [int(a) for a in sys.argv[1:]] or [0]:
        eng.set_option('gl_run_len', rl)
        print('workers', workers, 'run_len', rl, flush=True)
        for _ in range(3):
            eng.griffin_lim(mag, 6, 1102, 275, 2048, init_phase=init, want_mse=False)
            eng.synchronize()
