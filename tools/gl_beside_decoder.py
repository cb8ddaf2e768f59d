#!/usr/bin/env python3
"""Griffin-Lim launches on 224 workgroups with and without a persistent decoder running beside them on a second handle
(bench.py's gl_beside_decoder harness), on the timed workload's spectra and on the reference's shipped (trained) spectrogram.

    python tools/gl_beside_decoder.py [--launches 5] [--reps 4]
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--launches', type=int, default=5)
    ap.add_argument('--reps', type=int, default=4)
    ap.add_argument('--prio', action='store_true', help='the decoder on a stream of the greatest priority (as the front stream of the call pipeline)')
    a = ap.parse_args()
    sstts = importlib.import_module('single-speaker-tts_amd')
    P = importlib.import_module('single-speaker-tts_amd.tacotron.params')
    Wm = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
    hp = P.ModelParams()
    blob = Wm.pack_blob(Wm.synthetic_weights(0, hp), hp)
    eng = sstts.Engine(hp)
    eng.load_weights_blob(blob)
    B, T = bench.B_PER_GPU, bench.N_STEPS * hp.reduction
    ids = eng.to_device(bench.synthetic_ids(B, bench.TS, 1234))
    out = eng.synthesize(ids, bench.N_STEPS, bench.REF_DB, bench.MAX_DB, bench.POWER, 3, bench.WIN, bench.HOP, seed=1, want_linear=True)
    mags = {'contract': eng.denorm_power(out['linear'], bench.REF_DB, bench.MAX_DB, bench.POWER),
            'trained': eng.to_device(bench.trained_spectrum_batch(B, T)),
            'random': eng.to_device((np.random.default_rng(1).random((B, 1025, T), dtype=np.float32) ** 4) * 10 + 1e-3)}
    _, cus = eng.device_info()
    stream = None
    if a.prio:
        import ctypes
        hip = ctypes.CDLL('libamdhip64.so')
        lo, hi = ctypes.c_int(0), ctypes.c_int(0)
        assert hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) == 0
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithPriority(ctypes.byref(st), 1, hi.value) == 0   # hipStreamNonBlocking
        stream = st.value
    for with_dec in (True, False):
        r = bench.gl_beside_decoder(sstts, eng, hp, blob, ids, mags, B, T, cus, 32, 3, bench.N_STEPS, n_launches=a.launches, reps=a.reps,
                                    with_decoder=with_dec, decoder_stream=stream)
        print('decoder beside' if with_dec else 'no decoder   ', {k: round(v * 1e3, 1) for k, v in r.items()}, 'us per iteration', flush=True)


if __name__ == '__main__':
    main()
