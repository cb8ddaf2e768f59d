#!/usr/bin/env python3
"""Griffin-Lim-only micro benchmark (used under rocprofv3 for PMC passes).

    python tools/gl_bench.py [--B 64] [--T 1000] [--iters 60] [--reps 3]
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--T', type=int, default=1000)
    ap.add_argument('--iters', type=int, default=60)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--nocheck', action='store_true', help='ablation builds produce garbage')
    ap.add_argument('--pair', type=int, default=None, help='override the library default (iterations per launch, 1..3)')
    ap.add_argument('--workers', type=int, default=0, help='plan and launch for this many workgroups (compute units) instead of all')
    a = ap.parse_args()
    sstts = importlib.import_module('single-speaker-tts_amd')
    eng = sstts.Engine()
    rng = np.random.default_rng(0)
    mag = eng.to_device((rng.random((a.B, 1025, a.T), dtype=np.float32) ** 4) * 10)
    init = eng.to_device(rng.random((a.B, 1025, a.T), dtype=np.float32))
    if a.pair is not None:
        eng.set_option('gl_pair', a.pair)
    if a.workers:
        eng.set_option('debug_hooks', 1)
        eng.set_option('gl_workers', a.workers)
    eng.griffin_lim(mag, 2, 1102, 275, 2048, init_phase=init, want_mse=False)
    eng.set_option('profile', 1)
    eng.profile_reset()
    for _ in range(a.reps):
        wav, _ = eng.griffin_lim(mag, a.iters, 1102, 275, 2048, init_phase=init, want_mse=False)
    ms, n = eng.profile_get('gl_iter')
    msf, nf = eng.profile_get('gl_final')
    per = ms / max(1, n)
    alg = 20.0 * 1025 * a.T * a.B
    print('gl_iter: {:.1f} us/iteration over {} iterations -> {:.0f} GB/s algorithmic; gl_final {:.1f} us{}'.format(
        per * 1e3, n, alg / (per * 1e-3) / 1e9, 1e3 * msf / max(1, nf), '  ({} workgroups)'.format(a.workers) if a.workers else ''))
    if not a.nocheck:
        assert np.isfinite(wav.to_host()).all()


if __name__ == '__main__':
    main()
