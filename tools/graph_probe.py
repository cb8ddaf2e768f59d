#!/usr/bin/env python3
"""Probe for the hipGraph wrong-result problem of DESIGN.md section 8 (round 5: "replays of the decoder graph return wrong mel
spectrograms in a long-lived process"; round 6: it is the HIP runtime the process runs on -- PyTorch's bundled libamdhip64 of
ROCm 7.0, which a process that imports torch before the library gets instead of /opt/rocm's 7.2).

    python tools/graph_probe.py [--torch] [--forms 00220220] [--reps 5] [--steps 6]

The call sequence of tests/test_gpu_full_size.py::test_decoder_form_switched_between_pipelined_calls with the decoder graph
ON: serial calls against pipelined calls, call by call; prints which runtime the process maps and what is wrong where."""
import argparse
import importlib
import os
import sys

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument('--torch', action='store_true', help='import torch first (its bundled HIP runtime then serves the library)')
ap.add_argument('--forms', default='00220220', help='persistent_decoder option per call (0 = launch-per-layer, i.e. the graph)')
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--steps', type=int, default=6)
ap.add_argument('--graph', type=int, default=1, help='0: launches enqueued directly, 1: the graph, 2: the graph even on a runtime the library refuses it on (set by --torch)')
ap.add_argument('--prelude', type=int, default=1, help='calls of another shape first (what tests/test_gpu_full_size.py::test_end_to_end_synthesize_matches_staged makes): 1 = synthesize + stand-alone decoder, 2 = synthesize only, 3 = stand-alone decoder only, 0 = none')
a = ap.parse_args()
if a.torch:
    import torch  # noqa: F401
    a.graph = 2 if a.graph else 0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('single-speaker-tts_amd')
P = importlib.import_module('single-speaker-tts_amd.tacotron.params')
W = importlib.import_module('single-speaker-tts_amd.tacotron.weights')


def bench_ids(B, Ts, seed):
    rng = np.random.default_rng(seed)
    ids = np.zeros((B, Ts), np.int32)
    for b in range(B):
        L = int(np.clip(round(rng.normal(100, 30)), 20, Ts - 1))
        ids[b, :L] = rng.integers(2, 39, L)
        ids[b, L] = 1
    return ids


hp = P.ModelParams()
eng = pkg.Engine(hp, device_id=0)
eng.load_weights(W.synthetic_weights(0, hp))
eng.set_option('debug_hooks', 1)
print('runtime:', sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}))
WIN, HOP = 1102, 275
batches = [bench_ids(5, 21, 170 + i) for i in range(len(a.forms))]
forms = [int(c) for c in a.forms]


def run(pipeline, graph):
    eng.set_option('pipeline', pipeline)
    eng.set_option('use_graph', graph)
    dev = [eng.to_device(b) for b in batches]
    outs = []
    for i, d in enumerate(dev):
        eng.set_option('persistent_decoder', forms[i])
        outs.append(eng.synthesize(d, a.steps, 6.02, 99.89, 1.3, 4, WIN, HOP, seed=700 + i, want_mel=True, want_linear=True))
    eng.synchronize()
    return [{k: v.to_host() for k, v in o.items() if v is not None} for o in outs]


if a.prelude:
    eng.set_option('use_graph', a.graph)
    ids3 = bench_ids(3, 40, 5)
    if a.prelude in (1, 2):
        eng.synthesize(ids3, 10, 6.02, 99.89, 1.3, 4, WIN, HOP, seed=3, peak_normalize=True, want_mel=True, want_alignments=True, want_linear=True)
    if a.prelude in (1, 3):
        mem = eng.encoder_forward(ids3)
        eng.decoder_forward(mem, 10)
    eng.synchronize()
ref = run(0, 0)          # serial, every launch enqueued directly: the reference
run(1, a.graph)          # shapes known
for name, pipeline in (('serial', 0), ('pipelined', 1)):
    nbad = 0
    first = None
    for rep in range(a.reps):
        got = run(pipeline, a.graph)
        bad = [(i, forms[i], float(np.abs(g['mel']).max()), float(np.abs(g['mel'] - r['mel']).max()))
               for i, (r, g) in enumerate(zip(ref, got)) if forms[i] == 0 and not np.array_equal(r['mel'], g['mel'])]
        if bad:
            nbad += 1
            first = first or (rep, bad)
    print('{} with use_graph={}: {} of {} repetitions have wrong graph calls; first: {}'.format(name, 1 if a.graph else 0, nbad, a.reps, first), flush=True)
