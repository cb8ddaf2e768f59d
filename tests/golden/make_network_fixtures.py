#!/usr/bin/env python3
"""Generates tests/golden/network_small.npz and tests/golden/griffin_lim_small.npz.

The reference (TensorFlow 1.8 + librosa) cannot run anywhere in this build, so these vectors
come from the build's own float64 numpy restatement (oracle/) on seeded synthetic weights --
"parity unpinned" (see oracle/*.py headers and DESIGN.md).  They pin the ORACLE against silent
drift and give the GPU tests fixed inputs/outputs that do not depend on re-running the oracle.
"""
import hashlib
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    from oracle import audio_oracle as A
    from oracle import tacotron_oracle as O
    P = importlib.import_module('single-speaker-tts_amd.tacotron.params')
    W = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
    hp = P.ModelParams()
    w = W.synthetic_weights(0, hp)
    blob = W.pack_blob(w, hp)
    digest = hashlib.sha256(blob.tobytes()).hexdigest()
    rng = np.random.default_rng(2024)
    B, Ts, S = 2, 7, 3
    ids = rng.integers(2, 39, (B, Ts)).astype(np.int32)
    ids[0, 5:] = [1, 0]
    ids[1, 6] = 1
    out = O.tacotron_predict(ids, O.cast_weights(w, np.float64), hp, n_steps=S)
    np.savez_compressed(os.path.join(HERE, 'network_small.npz'), ids=ids, n_steps=S, weights_seed=0,
                        weights_sha256=digest, memory=out['memory'].astype(np.float32),
                        reduced_mel=out['reduced_mel'].astype(np.float32),
                        alignments=out['alignments'].astype(np.float32),
                        linear=out['linear'].astype(np.float32))
    # Griffin-Lim: 12 frames, 2 iterations, injected phases
    T = 12
    n = 275 * (T - 1)
    t = np.arange(n) / 22050.0
    y = (0.4 * np.sin(2 * np.pi * 330 * t) + 0.05 * rng.standard_normal(n)).astype(np.float32)
    mag = np.abs(A.stft(y, 2048, 275, 1102)).astype(np.float32)
    init = rng.random(mag.shape).astype(np.float32)
    hist = []
    wav, mse = A.griffin_lim_v2(mag, 1102, 275, 2048, 2, init_phase=init, history=hist)
    lin = (rng.random((T, 1025)) * 1.2 - 0.1).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'griffin_lim_small.npz'), mag=mag, init_phase=init, n_iter=2,
                        wav=wav.astype(np.float32), mse=np.float32(mse),
                        wav_n0=A.griffin_lim_v2(mag, 1102, 275, 2048, 0, init_phase=init)[0].astype(np.float32),
                        wav_n1=A.griffin_lim_v2(mag, 1102, 275, 2048, 1, init_phase=init)[0].astype(np.float32),
                        linear=lin, linear_mag=A.linear_to_magnitude(lin, 6.02, 99.89, 1.3).astype(np.float32),
                        peak_norm=A.peak_normalize(wav))
    print('weights sha256', digest)


if __name__ == '__main__':
    main()
