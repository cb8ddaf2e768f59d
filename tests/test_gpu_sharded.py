"""GPU: the sharded path of SURVEY.md 8(e) run for real -- two ranks (gloo, sharing the box's one GPU), one weight
broadcast, contiguous 64-utterance shards of a 128-utterance batch -- against single-process runs:

  * rank r's mel and linear spectrograms equal rows [64 r, 64 r + 64) of the single-process 128-utterance run
    BIT FOR BIT (an utterance's spectrograms do not depend on the batch it is in: what makes sharding exact);
  * rank r's waveforms equal the single-process run of the same 64 utterances bit for bit (the Griffin-Lim item cut,
    hence the overlap-add order, depends on the batch size and on whether the call is pipelined -- a rank's first call
    of a shape is not, so the single-process runs are made with the call pipeline off)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, pkg

pytestmark = pytest.mark.gpu

B, TS, N_STEPS, N_ITER = 64, 150, 200, 3


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_equal_the_single_process_runs(tmp_path, engine, hparams):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import shard_worker
    world = 2
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'shard_worker.py'), str(tmp_path), str(B),
                                       str(TS), str(N_STEPS), str(N_ITER)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(o[-2000:] for o in outs)

    T, F = N_STEPS * hparams.reduction, 1 + hparams.n_fft // 2
    ids, init = shard_worker.batch_inputs(world * B, TS, F, T)
    engine.set_option('pipeline', 0)
    try:
        _compare(engine, tmp_path, world, ids, init)
    finally:
        engine.set_option('pipeline', 1)


def _compare(engine, tmp_path, world, ids, init):
    full = engine.synthesize(ids, N_STEPS, 6.02, 99.89, 1.3, N_ITER, 1102, 275, init_phase=init, peak_normalize=True,
                             want_mel=True, want_linear=True)
    full_mel, full_lin = full['mel'].to_host(), full['linear'].to_host()
    for d in full.values():
        if d is not None:
            d.free()
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), 'rank{}.npz'.format(r)))
        lo, hi = int(got['lo']), int(got['hi'])
        assert (lo, hi) == (B * r, B * r + B)
        assert np.array_equal(got['mel'], full_mel[lo:hi]), 'rank {}: mel differs from the 128-utterance run'.format(r)
        assert np.array_equal(got['linear'], full_lin[lo:hi]), 'rank {}: linear differs from the 128-utterance run'.format(r)
        alone = engine.synthesize(ids[lo:hi], N_STEPS, 6.02, 99.89, 1.3, N_ITER, 1102, 275, init_phase=init[lo:hi],
                                  peak_normalize=True)
        assert np.array_equal(got['wav'], alone['wav'].to_host()), 'rank {}: waveforms differ from the same-size single run'.format(r)
        alone['wav'].free()
        assert np.isfinite(got['wav']).all() and np.abs(got['wav']).max() <= 1.0
