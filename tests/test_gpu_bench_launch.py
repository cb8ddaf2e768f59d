"""GPU: the real `bench.py --gpus 2` path, launched without torchrun.  With one GPU in the box the two ranks
share it (SSTTS_DIST_BACKEND=gloo: RCCL needs one device per rank); what is checked is the launcher, the weight
broadcast, the per-rank shards and the max-over-ranks timing -- not a scaling number."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_gpus2_self_launch_gloo():
    env = dict(os.environ, SSTTS_DIST_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                          '--no-cpu-baseline'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    rec = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert rec['n_gpus'] == 2 and rec['world_size_seen'] == 2
    assert rec['config']['global_batch'] == 128 and rec['scaling'] == 'weak'
    assert rec['value'] > 0 and rec['steps'] == 2
