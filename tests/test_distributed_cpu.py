"""CPU: the N > 1 path (weight broadcast + utterance shards) with world_size 2 over gloo."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT, PKG


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    import importlib
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    S = importlib.import_module(PKG + '.sharding')
    W = importlib.import_module(PKG + '.tacotron.weights')
    n = W.n_parameters()
    if rank == 0:
        blob = W.pack_blob(W.synthetic_weights(0))
    else:
        blob = np.zeros(n, np.float32)
    got = S.broadcast_blob(blob, src=0, device='cpu')
    # every rank ends up with the same named tensors
    w = W.unpack_blob(got)
    digest = float(sum(float(np.abs(v).sum()) for v in w.values()))
    # utterance shards of a 9-utterance batch padded to the GLOBAL length
    seqs = [list(range(2, 2 + L)) + [1] for L in (3, 9, 5, 4, 8, 2, 7, 6, 1)]
    batch = S.pad_batch(seqs)
    lo, hi = S.shard_range(len(seqs), world, rank)
    mine = batch[lo:hi]
    assert mine.shape[1] == batch.shape[1] == 10
    back = S.gather_host(mine, dst=0)
    np.save(os.path.join(tmp, 'digest{}.npy'.format(rank)), np.array([digest]))
    if rank == 0:
        assert np.array_equal(back, batch)
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_shards_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d0 = np.load(tmp_path / 'digest0.npy')[0]
    d1 = np.load(tmp_path / 'digest1.npy')[0]
    assert d0 == d1 and d0 > 0


def test_bench_self_launches_n_ranks(tmp_path):
    """`python bench.py --gpus 2` without torchrun: the parent spawns two children, which form a gloo group,
    broadcast the weight blob and report the world size they saw (--dist-selftest skips the GPU work)."""
    import json
    import subprocess
    env = dict(os.environ, SSTTS_DIST_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dist-selftest'], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith('{')][-1]
    rec = json.loads(line)
    assert rec['n_gpus'] == 2 and rec['world_size_seen'] == 2 and rec['ok'] is True

    # the N > 1 diagnostics of the JSON line (bench.rank_fields), rehearsed on stand-in values: per-rank times with the
    # straggler named, the first broadcast against the steady one, one device record per rank, the global padded length
    assert rec['rank_ms_per_step'] == [1.0, 2.0] and rec['straggler_rank'] == 1
    assert rec['rank_ms_per_step_min'] == 1.0 and rec['rank_ms_per_step_max'] == 2.0
    assert rec['weight_broadcast_ms'] > 0 and rec['weight_broadcast_ms_steady'] > 0 and rec['communicator_setup_ms'] > 0
    assert [d['uuid'] for d in rec['rank_devices']] == ['selftest-rank0', 'selftest-rank1']
    assert rec['padded_sentence_length'] == 150


def test_bench_under_torch_distributed_run():
    """The driver's own launch line for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`): bench.py must take RANK / LOCAL_RANK / WORLD_SIZE
    from the environment it is given instead of launching ranks of its own."""
    import json
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SSTTS_DIST_BACKEND='gloo')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', str(port),
                          os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dist-selftest'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['world_size_seen'] == 2 and rec['ok'] is True

    # the N > 1 diagnostics of the JSON line (bench.rank_fields), rehearsed on stand-in values: per-rank times with the
    # straggler named, the first broadcast against the steady one, one device record per rank, the global padded length
    assert rec['rank_ms_per_step'] == [1.0, 2.0] and rec['straggler_rank'] == 1
    assert rec['rank_ms_per_step_min'] == 1.0 and rec['rank_ms_per_step_max'] == 2.0
    assert rec['weight_broadcast_ms'] > 0 and rec['weight_broadcast_ms_steady'] > 0 and rec['communicator_setup_ms'] > 0
    assert [d['uuid'] for d in rec['rank_devices']] == ['selftest-rank0', 'selftest-rank1']
    assert rec['padded_sentence_length'] == 150


def _selftest_rank_env(cmd, env):
    import json
    import subprocess
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    rec = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert rec['ok'] is True and len(rec['rank_env']) == 2
    return rec['rank_env']


def test_both_launch_paths_give_a_rank_the_same_environment():
    """Round 5: what a rank process needs (HSA_ENABLE_IPC_MODE_LEGACY=0 -- RCCL's buffer sharing between the ranks of a
    node needs dmabuf IPC on this pool --, its share of the host's CPUs) is applied by the rank itself (bench.rank_env)
    before torch is imported, so a rank started by the driver's `python -m torch.distributed.run` line and one started by
    `python bench.py --gpus N` see the same thing -- also when the launcher's own environment does not carry the variable."""
    env = dict(os.environ, SSTTS_DIST_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY'):
        env.pop(k, None)
    bench = os.path.join(ROOT, 'bench.py')
    own = _selftest_rank_env([sys.executable, bench, '--gpus', '2', '--dist-selftest'], env)
    run = _selftest_rank_env([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                              '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), bench, '--gpus', '2',
                              '--dist-selftest'], env)
    assert own == run, (own, run)
    n_cpus = len(os.sched_getaffinity(0))
    for e in own:
        assert e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
        assert e['n_cpus_bound'] == (n_cpus // 2 if n_cpus >= 2 else n_cpus)
    # a value the launcher exported wins (setdefault), on both paths
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '1'
    kept = _selftest_rank_env([sys.executable, bench, '--gpus', '2', '--dist-selftest'], env)
    assert [e['HSA_ENABLE_IPC_MODE_LEGACY'] for e in kept] == ['1', '1']


def test_rank_cpu_share_partitions_the_cpus():
    import importlib.util
    spec = importlib.util.spec_from_file_location('_bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cpus = list(range(3, 35))
    shares = [bench.rank_cpu_share(cpus, r, 8) for r in range(8)]
    assert sorted(c for s in shares for c in s) == cpus and all(len(s) == 4 for s in shares)
    assert bench.rank_cpu_share(cpus, 0, 1) == cpus
    assert bench.rank_cpu_share([0, 1], 2, 4) == [0, 1]   # fewer CPUs than ranks: no binding
