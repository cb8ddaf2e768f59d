"""Helper of tests/test_gpu_sharded.py: ONE rank of the sharded path (bench.py's N > 1 structure with the outputs kept).

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment;  argv: out_dir B_per_rank Ts n_steps n_iter

rank 0 generates the weights, ONE broadcast (gloo here: the ranks share the box's only GPU; RCCL wants a device per
rank), every rank synthesizes its contiguous utterance shard of the world * B batch and writes mel / linear / wav."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def batch_inputs(n_utts, Ts, F, T):
    """ids and initial phases of the WHOLE batch, by global utterance index."""
    import bench
    ids = bench.synthetic_ids(n_utts, Ts, 1234)
    init = np.stack([np.random.default_rng(1000 + u).random((F, T), dtype=np.float32) for u in range(n_utts)])
    return ids, init


def main():
    out_dir, B, Ts, n_steps, n_iter = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sstts = importlib.import_module('single-speaker-tts_amd')
    P = importlib.import_module('single-speaker-tts_amd.tacotron.params')
    Wm = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
    shard = importlib.import_module('single-speaker-tts_amd.sharding')
    hp = P.ModelParams()
    blob = Wm.pack_blob(Wm.synthetic_weights(0, hp), hp) if rank == 0 else np.zeros(Wm.n_parameters(hp), np.float32)
    blob = shard.broadcast_blob(blob, src=0, device='cpu')
    eng = sstts.Engine(hp, device_id=0)
    eng.load_weights_blob(blob)
    T, F = n_steps * hp.reduction, 1 + hp.n_fft // 2
    ids, init = batch_inputs(world * B, Ts, F, T)
    lo, hi = shard.shard_range(world * B, world, rank)
    assert hi - lo == B
    out = eng.synthesize(ids[lo:hi], n_steps, 6.02, 99.89, 1.3, n_iter, 1102, 275, init_phase=init[lo:hi],
                         peak_normalize=True, want_mel=True, want_linear=True)
    np.savez(os.path.join(out_dir, 'rank{}.npz'.format(rank)), lo=lo, hi=hi, mel=out['mel'].to_host(),
             linear=out['linear'].to_host(), wav=out['wav'].to_host())
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
