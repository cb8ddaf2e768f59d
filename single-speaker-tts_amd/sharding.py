"""Multi-GPU plumbing: utterance sharding and the one-off weight broadcast.

The inference path has no cross-utterance dependency (batch-norm uses frozen statistics,
attention and softmax are per row), so a batch shards into contiguous utterance ranges, one
process per GPU, and the only collective is ONE broadcast of the flat float32 weight blob
(6,855,713 parameters = 27.4 MB) from rank 0 at start-up -- RCCL over xGMI when the process
group backend is "nccl", gloo on CPU in the tests.  Nothing is exchanged afterwards.

Padding is part of the reference's semantics (no sequence masking: the backward GRU starts in
the padding and the attention softmax covers it), so every shard must be padded to the GLOBAL
maximum sentence length to reproduce the single-batch result: see ``pad_batch``.
"""
import numpy as np


def shard_range(n_items, world_size, rank):
    """Contiguous [lo, hi) slice of ``n_items`` utterances owned by ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pad_batch(id_sequences, pad_token=0, max_len=None):
    """Pad id sequences to one length (reference tacotron/inference.py:22-27,152-156)."""
    max_len = max_len if max_len is not None else max(len(s) for s in id_sequences)
    out = np.full((len(id_sequences), max_len), pad_token, dtype=np.int32)
    for i, s in enumerate(id_sequences):
        out[i, :len(s)] = np.asarray(s, dtype=np.int32)
    return out


def broadcast_blob(blob, src=0, device='cpu'):
    """Broadcast the flat float32 weight blob from ``src`` over the default process group.

    ``blob`` must have the same length on every rank (contents matter only on ``src``).
    Returns a host numpy array."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.float32)).to(device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def gather_host(array, dst=0):
    """Gather per-rank host arrays (first axis = utterances) on ``dst``; used by tests and the
    CLI, never inside a timed region."""
    import torch.distributed as dist
    world = dist.get_world_size()
    objs = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(np.asarray(array), objs, dst=dst)
    if dist.get_rank() == dst:
        return np.concatenate(objs, axis=0)
    return None
