"""Test helper: writes a TensorFlow V2 checkpoint (tensor bundle) with the standard library.

Follows the published formats -- LevelDB table format (leveldb/doc/table_format.md: prefix-compressed
entries, restart array, 5-byte block trailer with masked CRC32C, 48-byte footer) and
tensorflow/core/protobuf/tensor_bundle.proto -- so that the importer can be exercised without
TensorFlow.  Only used by tests."""
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}


def _crc32c_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_TAB = _crc32c_table()


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _TAB[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(num, wt, payload):
    return varint((num << 3) | wt) + payload


def _shape_proto(shape):
    return b''.join(_field(2, 2, varint(len(d)) + d) for d in (_field(1, 0, varint(s)) for s in shape))


def _block(entries, restart_interval=16):
    buf = bytearray()
    restarts = []
    last = b''
    for i, (k, v) in enumerate(entries):
        if i % restart_interval == 0:
            restarts.append(len(buf))
            shared = 0
        else:
            shared = 0
            while shared < min(len(last), len(k)) and last[shared] == k[shared]:
                shared += 1
        buf += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        buf += struct.pack('<I', r)
    buf += struct.pack('<I', len(restarts))
    return bytes(buf)


def mask(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def write_tensor_bundle(prefix, tensors, block_entries=7, num_shards=1, restart_interval=16, crc_fn=None):
    """tensors: {name: ndarray}.  Writes <prefix>.index and <prefix>.data-0000k-of-0000n (tensors dealt round-robin over
    `num_shards` data files, as a sharded Saver writes them).  Every entry carries the masked CRC-32C of ALL its bytes
    (tensor_bundle.cc BundleWriter::Add).  crc_fn: a faster crc32c for large tensors (the byte loop here takes ~0.5 s per
    megabyte); the tests pass the importer's only after checking it against this module's on the same bytes."""
    crc_fn = crc_fn or crc32c
    names = sorted(tensors)
    data = [bytearray() for _ in range(num_shards)]
    kv = [(b'', _field(1, 0, varint(num_shards)) + _field(2, 0, varint(0)) + _field(3, 2, varint(2) + _field(1, 0, varint(1))))]
    for i, n in enumerate(names):
        a = np.asarray(tensors[n])   # (ascontiguousarray would turn a scalar into shape (1,))
        raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
        sid = i % num_shards
        entry = (_field(1, 0, varint(_DT[a.dtype])) + _field(2, 2, varint(len(_shape_proto(a.shape))) + _shape_proto(a.shape)) +
                 _field(3, 0, varint(sid)) + _field(4, 0, varint(len(data[sid]))) + _field(5, 0, varint(len(raw))) +
                 _field(6, 5, struct.pack('<I', mask(crc_fn(raw)))))
        kv.append((n.encode(), entry))
        data[sid] += raw
    for sid in range(num_shards):
        with open('{}.data-{:05d}-of-{:05d}'.format(prefix, sid, num_shards), 'wb') as f:
            f.write(bytes(data[sid]))
    out = bytearray()
    index_entries = []
    for i in range(0, len(kv), block_entries):
        chunk = kv[i:i + block_entries]
        blk = _block(chunk, restart_interval)
        off = len(out)
        out += blk + b'\x00' + struct.pack('<I', masked_crc(blk + b'\x00'))
        index_entries.append((chunk[-1][0], varint(off) + varint(len(blk))))
    meta = _block([])
    meta_off = len(out)
    out += meta + b'\x00' + struct.pack('<I', masked_crc(meta + b'\x00'))
    idx = _block(index_entries, restart_interval=1)
    idx_off = len(out)
    out += idx + b'\x00' + struct.pack('<I', masked_crc(idx + b'\x00'))
    footer = varint(meta_off) + varint(len(meta)) + varint(idx_off) + varint(len(idx))
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC)
    out += footer
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out))
