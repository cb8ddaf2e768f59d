"""TensorFlow checkpoint importer without TensorFlow (host logic).

The reference restores its weights with ``tf.train.Saver().restore(session, checkpoint_file)``
where ``checkpoint_file`` is either an explicit prefix or ``tf.train.latest_checkpoint(dir)``
(tacotron/inference.py:44-55, 71).  TF 1.8 writes "V2" checkpoints = a *tensor bundle*:

  ``<prefix>.index``                 an SSTable (LevelDB table format, uncompressed blocks) mapping
                                     ``""`` -> BundleHeaderProto and every variable name ->
                                     BundleEntryProto (dtype, shape, shard_id, offset, size, crc32c)
  ``<prefix>.data-0000k-of-0000n``   the raw little-endian tensor bytes
  ``checkpoint``                     a text CheckpointState: ``model_checkpoint_path: "<prefix>"``

This module parses those three files with the standard library + numpy and maps the variables onto
the manifest of :mod:`.weights`.  Like ``Saver.restore`` it VERIFIES what it reads: the masked CRC-32C of
every table block (LevelDB block trailer, TF table/format.cc ReadBlock) and of every tensor's bytes
(BundleEntryProto.crc32c, TF tensor_bundle.cc BundleReader::GetValue); a mismatch raises
:class:`ChecksumError` -- a corrupt checkpoint must not load silently.  No checkpoint ships with the reference (README.md:357-361), so the
parser is validated against a writer of the same published format in the tests, not against a file
produced by TensorFlow.
"""
import os
import re
import struct

import numpy as np

from .params import ModelParams
from .weights import manifest

TABLE_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}   # tensorflow/core/framework/types.proto


class ChecksumError(ValueError):
    """A stored CRC-32C does not match the bytes read (TensorFlow: DataLoss 'Checksum does not match')."""


# ------------------------------------------------------------------------------------------ CRC-32C (Castagnoli)
def _crc_table():
    t = np.arange(256, dtype=np.uint32)
    for _ in range(8):
        t = np.where(t & 1, (t >> 1) ^ np.uint32(0x82F63B78), t >> 1).astype(np.uint32)
    return t


_CRC_TAB = _crc_table()
_CRC_TAB_LIST = [int(v) for v in _CRC_TAB]


def _crc_raw(data, reg):
    """The CRC register after `data` (bytes-like), starting from `reg`; no pre- / post-conditioning."""
    tab = _CRC_TAB_LIST
    for b in data:
        reg = tab[(reg ^ b) & 0xFF] ^ (reg >> 8)
    return reg


def _gf2_times(mat, vec):
    out = 0
    i = 0
    while vec:
        if vec & 1:
            out ^= mat[i]
        vec >>= 1
        i += 1
    return out


def _zero_shift_matrix(n_bytes):
    """32x32 GF(2) matrix (as 32 column words) that advances the CRC register through n_bytes zero bytes."""
    m = [_crc_raw(b'\0', 1 << i) for i in range(32)]           # one zero byte
    result = [1 << i for i in range(32)]                       # identity
    while n_bytes:
        if n_bytes & 1:
            result = [_gf2_times(m, c) for c in result]
        m = [_gf2_times(m, c) for c in m]
        n_bytes >>= 1
    return result


def crc32c(data):
    """CRC-32C of a bytes-like object (RFC 3720 check value: crc32c(b'123456789') == 0xE3069283).

    A checkpoint's tensors are tens of megabytes and the register recurrence is byte-serial, so long inputs are cut
    into K equal segments whose registers advance in lockstep as one numpy vector (the CRC is linear over GF(2):
    register(A || B, r) = register(B, 0) xor shift_{|B|}(register(A, r))), and the K partial registers are folded
    with the zero-shift matrix of one segment length."""
    buf = np.frombuffer(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview, np.ndarray)) else data, dtype=np.uint8)
    n = buf.size
    if n < 1 << 14:
        return _crc_raw(buf.tolist(), 0xFFFFFFFF) ^ 0xFFFFFFFF
    # init 0xFFFFFFFF == init 0 on the message with its first four bytes inverted; leading zero bytes leave a zero
    # register untouched, so the message may be padded IN FRONT to a whole number of segments
    K = 4096 if n >= 1 << 20 else 256
    seg = -(-n // K)
    m = np.zeros(K * seg, dtype=np.uint8)
    m[K * seg - n:] = buf
    m[K * seg - n:K * seg - n + 4] ^= 0xFF
    cols = m.reshape(K, seg).T.astype(np.uint32)               # cols[i] = byte i of every segment
    reg = np.zeros(K, dtype=np.uint32)
    for i in range(seg):
        reg = _CRC_TAB[(reg ^ cols[i]) & 0xFF] ^ (reg >> 8)
    shift = _zero_shift_matrix(seg)
    r = 0
    for part in reg.tolist():
        r = _gf2_times(shift, r) ^ part
    return r ^ 0xFFFFFFFF


def unmask_crc(masked):
    """Inverse of TF / LevelDB crc32c::Mask: rotate right by 15 bits and add 0xa282ead8."""
    rot = (masked - 0xa282ead8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------ varints / protobuf
def _varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _proto_fields(buf):
    """Yield (field_number, wire_type, value) of one protobuf message (value: int or bytes)."""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type {}'.format(wt))
        yield field, wt, val


def _parse_shape(buf):
    dims = []
    for f, _wt, v in _proto_fields(buf):
        if f == 2:                                   # TensorShapeProto.dim
            size = 0
            for f2, _w2, v2 in _proto_fields(v):
                if f2 == 1:
                    size = v2
            dims.append(int(size))
    return tuple(dims)


def _parse_bundle_entry(buf):
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for f, _wt, v in _proto_fields(buf):
        if f == 1:
            e['dtype'] = v
        elif f == 2:
            e['shape'] = _parse_shape(v)
        elif f == 3:
            e['shard_id'] = v
        elif f == 4:
            e['offset'] = v
        elif f == 5:
            e['size'] = v
        elif f == 6:
            e['crc32c'] = struct.unpack('<I', v)[0]
        elif f == 7:
            e['sliced'] = True
    return e


# ------------------------------------------------------------------------------------------ SSTable
def _block_handle(buf, pos):
    off, pos = _varint(buf, pos)
    size, pos = _varint(buf, pos)
    return off, size, pos


def _read_block(data, off, size, path='table'):
    if off + size + 5 > len(data):
        raise ValueError('{}: block [{}, +{}) runs past the end of the file'.format(path, off, size))
    block = data[off:off + size]
    ctype = data[off + size]
    stored = unmask_crc(struct.unpack('<I', data[off + size + 1:off + size + 5])[0])
    if crc32c(data[off:off + size + 1]) != stored:            # contents + type byte (table_format.md, block trailer)
        raise ChecksumError('{}: block checksum mismatch at offset {}'.format(path, off))
    if ctype != 0:
        raise NotImplementedError('compressed SSTable blocks are not supported (TF writes the bundle index uncompressed)')
    n_restarts = struct.unpack('<I', block[-4:])[0]
    limit = len(block) - 4 - 4 * n_restarts
    entries = []
    pos = 0
    key = b''
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        entries.append((key, block[pos:pos + vlen]))
        pos += vlen
    return entries


def read_table(path):
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 48 or struct.unpack('<Q', data[-8:])[0] != TABLE_MAGIC:
        raise ValueError('{} is not an SSTable (bad magic)'.format(path))
    footer = data[-48:]
    _mo, _ms, pos = _block_handle(footer, 0)
    io, isz, _ = _block_handle(footer, pos)
    out = []
    for _key, handle in _read_block(data, io, isz, path):
        bo, bs, _ = _block_handle(handle, 0)
        out.extend(_read_block(data, bo, bs, path))
    return out


# ------------------------------------------------------------------------------------------ tensor bundle
def read_tensor_bundle(prefix, verify=True):
    """{variable name: ndarray} of a TF V2 checkpoint ``prefix`` (no TensorFlow needed).  ``verify``: check every
    tensor's bytes against the masked CRC-32C its index entry stores (always on in load_checkpoint)."""
    entries = read_table(prefix + '.index')
    num_shards = 1
    tensors = {}
    metas = []
    for key, val in entries:
        if key == b'':
            for f, _wt, v in _proto_fields(val):      # BundleHeaderProto
                if f == 1:
                    num_shards = v
                elif f == 2 and v != 0:
                    raise NotImplementedError('big-endian tensor bundles are not supported')
            continue
        metas.append((key.decode('utf-8'), _parse_bundle_entry(val)))
    shards = {}
    for name, e in metas:
        if e['sliced']:
            raise NotImplementedError('partitioned variable {} (tensor slices) is not supported'.format(name))
        if e['dtype'] not in _DTYPES:
            continue                                   # strings etc. are of no use here
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = np.memmap('{}.data-{:05d}-of-{:05d}'.format(prefix, sid, num_shards), dtype=np.uint8, mode='r')
        dt = np.dtype(_DTYPES[e['dtype']]).newbyteorder('<')
        if e['offset'] + e['size'] > shards[sid].size:
            raise ValueError('{}: tensor {} runs past the end of its data shard'.format(prefix, name))
        raw = shards[sid][e['offset']:e['offset'] + e['size']].tobytes()
        if verify and e['crc32c'] is not None and crc32c(raw) != unmask_crc(e['crc32c']):
            raise ChecksumError('{}: checksum of tensor {} does not match its bytes'.format(prefix, name))
        arr = np.frombuffer(raw, dtype=dt).reshape(e['shape'])
        tensors[name] = arr
    return tensors


def latest_checkpoint(checkpoint_dir):
    """tf.train.latest_checkpoint: the prefix named by ``model_checkpoint_path`` in ``<dir>/checkpoint``
    (relative paths are resolved against the directory); None if there is no usable state."""
    state = os.path.join(checkpoint_dir, 'checkpoint')
    if not os.path.isfile(state):
        return None
    with open(state) as f:
        m = re.search(r'^model_checkpoint_path:\s*"([^"]*)"', f.read(), flags=re.M)
    if not m:
        return None
    path = m.group(1)
    if not os.path.isabs(path):
        path = os.path.join(checkpoint_dir, path)
    return path if os.path.exists(path + '.index') else None


_SLOT = re.compile(r'/(Adam(_\d+)?|Momentum|RMSProp(_\d+)?|ExponentialMovingAverage)$')

# ------------------------------------------------------------------------------------------ CudnnGRU parameters
# With force_cudnn=True (the shipped default, params/model.py:51) the CBHG bi-GRUs are tf.contrib.cudnn_rnn.CudnnGRU
# layers (reference tacotron/layers.py:560-577) whose trainable variable is ONE opaque buffer.  TF 1.8 saves it
# through CudnnGRUSaveable in the "canonical" form of CudnnCompatibleGRUCell [TF-1.8 contrib/cudnn_rnn/python/ops/
# cudnn_rnn_ops.py]: per direction ``<scope>/stack_bidirectional_rnn/cell_0/bidirectional_rnn/{fw,bw}/
# cudnn_compatible_gru_cell/{gates/{kernel,bias}, candidate/{input,hidden}_projection/{kernel,bias}}``.  Both forms
# are accepted: canonical names are renamed onto the manifest, a raw opaque buffer is unpacked first.
_CUDNN_CANON = re.compile(r'^(?P<scope>.*?)/(?:gru/)?(?:cudnn_gru/)?stack_bidirectional_rnn/cell_0/bidirectional_rnn/'
                          r'(?P<d>fw|bw)/cudnn_compatible_gru_cell/(?P<rest>.+)$')
_CUDNN_OPAQUE = re.compile(r'^(?P<scope>.*?)/(?:gru/)?(?:cudnn_gru/)?opaque_kernel$')


def cudnn_gru_opaque_to_canonical(opaque, input_size, num_units, bidirectional=True):
    """Unpack a one-layer cuDNN GRU parameter buffer into CudnnCompatibleGRUCell tensors.

    cuDNN's layout (what cudnnGetRNNLinLayerMatrixParams walks, and what TF's CudnnGRUSaveable assumes): the
    weight matrices of every pseudo-layer first (forward, then backward), then the biases of every pseudo-layer;
    inside a pseudo-layer the input matrices W_r, W_u, W_c (each (num_units, input_size)) come before the
    recurrent matrices R_r, R_u, R_c (each (num_units, num_units)), the biases likewise bW_r, bW_u, bW_c, bR_r,
    bR_u, bR_c.  Gate order r (reset), u (update, cuDNN's z), c (candidate, cuDNN's h).

    TF canonical form: gates/kernel = [[W_r; R_r]^T | [W_u; R_u]^T] with rows ordered [input ; state], gates/bias
    = bW + bR; candidate/input_projection = (W_c^T, bW_c); candidate/hidden_projection = (R_c^T, bR_c) -- the
    CudnnCompatibleGRUCell formulation c = tanh(x W_c + b_Wc + r * (h R_c + b_Rc)).

    Returns {'fw' | 'bw': {suffix: array}}."""
    opaque = np.asarray(opaque, dtype=np.float32).reshape(-1)
    dirs = ('fw', 'bw') if bidirectional else ('fw',)
    n_w = 3 * num_units * input_size + 3 * num_units * num_units
    n_b = 6 * num_units
    need = len(dirs) * (n_w + n_b)
    if opaque.size != need:   # a buffer of other dimensions would be sliced into wrong matrices without any error
        raise ValueError('opaque CudnnGRU buffer has {} floats, a {}-directional {}->{} layer has {}'.format(
            opaque.size, len(dirs), input_size, num_units, need))
    out = {}
    for di, d in enumerate(dirs):
        w = opaque[di * n_w:(di + 1) * n_w]
        b = opaque[len(dirs) * n_w + di * n_b:len(dirs) * n_w + (di + 1) * n_b]
        wi = w[:3 * num_units * input_size].reshape(3, num_units, input_size)
        wr = w[3 * num_units * input_size:].reshape(3, num_units, num_units)
        bw, br = b[:3 * num_units].reshape(3, num_units), b[3 * num_units:].reshape(3, num_units)
        out[d] = {
            'gates/kernel': np.concatenate([np.concatenate([wi[0].T, wr[0].T], 0), np.concatenate([wi[1].T, wr[1].T], 0)], 1),
            'gates/bias': np.concatenate([bw[0] + br[0], bw[1] + br[1]]),
            'candidate/input_projection/kernel': np.ascontiguousarray(wi[2].T),
            'candidate/input_projection/bias': bw[2].copy(),
            'candidate/hidden_projection/kernel': np.ascontiguousarray(wr[2].T),
            'candidate/hidden_projection/bias': br[2].copy(),
        }
    return out


def expand_cudnn_gru(tensors, hparams=None):
    """CudnnGRU variables of a checkpoint -> manifest names (``<scope>/gru/{fw,bw}/gru_cell_{fw,bw}/...``)."""
    hp = hparams or ModelParams()
    out = dict(tensors)
    for name, arr in tensors.items():
        m = _CUDNN_CANON.match(name)
        if m:
            out['{}/gru/{}/gru_cell_{}/{}'.format(m.group('scope'), m.group('d'), m.group('d'), m.group('rest'))] = arr
            continue
        m = _CUDNN_OPAQUE.match(name)
        if m:
            # input of the bi-GRU = the highway width, units = n_gru_units (reference layers.py:555-566) -- of the
            # CBHG the buffer belongs to: the post-processing net has its own sizes (model.py:336-352)
            scope = m.group('scope')
            cb = hp.post if (scope.startswith('post') or '/post' in scope) else hp.encoder
            canon = cudnn_gru_opaque_to_canonical(arr, cb.n_highway_units, cb.n_gru_units, True)
            for d, parts in canon.items():
                for suffix, v in parts.items():
                    out['{}/gru/{}/gru_cell_{}/{}'.format(m.group('scope'), d, d, suffix)] = v
    return out



def select_model_variables(tensors, hparams=None, aliases=None):
    """Pick the manifest's variables out of a checkpoint's tensors.

    Optimizer slots (``.../Adam``, ``.../Adam_1``), ``global_step`` and the ``beta?_power``
    accumulators that training checkpoints carry are ignored.  ``aliases`` ({checkpoint name:
    manifest name}) lets a caller adapt scope spellings; shapes are checked against the manifest."""
    hp = hparams or ModelParams()
    m = manifest(hp)
    aliases = aliases or {}
    out = {}
    if hp.force_cudnn:
        tensors = expand_cudnn_gru(tensors, hp)
    for name, arr in tensors.items():
        if _SLOT.search(name) or name in ('global_step', 'beta1_power', 'beta2_power'):
            continue
        key = aliases.get(name, name)
        if key in m:
            if tuple(arr.shape) != tuple(m[key]):
                raise ValueError('checkpoint variable {} has shape {}, the model needs {}'.format(name, arr.shape, m[key]))
            out[key] = np.ascontiguousarray(arr, dtype=np.float32)
    missing = [k for k in m if k not in out]
    if missing:
        hint = ''
        if any('cudnn' in n.lower() or 'opaque_kernel' in n for n in tensors) and not hp.force_cudnn:
            hint = ' (the checkpoint holds CudnnGRU parameters: load it with hparams.force_cudnn = True)'
        raise KeyError('checkpoint lacks {} model variables, e.g. {}{}'.format(len(missing), missing[:3], hint))
    return out


def load_checkpoint(path, hparams=None, aliases=None):
    """``path``: a checkpoint prefix or a run directory (then its latest checkpoint, as the reference
    resolves it at tacotron/inference.py:49-53).  Returns {manifest name: float32 array}."""
    prefix = path
    if os.path.isdir(path):
        prefix = latest_checkpoint(path)
        if prefix is None:
            raise FileNotFoundError('no checkpoint state in {}'.format(path))
    return select_model_variables(read_tensor_bundle(prefix), hparams, aliases)
