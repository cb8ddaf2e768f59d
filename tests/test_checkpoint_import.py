"""CPU: the TensorFlow-free checkpoint importer (tacotron/checkpoint.py) against a bundle written in the
published format by tests/tf_bundle_writer.py."""
import os

import numpy as np
import pytest

from conftest import pkg
from tf_bundle_writer import crc32c, masked_crc, write_tensor_bundle


def test_crc32c_known_answer():
    """Both CRC-32C implementations (the writer's byte loop, the importer's segment-parallel one) against the
    published vectors of RFC 3720 (B.4) and against each other across the importer's size thresholds."""
    C = pkg('tacotron.checkpoint')
    vectors = [(b'123456789', 0xE3069283), (bytes(32), 0x8A9136AA), (b'\xff' * 32, 0x62A8AB43),
               (bytes(range(32)), 0x46DD794E), (bytes(range(31, -1, -1)), 0x113FDB5C)]
    for data, want in vectors:
        assert crc32c(data) == want and C.crc32c(data) == want
    rng = np.random.default_rng(5)
    for n in (0, 1, 3, 4, 5, 1000, (1 << 14) - 1, 1 << 14, (1 << 14) + 1, 70001, (1 << 20) + 13):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert C.crc32c(d) == crc32c(d), n
        assert C.unmask_crc(masked_crc(d)) == crc32c(d)


def test_corrupt_checkpoints_are_refused(tmp_path):
    """What Saver.restore does with a damaged file (tacotron/inference.py:55,71 -> DataLoss): one flipped bit in a
    tensor's bytes or in an index block must raise, in whichever data shard it sits; restart arrays with several
    entries and several data shards parse."""
    C = pkg('tacotron.checkpoint')
    rng = np.random.default_rng(1)
    tensors = {'scope/v{:02d}/kernel'.format(i): rng.standard_normal((5, 3 + i)).astype(np.float32) for i in range(23)}
    tensors['global_step'] = np.array(7, dtype=np.int64)
    prefix = str(tmp_path / 'm')
    write_tensor_bundle(prefix, tensors, block_entries=9, num_shards=3, restart_interval=2)
    assert sorted(f.name for f in tmp_path.iterdir()) == ['m.data-00000-of-00003', 'm.data-00001-of-00003',
                                                          'm.data-00002-of-00003', 'm.index']
    got = C.read_tensor_bundle(prefix)
    assert set(got) == set(tensors) and all(np.array_equal(got[k], tensors[k]) for k in tensors)
    # a flipped bit in the second data shard
    shard = tmp_path / 'm.data-00001-of-00003'
    raw = bytearray(shard.read_bytes())
    raw[len(raw) // 2] ^= 0x10
    shard.write_bytes(bytes(raw))
    with pytest.raises(C.ChecksumError, match='tensor'):
        C.read_tensor_bundle(prefix)
    assert set(C.read_tensor_bundle(prefix, verify=False)) == set(tensors)      # (the escape hatch reads the damaged bytes)
    raw[len(raw) // 2] ^= 0x10
    shard.write_bytes(bytes(raw))
    C.read_tensor_bundle(prefix)
    # a flipped bit inside an index block (here: in a key of the first data block)
    index = tmp_path / 'm.index'
    good = index.read_bytes()
    bad = bytearray(good)
    bad[40] ^= 0x01
    index.write_bytes(bytes(bad))
    with pytest.raises(C.ChecksumError, match='block'):
        C.read_tensor_bundle(prefix)
    # a wrong stored tensor checksum with intact bytes
    index.write_bytes(good)
    entries = dict(C.read_table(prefix + '.index'))
    e = C._parse_bundle_entry(entries[b'scope/v03/kernel'])
    assert C.unmask_crc(e['crc32c']) == crc32c(tensors['scope/v03/kernel'].tobytes())


def test_bundle_round_trip(tmp_path):
    C = pkg('tacotron.checkpoint')
    rng = np.random.default_rng(0)
    tensors = {'a/b/kernel': rng.standard_normal((3, 5)).astype(np.float32),
               'a/b/bias': rng.standard_normal(5).astype(np.float32),
               'a/b/kernel/Adam': np.zeros((3, 5), np.float32),
               'global_step': np.array(510000, dtype=np.int64),
               'z/long/name/with/shared/prefix/one': np.arange(7, dtype=np.int32),
               'z/long/name/with/shared/prefix/two': rng.standard_normal((2, 2, 2)).astype(np.float64)}
    prefix = str(tmp_path / 'model.ckpt-510000')
    write_tensor_bundle(prefix, tensors, block_entries=3)       # several data blocks + prefix compression
    got = C.read_tensor_bundle(prefix)
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    with pytest.raises(ValueError):
        open(str(tmp_path / 'junk.index'), 'wb').write(b'0' * 64)
        C.read_table(str(tmp_path / 'junk.index'))


def test_latest_checkpoint_and_model_variable_selection(tmp_path, weights, hparams):
    C = pkg('tacotron.checkpoint')
    run = tmp_path / 'train'
    run.mkdir()
    assert C.latest_checkpoint(str(run)) is None
    ck = dict(weights)
    ck['global_step'] = np.array(215000, dtype=np.int64)
    ck['beta1_power'] = np.array(0.1, dtype=np.float32)
    ck['dense/kernel/Adam'] = np.zeros_like(weights['dense/kernel'])
    ck['dense/kernel/Adam_1'] = np.zeros_like(weights['dense/kernel'])
    write_tensor_bundle(str(run / 'model.ckpt-215000'), ck, block_entries=16, num_shards=2, crc_fn=C.crc32c)
    (run / 'checkpoint').write_text('model_checkpoint_path: "model.ckpt-215000"\n'
                                    'all_model_checkpoint_paths: "model.ckpt-210000"\n'
                                    'all_model_checkpoint_paths: "model.ckpt-215000"\n')
    assert C.latest_checkpoint(str(run)) == os.path.join(str(run), 'model.ckpt-215000')
    got = C.load_checkpoint(str(run), hparams)                  # directory -> latest, like inference.py:49-53
    assert set(got) == set(weights)
    assert all(np.array_equal(got[k], weights[k]) for k in weights)
    # a scope spelled differently in the checkpoint can be aliased; a missing variable is an error
    renamed = {('decoder/memory_layer/kernel' if k == 'decoder2/memory_layer/kernel' else k): v for k, v in weights.items()}
    write_tensor_bundle(str(tmp_path / 'other'), renamed, block_entries=16, crc_fn=C.crc32c)
    with pytest.raises(KeyError):
        C.load_checkpoint(str(tmp_path / 'other'), hparams)
    ok = C.load_checkpoint(str(tmp_path / 'other'), hparams, aliases={'decoder/memory_layer/kernel': 'decoder2/memory_layer/kernel'})
    assert np.array_equal(ok['decoder2/memory_layer/kernel'], weights['decoder2/memory_layer/kernel'])
    bad = dict(weights)
    bad['dense/bias'] = np.zeros(7, np.float32)
    write_tensor_bundle(str(tmp_path / 'bad'), bad, block_entries=16, crc_fn=C.crc32c)
    with pytest.raises(ValueError):
        C.load_checkpoint(str(tmp_path / 'bad'), hparams)


def _torch_gru_as_cudnn_opaque(gru):
    """The parameters of a one-layer torch.nn.GRU laid out as cuDNN's opaque buffer: weights of every direction
    first ([W_r W_u W_c | R_r R_u R_c], torch's gate order r, z, n is cuDNN's), then the biases."""
    ws, bs = [], []
    for suf in ('', '_reverse') if gru.bidirectional else ('',):
        ws += [getattr(gru, 'weight_ih_l0' + suf).detach().numpy().reshape(-1),
               getattr(gru, 'weight_hh_l0' + suf).detach().numpy().reshape(-1)]
        bs += [getattr(gru, 'bias_ih_l0' + suf).detach().numpy(), getattr(gru, 'bias_hh_l0' + suf).detach().numpy()]
    return np.concatenate(ws + bs).astype(np.float32)


def test_cudnn_opaque_buffer_unpacks_to_the_cell_torch_computes():
    """CudnnGRU's opaque parameters -> CudnnCompatibleGRUCell tensors (reference tacotron/layers.py:560-577,
    force_cudnn=True): checked against torch.nn.GRU, which runs the same cuDNN formulation from the same layout."""
    import torch
    from oracle import tacotron_oracle as O
    C = pkg('tacotron.checkpoint')
    torch.manual_seed(0)
    n_in, U, B, T = 6, 5, 3, 7
    gru = torch.nn.GRU(n_in, U, batch_first=True, bidirectional=True)
    canon = C.cudnn_gru_opaque_to_canonical(_torch_gru_as_cudnn_opaque(gru), n_in, U, True)
    w = {'gru/{}/gru_cell_{}/{}'.format(d, d, k): v.astype(np.float64) for d, parts in canon.items() for k, v in parts.items()}
    assert w['gru/fw/gru_cell_fw/gates/kernel'].shape == (n_in + U, 2 * U)
    x = np.random.default_rng(1).standard_normal((B, T, n_in))
    ref, _ = gru.double()(torch.tensor(x))
    assert np.allclose(O.bi_gru(x, w, 'gru', U, cudnn=True), ref.detach().numpy(), atol=1e-6)
    with pytest.raises(ValueError):
        C.cudnn_gru_opaque_to_canonical(np.zeros(10, np.float32), n_in, U, True)


def test_cudnn_checkpoints_load_in_both_saved_forms(tmp_path):
    """A force_cudnn=True checkpoint holds the CBHG bi-GRUs either as TF's canonical CudnnCompatibleGRUCell tensors
    (what CudnnGRUSaveable writes) or as the raw opaque buffer; both must give the manifest's variables."""
    import copy
    C = pkg('tacotron.checkpoint')
    W = pkg('tacotron.weights')
    hp = copy.deepcopy(pkg('tacotron.params').ModelParams())
    hp.force_cudnn = True
    weights = W.synthetic_weights(3, hp)
    U, H = hp.encoder.n_highway_units, hp.encoder.n_gru_units

    def gru_keys(scope, d):
        return [k for k in weights if k.startswith('{}/gru/{}/gru_cell_{}/'.format(scope, d, d))]

    # (1) canonical names
    canon = {}
    for k, v in weights.items():
        for scope in ('encoder', 'post_process'):
            for d in ('fw', 'bw'):
                pre = '{}/gru/{}/gru_cell_{}/'.format(scope, d, d)
                if k.startswith(pre):
                    k = '{}/gru/cudnn_gru/stack_bidirectional_rnn/cell_0/bidirectional_rnn/{}/cudnn_compatible_gru_cell/{}'.format(
                        scope, d, k[len(pre):])
        canon[k] = v
    assert not any('/gru_cell_fw/' in k and k.startswith('encoder/gru') for k in canon)
    write_tensor_bundle(str(tmp_path / 'canon'), canon, block_entries=16, crc_fn=C.crc32c)
    got = C.load_checkpoint(str(tmp_path / 'canon'), hp)
    assert set(got) == set(weights) and all(np.array_equal(got[k], weights[k]) for k in weights)
    # (2) opaque buffers, built by inverting the documented layout
    opaque = {k: v for k, v in weights.items() if '/gru/fw/gru_cell_fw/' not in k and '/gru/bw/gru_cell_bw/' not in k
              or k.startswith('decoder2')}
    for scope in ('encoder', 'post_process'):
        ws, bs = [], []
        for d in ('fw', 'bw'):
            pre = '{}/gru/{}/gru_cell_{}/'.format(scope, d, d)
            gk, gb = weights[pre + 'gates/kernel'], weights[pre + 'gates/bias']
            wi = [gk[:U, :H].T, gk[:U, H:].T, weights[pre + 'candidate/input_projection/kernel'].T]
            wr = [gk[U:, :H].T, gk[U:, H:].T, weights[pre + 'candidate/hidden_projection/kernel'].T]
            ws += [np.concatenate([m.reshape(-1) for m in wi]), np.concatenate([m.reshape(-1) for m in wr])]
            # gates bias: cuDNN keeps two biases that TF adds up; put everything into the input-side one
            bs += [np.concatenate([gb[:H], gb[H:], weights[pre + 'candidate/input_projection/bias']]),
                   np.concatenate([np.zeros(2 * H, np.float32), weights[pre + 'candidate/hidden_projection/bias']])]
        opaque['{}/gru/cudnn_gru/opaque_kernel'.format(scope)] = np.concatenate(ws + bs).astype(np.float32)
    write_tensor_bundle(str(tmp_path / 'opaque'), opaque, block_entries=16, crc_fn=C.crc32c)
    got = C.load_checkpoint(str(tmp_path / 'opaque'), hp)
    assert set(got) == set(weights) and all(np.array_equal(got[k], weights[k]) for k in weights)
    # the GRUCell configuration cannot take such a checkpoint: the error says why
    with pytest.raises(KeyError, match='force_cudnn'):
        C.load_checkpoint(str(tmp_path / 'opaque'), pkg('tacotron.params').ModelParams())


# ---- byte-level known answer of the bundle layout: literal bytes, laid out by hand from the published formats (LevelDB
# table_format.md; tensorflow/core/protobuf/tensor_bundle.proto; tensorflow/core/lib/io/table_builder.cc for what TensorFlow's
# builder emits: restart interval 16, no compression, the index key a short successor of the block's last key, an empty
# metaindex block, 48-byte footer).  tests/tf_bundle_writer.py has no part in it.
_KAT_INDEX = bytes.fromhex(
    # data block, entry 1: key "" -> BundleHeaderProto {num_shards: 1, version {producer: 1}}
    '00' '00' '06'                               # shared 0, non-shared 0, value 6 bytes
    '0801' '1a02' '0801'                         # field 1 varint 1; field 3 (VersionDef) len 2 {field 1 varint 1}
    # entry 2: key "dense/bias" -> BundleEntryProto {dtype DT_FLOAT, shape [2], size 8, crc32c}
    '00' '0a' '0f' '64656e73652f62696173'        # shared 0, non-shared 10, value 15 bytes, "dense/bias"
    '0801' '1204' '1202' '0802' '2808' '35' '593ce535'   # dtype 1; shape{dim{size 2}}; size 8; masked crc32c (fixed32, LE)
    # entry 3: key "dense/kernel": shares "dense/" (6 bytes) with its predecessor -> only "kernel" is stored
    '06' '06' '15' '6b65726e656c'                # shared 6, non-shared 6, value 21 bytes
    '0801' '1208' '1202' '0802' '1202' '0802' '2008' '2810' '35' '17be60c6'   # dtype 1; shape [2][2]; offset 8; size 16; crc
    '00000000' '01000000'                        # restart array: one restart at offset 0; number of restarts
    '00' '24ff6942'                              # block trailer: no compression; masked crc32c of block + type byte
    # metaindex block: empty (one restart at 0) + trailer
    '00000000' '01000000' '00' 'c0f2a1b0'
    # index block: key "dense/l" (shortest successor of "dense/kernel") -> BlockHandle {offset 0, size 0x4b} + trailer
    '00' '07' '02' '64656e73652f6c' '004b' '00000000' '01000000' '00' '1281424c'
    # footer: metaindex handle {0x50, 8}, index handle {0x5d, 0x14}, zero padding to 40 bytes, magic 0xdb4775248b80fb57 (LE)
    '5008' '5d14' + '00' * 36 + '57fb808b247547db')
_KAT_DATA = bytes.fromhex('0000803f00000040' '0000003f000080bf0000404000008040')   # bias [1, 2]; kernel [[0.5, -1], [3, 4]]


def test_bundle_layout_known_answer(tmp_path):
    C = pkg('tacotron.checkpoint')
    """The importer reads a bundle whose every byte is written out above: prefix-compressed keys, the block trailers' masked
    CRC32Cs, the index block's successor key, the footer -- and rejects it when one payload byte or one index byte changes."""
    assert len(_KAT_INDEX) == 166 and _KAT_INDEX[-8:] == bytes.fromhex('57fb808b247547db')
    prefix = str(tmp_path / 'model.ckpt-7')
    with open(prefix + '.index', 'wb') as f:
        f.write(_KAT_INDEX)
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(_KAT_DATA)
    got = C.read_tensor_bundle(prefix)
    assert sorted(got) == ['dense/bias', 'dense/kernel']
    assert got['dense/bias'].dtype == np.float32 and got['dense/bias'].tolist() == [1.0, 2.0]
    assert got['dense/kernel'].shape == (2, 2) and got['dense/kernel'].tolist() == [[0.5, -1.0], [3.0, 4.0]]
    # the masked CRC32C of the literal (what TensorFlow stores: rotate right 15, add 0xa282ead8): known answers of the entries
    assert C.crc32c(_KAT_DATA[:8]) == C.unmask_crc(0x35e53c59) and C.crc32c(_KAT_DATA[8:]) == C.unmask_crc(0xc660be17)
    # a flipped payload bit: the tensor's CRC32C no longer matches
    bad = bytearray(_KAT_DATA)
    bad[9] ^= 0x10
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(bytes(bad))
    with pytest.raises(C.ChecksumError):
        C.read_tensor_bundle(prefix)
    assert C.read_tensor_bundle(prefix, verify=False)['dense/kernel'].shape == (2, 2)
    # a flipped byte inside the data block of the index: the block trailer's CRC32C no longer matches
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(_KAT_DATA)
    badi = bytearray(_KAT_INDEX)
    badi[12] ^= 0x01
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(badi))
    with pytest.raises(ValueError):
        C.read_tensor_bundle(prefix)
    # a wrong magic number is not a table
    with open(prefix + '.index', 'wb') as f:
        f.write(_KAT_INDEX[:-1] + b'\x00')
    with pytest.raises(ValueError):
        C.read_tensor_bundle(prefix)
