"""CPU: the TensorFlow-free checkpoint importer (tacotron/checkpoint.py) against a bundle written in the
published format by tests/tf_bundle_writer.py."""
import os

import numpy as np
import pytest

from conftest import pkg
from tf_bundle_writer import crc32c, write_tensor_bundle


def test_crc32c_known_answer():
    assert crc32c(b'123456789') == 0xE3069283        # RFC 3720 check value


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
    write_tensor_bundle(str(run / 'model.ckpt-215000'), ck, block_entries=16)
    (run / 'checkpoint').write_text('model_checkpoint_path: "model.ckpt-215000"\n'
                                    'all_model_checkpoint_paths: "model.ckpt-210000"\n'
                                    'all_model_checkpoint_paths: "model.ckpt-215000"\n')
    assert C.latest_checkpoint(str(run)) == os.path.join(str(run), 'model.ckpt-215000')
    got = C.load_checkpoint(str(run), hparams)                  # directory -> latest, like inference.py:49-53
    assert set(got) == set(weights)
    assert all(np.array_equal(got[k], weights[k]) for k in weights)
    # a scope spelled differently in the checkpoint can be aliased; a missing variable is an error
    renamed = {('decoder/memory_layer/kernel' if k == 'decoder2/memory_layer/kernel' else k): v for k, v in weights.items()}
    write_tensor_bundle(str(tmp_path / 'other'), renamed, block_entries=16)
    with pytest.raises(KeyError):
        C.load_checkpoint(str(tmp_path / 'other'), hparams)
    ok = C.load_checkpoint(str(tmp_path / 'other'), hparams, aliases={'decoder/memory_layer/kernel': 'decoder2/memory_layer/kernel'})
    assert np.array_equal(ok['decoder2/memory_layer/kernel'], weights['decoder2/memory_layer/kernel'])
    bad = dict(weights)
    bad['dense/bias'] = np.zeros(7, np.float32)
    write_tensor_bundle(str(tmp_path / 'bad'), bad, block_entries=16)
    with pytest.raises(ValueError):
        C.load_checkpoint(str(tmp_path / 'bad'), hparams)
