"""GPU parity for force_cudnn=True: the reference's shipped default (params/model.py:51) builds
CudnnGRU / CudnnCompatibleGRUCell (layers.py:560-577, model.py:226-229,257-262), whose candidate is
tanh(x W_ci + b_ci + r * (h W_ch + b_ch)).  Both formulations are implemented; this file checks the
cudnn one end to end against the oracle."""
import copy

import numpy as np
import pytest

from conftest import pkg, rel_l2
from oracle import tacotron_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cudnn_setup(hparams):
    hp = copy.deepcopy(hparams)
    hp.force_cudnn = True
    W = pkg('tacotron.weights')
    w = W.synthetic_weights(1, hp)
    assert 'encoder/gru/fw/gru_cell_fw/candidate/hidden_projection/kernel' in w
    eng = pkg().Engine(hp)
    eng.load_weights(w)
    yield hp, w, eng
    eng.close()


def test_manifest_has_cudnn_variables(cudnn_setup):
    hp, w, eng = cudnn_setup
    names = [n for n, _ in eng.manifest()]
    assert set(names) == set(w)
    assert not any(n.endswith('candidate/kernel') for n in names)


@pytest.mark.parametrize('B,Ts,S', [(2, 9, 4), (5, 60, 12)])
def test_cudnn_full_network(cudnn_setup, B, Ts, S):
    hp, w, eng = cudnn_setup
    rng = np.random.default_rng(B)
    ids = rng.integers(2, 39, (B, Ts)).astype(np.int32)
    ids[:, -1] = 1
    ref = O.tacotron_predict(ids, O.cast_weights(w, np.float64), hp, n_steps=S)
    mem = eng.encoder_forward(ids)
    mel, al = eng.decoder_forward(mem, S)
    lin = eng.postnet_forward(mel.to_host().reshape(B, -1, 80))
    errs = dict(memory=rel_l2(mem.to_host(), ref['memory']), mel=rel_l2(mel.to_host(), ref['reduced_mel']),
                align=float(np.abs(al.to_host() - ref['alignments']).max()), linear=rel_l2(lin.to_host(), ref['linear']))
    print('cudnn variant B={} Ts={} S={}: {}'.format(B, Ts, S, errs))
    assert errs['memory'] < 1e-3 and errs['mel'] < 1e-3 and errs['linear'] < 1e-3 and errs['align'] < 1e-4


def test_cudnn_differs_from_grucell(cudnn_setup, engine, hparams):
    """Sanity: the two formulations are really different computations."""
    hp, w, eng = cudnn_setup
    mel = np.random.default_rng(0).random((1, 30, 80)).astype(np.float32)
    a = eng.postnet_forward(mel).to_host()
    ref = O.post_process(mel.astype(np.float64), O.cast_weights(w, np.float64), hp)
    assert rel_l2(a, ref) < 1e-3
