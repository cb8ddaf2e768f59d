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


@pytest.mark.parametrize('B,Ts,S', [(2, 9, 4), (3, 11, 3), (5, 60, 12)])   # Ts % 3 = 0, 2, 0; T = 5 S: % 3 = 2, 0, 0
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


@pytest.mark.parametrize('B,Ts,S', [(2, 9, 4), (19, 60, 12)])
def test_cudnn_persistent_decoder(cudnn_setup, B, Ts, S):
    """CudnnCompatibleGRUCell through the persistent kernel: r, u, h W_ch and x W_ci from ONE staged [x ; h] tile, one
    hand-off per cell.  Against the oracle and against the launch-per-layer path."""
    hp, w, eng = cudnn_setup
    rng = np.random.default_rng(40 + B)
    memory = (rng.standard_normal((B, Ts, 256)) * 1.5).astype(np.float32)
    ref_mel, ref_al = O.decoder(memory.astype(np.float64), O.cast_weights(w, np.float64), hp, n_steps=S)
    dev = eng.to_device(memory)
    try:
        eng.set_option('persistent_decoder', 2)
        assert eng.decoder_kernel_choice(B, Ts, pipelined=False) == 2   # the weight-stationary kernel (round 5: both GRU forms)
        mel, al = eng.decoder_forward(dev, S)
        eng.synchronize()
        mel, al = mel.to_host(), al.to_host()
        eng.set_option('pd_ws', 0)                                       # ... and decoder_persistent.hip
        assert eng.decoder_kernel_choice(B, Ts, pipelined=False) == 1
        mel_s, al_s = eng.decoder_forward(dev, S)
        eng.synchronize()
        assert rel_l2(mel_s.to_host(), ref_mel) < 1e-3 and np.abs(al_s.to_host() - ref_al).max() < 1e-4
        eng.set_option('pd_ws', 1)
        eng.set_option('persistent_decoder', 0)
        mel0, al0 = eng.decoder_forward(dev, S)
        mel0, al0 = mel0.to_host(), al0.to_host()
    finally:
        eng.set_option('pd_ws', 1)
        eng.set_option('persistent_decoder', 1)
    e_mel, e_al = rel_l2(mel, ref_mel), float(np.abs(al - ref_al).max())
    print('cudnn persistent decoder B={} Ts={} S={}: mel rel-L2 {:.3e}, align max-abs {:.3e}; vs launch path {:.3e}'.format(
        B, Ts, S, e_mel, e_al, rel_l2(mel, mel0)))
    assert e_mel < 1e-3 and e_al < 1e-4
    assert rel_l2(mel, mel0) < 1e-5 and np.abs(al - al0).max() < 1e-4


@pytest.mark.parametrize('B,Ts,S,delay', [(20, 50, 40, 1), (64, 150, 60, 2)])
def test_persistent_cudnn_decoder_with_a_late_stager(cudnn_setup, B, Ts, S, delay):
    """CudnnCompatibleGRUCell form in the persistent kernel: the candidate phase continues on the staged tile without
    a wait, so a workgroup stores its slice of h' while a peer may still be STAGING the previous h.  With one state
    buffer that peer read a mix of h_{t-1} and h_t (round-2 advisory); the states are double-buffered by step parity
    now.  `pd_debug_delay` makes workgroup 3 of every cluster sleep ~3.4 us x delay between each wait and its staging
    loads -- the late stager -- and the result must still equal the launch-per-layer path (1e-5: other summation
    order) and be bit-identical to the undelayed persistent run."""
    hp, w, eng = cudnn_setup
    rng = np.random.default_rng(40 + B)
    memory = eng.to_device((rng.standard_normal((B, Ts, 256)) * 0.7).astype(np.float32))
    try:
        eng.set_option('persistent_decoder', 0)
        mel0, al0 = eng.decoder_forward(memory, S)
        mel0, al0 = mel0.to_host(), al0.to_host()
        eng.set_option('persistent_decoder', 2)
        mel1, al1 = eng.decoder_forward(memory, S)
        eng.synchronize()
        mel1, al1 = mel1.to_host(), al1.to_host()
        eng.set_option('debug_hooks', 1)
        eng.set_option('pd_debug_delay', delay)
        mel2, al2 = eng.decoder_forward(memory, S)
        eng.synchronize()
        mel2, al2 = mel2.to_host(), al2.to_host()
    finally:
        eng.set_option('pd_debug_delay', 0)
        eng.set_option('debug_hooks', 0)
        eng.set_option('persistent_decoder', 1)
    print('cudnn persistent vs launch path: mel rel-L2 {:.3e}; late stager vs undelayed: equal = {}'.format(
        rel_l2(mel1, mel0), np.array_equal(mel1, mel2)))
    assert rel_l2(mel1, mel0) < 1e-5 and np.abs(al1 - al0).max() < 1e-5
    assert np.array_equal(mel1, mel2) and np.array_equal(al1, al2)
