"""CPU: the numpy oracle against (a) its committed golden vectors, (b) independent torch-CPU
formulations of the same ops, (c) properties of model outputs the reference ships."""
import hashlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import pkg, rel_l2
from oracle import tacotron_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_synthetic_weights_are_reproducible(weights, hparams):
    W = pkg('tacotron.weights')
    g = np.load(os.path.join(GOLD, 'network_small.npz'))
    blob = W.pack_blob(weights, hparams)
    assert blob.size == 6855713 == W.n_parameters(hparams)      # 27.4 MB fp32
    assert hashlib.sha256(blob.tobytes()).hexdigest() == str(g['weights_sha256'])
    back = W.unpack_blob(blob, hparams)
    assert all(np.array_equal(back[k], weights[k]) for k in weights)


def test_oracle_reproduces_golden_network(weights64, hparams):
    g = np.load(os.path.join(GOLD, 'network_small.npz'))
    out = O.tacotron_predict(g['ids'], weights64, hparams, n_steps=int(g['n_steps']))
    for k in ('memory', 'reduced_mel', 'alignments', 'linear'):
        assert rel_l2(out[k], g[k]) < 1e-6, k
    assert out['mel'].shape == (2, 15, 80) and out['linear'].shape == (2, 15, 1025)


@pytest.mark.parametrize('k', [1, 2, 3, 4, 7, 16])
def test_conv1d_same_matches_torch(k):
    rng = np.random.default_rng(k)
    x = rng.standard_normal((2, 9, 5))
    w = rng.standard_normal((k, 5, 6))
    b = rng.standard_normal(6)
    pad_l = (k - 1) // 2
    xt = F.pad(torch.tensor(x).permute(0, 2, 1), (pad_l, k - 1 - pad_l))   # TF SAME: extra pad on the right
    ref = F.conv1d(xt, torch.tensor(w).permute(2, 1, 0), torch.tensor(b)).permute(0, 2, 1).numpy()
    assert np.allclose(O.conv1d_same(x, w, b), ref, atol=1e-12)


def test_max_pool_matches_torch():
    x = np.random.default_rng(0).standard_normal((3, 8, 4))
    xt = F.pad(torch.tensor(x).permute(0, 2, 1), (0, 1), value=-np.inf)
    ref = F.max_pool1d(xt, 2, 1).permute(0, 2, 1).numpy()
    assert np.array_equal(O.max_pool_2_1_same(x), ref)


def test_batch_norm_matches_torch():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 5, 4))
    w = {'bn/moving_mean': rng.standard_normal(4), 'bn/moving_variance': rng.random(4) + 0.5,
         'bn/beta': rng.standard_normal(4), 'bn/gamma': rng.random(4) + 0.5}
    ref = F.batch_norm(torch.tensor(x).permute(0, 2, 1), torch.tensor(w['bn/moving_mean']),
                       torch.tensor(w['bn/moving_variance']), torch.tensor(w['bn/gamma']),
                       torch.tensor(w['bn/beta']), False, 0.0, 1e-3).permute(0, 2, 1).numpy()
    assert np.allclose(O.batch_norm_inference(x, w, 'bn', scale=True), ref, atol=1e-12)
    w1 = dict(w)
    ref1 = F.batch_norm(torch.tensor(x).permute(0, 2, 1), torch.tensor(w['bn/moving_mean']),
                        torch.tensor(w['bn/moving_variance']), None, torch.tensor(w['bn/beta']), False, 0.0,
                        1e-3).permute(0, 2, 1).numpy()
    assert np.allclose(O.batch_norm_inference(x, w1, 'bn', scale=False), ref1, atol=1e-12)


def test_cudnn_gru_formulation_matches_torch_gru():
    """CudnnCompatibleGRUCell (S5') is torch.nn.GRU's formulation: independent cross-check of the
    gate order [r|u], h' = u*h + (1-u)*c and the r * (h W + b) placement."""
    rng = np.random.default_rng(2)
    n_in, U, B, T = 5, 4, 3, 6
    w = {'g/gates/kernel': rng.standard_normal((n_in + U, 2 * U)), 'g/gates/bias': rng.standard_normal(2 * U),
         'g/candidate/input_projection/kernel': rng.standard_normal((n_in, U)),
         'g/candidate/input_projection/bias': rng.standard_normal(U),
         'g/candidate/hidden_projection/kernel': rng.standard_normal((U, U)),
         'g/candidate/hidden_projection/bias': rng.standard_normal(U)}
    gru = torch.nn.GRU(n_in, U, batch_first=True).double()
    gk = w['g/gates/kernel']
    with torch.no_grad():   # torch packs [r | z | n]; z is TF's u
        gru.weight_ih_l0.copy_(torch.tensor(np.concatenate([gk[:n_in].T, w['g/candidate/input_projection/kernel'].T])))
        gru.weight_hh_l0.copy_(torch.tensor(np.concatenate([gk[n_in:].T, w['g/candidate/hidden_projection/kernel'].T])))
        gru.bias_ih_l0.copy_(torch.tensor(np.concatenate([w['g/gates/bias'], w['g/candidate/input_projection/bias']])))
        gru.bias_hh_l0.copy_(torch.tensor(np.concatenate([np.zeros(2 * U), w['g/candidate/hidden_projection/bias']])))
    x = rng.standard_normal((B, T, n_in))
    ref, _ = gru(torch.tensor(x))
    h = np.zeros((B, U))
    for t in range(T):
        h = O.gru_cell(x[:, t], h, w, 'g', cudnn=True)
        assert np.allclose(h, ref[:, t].detach().numpy(), atol=1e-12)


def test_tf_gru_cell_equals_cudnn_form_when_reset_commutes():
    """GRUCell (S5) and the cudnn form coincide when r multiplies before or after a diagonal
    hidden projection -- checks that the two code paths differ ONLY in that placement."""
    rng = np.random.default_rng(3)
    n_in, U, B = 3, 4, 2
    d = rng.standard_normal(U)
    w = {'g/gates/kernel': rng.standard_normal((n_in + U, 2 * U)), 'g/gates/bias': rng.standard_normal(2 * U)}
    ck_in = rng.standard_normal((n_in, U))
    w['g/candidate/kernel'] = np.concatenate([ck_in, np.diag(d)])
    w['g/candidate/bias'] = rng.standard_normal(U)
    w['g/candidate/input_projection/kernel'] = ck_in
    w['g/candidate/input_projection/bias'] = w['g/candidate/bias']
    w['g/candidate/hidden_projection/kernel'] = np.diag(d)
    w['g/candidate/hidden_projection/bias'] = np.zeros(U)
    x, h = rng.standard_normal((B, n_in)), rng.standard_normal((B, U))
    assert np.allclose(O.gru_cell(x, h, w, 'g', False), O.gru_cell(x, h, w, 'g', True), atol=1e-12)


def test_bi_gru_runs_over_padding_and_reverses():
    rng = np.random.default_rng(4)
    U = 3
    w = {}
    for d in ('fw', 'bw'):
        s = 'gru/{}/gru_cell_{}'.format(d, d)
        w[s + '/gates/kernel'] = rng.standard_normal((2 + U, 2 * U))
        w[s + '/gates/bias'] = np.ones(2 * U)
        w[s + '/candidate/kernel'] = rng.standard_normal((2 + U, U))
        w[s + '/candidate/bias'] = np.zeros(U)
    x = rng.standard_normal((2, 5, 2))
    y = O.bi_gru(x, w, 'gru', U)
    # backward half = forward pass of the time-reversed input with the bw cell
    w2 = {k.replace('/bw/gru_cell_bw', '/fw/gru_cell_fw'): v for k, v in w.items() if '/bw/' in k}
    w2.update({k.replace('/fw/gru_cell_fw', '/bw/gru_cell_bw'): v for k, v in w.items() if '/fw/' in k})
    y2 = O.bi_gru(x[:, ::-1], w2, 'gru', U)
    assert np.allclose(y[..., U:], y2[:, ::-1, :U])
    assert np.allclose(y[..., :U], y2[:, ::-1, U:])


def test_decoder_attention_is_an_unmasked_softmax(weights64, hparams):
    rng = np.random.default_rng(5)
    mem = rng.standard_normal((2, 6, 256))
    mel, al = O.decoder(mem, weights64, hparams, n_steps=2)
    assert mel.shape == (2, 2, 400) and al.shape == (2, 2, 6)
    assert np.allclose(al.sum(-1), 1.0) and (al > 0).all()


def test_reference_shipped_alignments_have_the_modelled_properties():
    """The reference ships alignment dumps of a trained model (visualization/data/...): 200 decoder
    steps = 1000 // 5, and every column is a softmax over ALL memory positions (no masking)."""
    a = np.load(os.path.join(GOLD, 'reference_alignments_nancy_1.npz'))['alignments']
    assert a.shape == (1, 92, 200) and a.dtype == np.float32
    assert np.allclose(a.sum(1), 1.0, atol=5e-6)
    assert (a > 0).all()
    assert np.allclose(a[0, :, 0], a[0, :, 0].mean(), rtol=0.2)   # first step is near-uniform


@pytest.mark.parametrize('t,Ts,D', [(0, 40, 10), (17, 40, 10), (39, 40, 10), (5, 21, 10), (3, 9, 2)])
def test_local_luong_window_equals_masked_softmax(t, Ts, D):
    """Independent formulation of LocalLuongAttention (reference tacotron/attention.py:263-328,52-92):
    a softmax over the in-memory window == a softmax over the whole memory with -inf outside it."""
    rng = np.random.default_rng(t + Ts)
    B, U = 3, 16
    q = rng.standard_normal((B, U))
    keys = rng.standard_normal((B, Ts, U))
    values = rng.standard_normal((B, Ts, U))
    for gaussian in (False, True):
        ctx, al = O.local_luong_monotonic(q, keys, values, t, D, gaussian)
        p = min(max(t, D), Ts - (D + 1))
        score = torch.einsum('bd,btd->bt', torch.from_numpy(q), torch.from_numpy(keys))
        mask = torch.full((Ts,), float('-inf'), dtype=torch.float64)
        mask[p - D:p + D + 1] = 0.0
        a = torch.softmax(score + mask, -1)
        ctx_t = torch.einsum('bt,btd->bd', a, torch.from_numpy(values)).numpy()
        assert rel_l2(ctx, ctx_t) < 1e-12
        exp = a.numpy()
        if gaussian:
            exp = exp * np.exp(-((np.arange(Ts) - p) ** 2) / 2 * (D / 2) ** 2)
        np.testing.assert_allclose(al, exp, atol=1e-14)
        assert al.shape == (B, Ts)


def test_local_luong_decoder_reduces_to_global_when_window_is_the_memory(weights64, hparams):
    import copy
    hp = copy.deepcopy(hparams)
    hp.attention.mechanism = 'LocalLuongAttention'
    hp.attention.luong_local_window_D = 4
    hp.attention.luong_force_gaussian = False
    mem = np.random.default_rng(0).standard_normal((2, 9, 256))
    a, al_a = O.decoder(mem, weights64, hp, n_steps=5)
    b, al_b = O.decoder(mem, weights64, hparams, n_steps=5)
    assert rel_l2(a, b) < 1e-12 and np.abs(al_a - al_b).max() < 1e-14
    hp.attention.luong_local_score = 'general'
    with pytest.raises(NotImplementedError):
        O.decoder(mem, weights64, hp, n_steps=1)


@pytest.mark.parametrize('Ts,D,gaussian', [(40, 10, True), (25, 3, False), (64, 5, True)])
def test_predictive_local_luong_equals_masked_softmax_around_the_predicted_centre(Ts, D, gaussian):
    """Independent formulation of LocalLuongAttention's PREDICTIVE mode (reference tacotron/attention.py:246-258,
    273-342, 52-92): p = T_s sigmoid(v_p^T tanh(W_p h)) per row, then a softmax over the whole memory with -inf
    outside [floor(p) - D, floor(p) + D]; the gaussian uses the real-valued p."""
    rng = np.random.default_rng(Ts + D)
    B, U = 4, 16
    q = rng.standard_normal((B, U))
    keys, values = rng.standard_normal((B, Ts, U)), rng.standard_normal((B, Ts, U))
    wp, vp = rng.standard_normal((U, U)) * 0.3, rng.standard_normal((U, 1)) * 0.05   # centres stay near T_s / 2
    ctx, al, p = O.local_luong_predictive(q, keys, values, wp, vp, D, gaussian)
    tq = torch.from_numpy(q)
    p_t = Ts * torch.sigmoid(torch.tanh(tq @ torch.from_numpy(wp)) @ torch.from_numpy(vp))[:, 0]
    assert np.allclose(p, p_t.numpy(), atol=1e-12)
    score = torch.einsum('bd,btd->bt', tq, torch.from_numpy(keys))
    pos = torch.arange(Ts, dtype=torch.float64)[None, :]
    c = torch.floor(p_t)[:, None]
    mask = torch.where((pos >= c - D) & (pos <= c + D), 0.0, float('-inf'))
    a = torch.softmax(score + mask, -1)
    assert rel_l2(ctx, torch.einsum('bt,btd->bd', a, torch.from_numpy(values)).numpy()) < 1e-12
    exp = a.numpy()
    if gaussian:
        exp = exp * np.exp(-((np.arange(Ts)[None, :] - p[:, None]) ** 2) / 2 * (D / 2) ** 2)
    np.testing.assert_allclose(al, exp, atol=1e-14)
    # a window that leaves the memory is where the reference stops being defined
    with pytest.raises(ValueError):
        O.local_luong_predictive(q, keys, values, wp, vp, Ts, gaussian)   # 2D+1 > T_s
