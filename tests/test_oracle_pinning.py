"""CPU: what pins the oracle beyond its golden vectors (parity stays "unpinned": TensorFlow 1.8 and librosa cannot
run here) -- independent formulations on hypothesis-generated shapes, so that a drift of the numpy
restatement cannot go unnoticed:

  * conv1d 'SAME' / dense / highway / batch-norm / max-pool against torch on random shapes and kernel sizes;
  * TF-1.8 GRUCell against a second restatement written from the cell's definition with SPLIT kernels
    (x Wx + h Wh instead of concat([x, h]) W) in torch autograd-free float64, and through a bidirectional run;
  * the CudnnCompatibleGRUCell form against torch.nn.GRU(bidirectional=True) on random shapes;
  * one Luong attention step (score, softmax over the whole memory, context, attention layer) against torch einsum;
  * STFT / iSTFT against scipy.signal (a third formulation next to numpy and torch.stft) at the reference's
    2048 / 1102 / 275 and on random window / hop pairs."""
import numpy as np
import pytest
import scipy.signal
import torch
import torch.nn.functional as F
from hypothesis import given, settings, HealthCheck
from hypothesis import strategies as st

from oracle import audio_oracle as A
from oracle import tacotron_oracle as O

SET = settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.too_slow])


def _rng(seed):
    return np.random.default_rng(seed)


@SET
@given(B=st.integers(1, 3), T=st.integers(1, 12), cin=st.integers(1, 6), cout=st.integers(1, 5), k=st.integers(1, 16),
       seed=st.integers(0, 2 ** 31))
def test_conv1d_same_any_shape(B, T, cin, cout, k, seed):
    r = _rng(seed)
    x, w, b = r.standard_normal((B, T, cin)), r.standard_normal((k, cin, cout)), r.standard_normal(cout)
    pad_l = (k - 1) // 2                                           # TF 'SAME': the extra pad goes to the right
    xt = F.pad(torch.tensor(x).permute(0, 2, 1), (pad_l, k - 1 - pad_l))
    ref = F.conv1d(xt, torch.tensor(w).permute(2, 1, 0), torch.tensor(b)).permute(0, 2, 1).numpy()
    got = O.conv1d_same(x, w, b)
    assert got.shape == (B, T, cout) and np.allclose(got, ref, atol=1e-11)


@SET
@given(B=st.integers(1, 3), T=st.integers(1, 9), n_in=st.integers(1, 7), n_out=st.integers(1, 7),
       act=st.sampled_from([None, 'relu', 'sigmoid', 'tanh']), seed=st.integers(0, 2 ** 31))
def test_dense_and_highway_any_shape(B, T, n_in, n_out, act, seed):
    r = _rng(seed)
    x = r.standard_normal((B, T, n_in))
    w = {'d/kernel': r.standard_normal((n_in, n_out)), 'd/bias': r.standard_normal(n_out)}
    ref = torch.tensor(x) @ torch.tensor(w['d/kernel']) + torch.tensor(w['d/bias'])
    ref = {None: ref, 'relu': torch.relu(ref), 'sigmoid': torch.sigmoid(ref), 'tanh': torch.tanh(ref)}[act].numpy()
    assert np.allclose(O.dense(x, w, 'd', activation=act), ref, atol=1e-12)
    # highway layer (reference layers.py:241-258): H = relu, T = sigmoid, y = H T + x (1 - T)
    hw = {'h/H/kernel': r.standard_normal((n_in, n_in)), 'h/H/bias': r.standard_normal(n_in),
          'h/T/kernel': r.standard_normal((n_in, n_in)), 'h/T/bias': r.standard_normal(n_in)}
    xt = torch.tensor(x)
    hh = torch.relu(xt @ torch.tensor(hw['h/H/kernel']) + torch.tensor(hw['h/H/bias']))
    tt = torch.sigmoid(xt @ torch.tensor(hw['h/T/kernel']) + torch.tensor(hw['h/T/bias']))
    assert np.allclose(O.highway_layer(x, hw, 'h'), (hh * tt + xt * (1 - tt)).numpy(), atol=1e-12)


@SET
@given(B=st.integers(1, 3), T=st.integers(1, 9), C=st.integers(1, 6), scale=st.booleans(), seed=st.integers(0, 2 ** 31))
def test_batch_norm_and_max_pool_any_shape(B, T, C, scale, seed):
    r = _rng(seed)
    x = r.standard_normal((B, T, C))
    w = {'bn/moving_mean': r.standard_normal(C), 'bn/moving_variance': r.random(C) + 0.1, 'bn/beta': r.standard_normal(C),
         'bn/gamma': r.random(C) + 0.5}
    ref = F.batch_norm(torch.tensor(x).permute(0, 2, 1), torch.tensor(w['bn/moving_mean']), torch.tensor(w['bn/moving_variance']),
                       torch.tensor(w['bn/gamma']) if scale else None, torch.tensor(w['bn/beta']), False, 0.0,
                       1e-3).permute(0, 2, 1).numpy()
    assert np.allclose(O.batch_norm_inference(x, w, 'bn', scale=scale), ref, atol=1e-12)
    xt = F.pad(torch.tensor(x).permute(0, 2, 1), (0, 1), value=-np.inf)
    assert np.array_equal(O.max_pool_2_1_same(x), F.max_pool1d(xt, 2, 1).permute(0, 2, 1).numpy())


def _tf_gru_cell_split(x, h, gk, gb, ck, cb):
    """tf.nn.rnn_cell.GRUCell.call [TF-1.8 rnn_cell_impl.py], restated with the kernels split into their input and
    state row blocks (rows are ordered [input ; state]): no concatenation, different summation grouping."""
    n_in, U = x.shape[-1], h.shape[-1]
    gx, gh = gk[:n_in], gk[n_in:]
    value = torch.sigmoid(x @ gx + h @ gh + gb)
    r, u = value[..., :U], value[..., U:]                      # array_ops.split(value, 2): r first, then u
    cx, ch = ck[:n_in], ck[n_in:]
    c = torch.tanh(x @ cx + (r * h) @ ch + cb)
    return u * h + (1 - u) * c


@SET
@given(B=st.integers(1, 4), T=st.integers(1, 8), n_in=st.integers(1, 6), U=st.integers(1, 6), seed=st.integers(0, 2 ** 31))
def test_tf_gru_cell_against_split_kernel_restatement(B, T, n_in, U, seed):
    r = _rng(seed)
    w = {}
    for d in ('fw', 'bw'):
        s = 'gru/{}/gru_cell_{}'.format(d, d)
        w[s + '/gates/kernel'] = r.standard_normal((n_in + U, 2 * U))
        w[s + '/gates/bias'] = 1.0 + 0.1 * r.standard_normal(2 * U)
        w[s + '/candidate/kernel'] = r.standard_normal((n_in + U, U))
        w[s + '/candidate/bias'] = 0.1 * r.standard_normal(U)
    x = r.standard_normal((B, T, n_in))
    got = O.bi_gru(x, w, 'gru', U)
    xt = torch.tensor(x)
    ref = torch.zeros(B, T, 2 * U, dtype=torch.float64)
    for di, d in enumerate(('fw', 'bw')):
        s = 'gru/{}/gru_cell_{}'.format(d, d)
        p = [torch.tensor(w[s + k]) for k in ('/gates/kernel', '/gates/bias', '/candidate/kernel', '/candidate/bias')]
        h = torch.zeros(B, U, dtype=torch.float64)
        for t in (range(T) if di == 0 else range(T - 1, -1, -1)):    # no sequence_length: the whole padded length
            h = _tf_gru_cell_split(xt[:, t], h, *p)
            ref[:, t, di * U:(di + 1) * U] = h
    assert np.allclose(got, ref.numpy(), atol=1e-11)


@SET
@given(B=st.integers(1, 3), T=st.integers(1, 7), n_in=st.integers(1, 5), U=st.integers(1, 5), seed=st.integers(0, 2 ** 31))
def test_cudnn_form_bidirectional_against_torch_gru(B, T, n_in, U, seed):
    r = _rng(seed)
    w = {}
    gru = torch.nn.GRU(n_in, U, batch_first=True, bidirectional=True).double()
    with torch.no_grad():
        for d, suf in (('fw', ''), ('bw', '_reverse')):
            s = 'gru/{}/gru_cell_{}'.format(d, d)
            gk, gb = r.standard_normal((n_in + U, 2 * U)), r.standard_normal(2 * U)
            ci, cib = r.standard_normal((n_in, U)), r.standard_normal(U)
            ch, chb = r.standard_normal((U, U)), r.standard_normal(U)
            w.update({s + '/gates/kernel': gk, s + '/gates/bias': gb, s + '/candidate/input_projection/kernel': ci,
                      s + '/candidate/input_projection/bias': cib, s + '/candidate/hidden_projection/kernel': ch,
                      s + '/candidate/hidden_projection/bias': chb})
            # torch packs [r | z | n] with z = TF's u
            getattr(gru, 'weight_ih_l0' + suf).copy_(torch.tensor(np.concatenate([gk[:n_in].T, ci.T])))
            getattr(gru, 'weight_hh_l0' + suf).copy_(torch.tensor(np.concatenate([gk[n_in:].T, ch.T])))
            getattr(gru, 'bias_ih_l0' + suf).copy_(torch.tensor(np.concatenate([gb, cib])))
            getattr(gru, 'bias_hh_l0' + suf).copy_(torch.tensor(np.concatenate([np.zeros(2 * U), chb])))
    x = r.standard_normal((B, T, n_in))
    ref, _ = gru(torch.tensor(x))
    assert np.allclose(O.bi_gru(x, w, 'gru', U, cudnn=True), ref.detach().numpy(), atol=1e-11)


@SET
@given(B=st.integers(1, 4), Ts=st.integers(1, 20), A_=st.integers(1, 8), seed=st.integers(0, 2 ** 31))
def test_luong_attention_step_against_torch(B, Ts, A_, seed):
    """LuongAttention(scale=False) + AttentionWrapper's attention_layer [TF-1.8]: keys = memory W_mem (no bias),
    score = q . keys^T, softmax over ALL positions, context = a . memory (unprojected), att = [q; ctx] W_att."""
    r = _rng(seed)
    q = torch.tensor(r.standard_normal((B, A_)))
    mem = torch.tensor(r.standard_normal((B, Ts, A_)))
    w_mem, w_att = torch.tensor(r.standard_normal((A_, A_))), torch.tensor(r.standard_normal((2 * A_, A_)))
    keys = mem @ w_mem
    a = torch.softmax(torch.einsum('ba,bta->bt', q, keys), -1)
    ctx = torch.einsum('bt,bta->ba', a, mem)
    att = torch.cat([q, ctx], -1) @ w_att
    # the oracle's pieces
    s = np.einsum('ba,bta->bt', q.numpy(), (mem.numpy() @ w_mem.numpy()))
    al = O.softmax_lastaxis(s)
    ctx_o = np.einsum('bt,bta->ba', al, mem.numpy())
    att_o = np.concatenate([q.numpy(), ctx_o], -1) @ w_att.numpy()
    assert np.allclose(al, a.numpy(), atol=1e-13) and np.allclose(al.sum(-1), 1.0)
    assert np.allclose(att_o, att.numpy(), atol=1e-11)


def _scipy_stft(y, n_fft, hop, win):
    window = A.pad_center(scipy.signal.get_window('hann', win, fftbins=True), n_fft)
    _, _, Z = scipy.signal.stft(y, window=window, nperseg=n_fft, noverlap=n_fft - hop, nfft=n_fft, boundary='even',
                                padded=False, return_onesided=True)
    return Z * window.sum()                                      # scipy scales by 1 / sum(window)


def test_stft_against_scipy_at_the_reference_sizes():
    y = _rng(3).standard_normal(275 * 24).astype(np.float32)
    S = A.stft(y, 2048, 275, 1102, dtype=np.complex128)
    Z = _scipy_stft(y.astype(np.float64), 2048, 275, 1102)
    assert S.shape == Z.shape == (1025, 25)
    assert np.linalg.norm(S - Z) / np.linalg.norm(Z) < 1e-12


@SET
@given(hop=st.integers(40, 600), ratio=st.floats(1.5, 7.5), frames=st.integers(9, 20), seed=st.integers(0, 2 ** 31))
def test_stft_istft_against_scipy_any_window(hop, ratio, frames, seed):
    n_fft = 2048
    win = int(min(n_fft, max(hop + 1, round(hop * ratio))))
    y = _rng(seed).standard_normal(max(hop * frames + 5, n_fft + hop)).astype(np.float64)   # scipy: signal >= window
    S = A.stft(y, n_fft, hop, win, dtype=np.complex128)
    Z = _scipy_stft(y, n_fft, hop, win)
    assert S.shape == Z.shape and np.linalg.norm(S - Z) / np.linalg.norm(Z) < 1e-11
    # inverse: scipy's overlap-add / window-sum-square normalisation agrees with librosa's away from the edges
    window = A.pad_center(scipy.signal.get_window('hann', win, fftbins=True), n_fft)
    _, yi = scipy.signal.istft(Z / window.sum(), window=window, nperseg=n_fft, noverlap=n_fft - hop, nfft=n_fft,
                               boundary=True, input_onesided=True)
    got = A.istft(S, hop, win, dtype=np.float64)
    m = min(len(yi), len(got))
    lo = n_fft
    if m > 2 * lo + 8:
        assert np.abs(got[lo:m - lo] - yi[lo:m - lo]).max() < 1e-9
        assert np.abs(got[lo:m - lo] - y[lo:m - lo]).max() < 1e-9   # ... and both invert the transform
