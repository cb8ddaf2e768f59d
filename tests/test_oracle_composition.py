"""A second, independently written formulation of the WHOLE forward pass (torch, float64) against the oracle.

tests/test_oracle_network.py and tests/test_oracle_pinning.py pin the oracle op by op; this file pins the
COMPOSITION -- the order of relu and batch-norm, the residual and lifter wiring of the CBHG, what the attention
wrapper feeds to whom, which frame is fed back, how the reduced frames are reshaped -- by restating the network a
second time from the reference's graph-building code (tacotron/model.py:124-401, layers.py:150-594,
wrappers.py:94-124, helpers.py:83-205) on torch primitives (F.conv1d, F.max_pool1d, F.batch_norm, F.linear, bmm,
softmax) instead of the oracle's numpy loops, reading the same TensorFlow-named weight dictionary.  Nothing here
imports or calls the oracle except the final comparison.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import pkg
from oracle import tacotron_oracle as O

P = pkg('tacotron.params')
W = pkg('tacotron.weights')

T64 = torch.float64
BN_EPS = 1e-3   # tf.layers.batch_normalization default epsilon


def t(a):
    return torch.as_tensor(np.asarray(a), dtype=T64)


def dense(x, w, name, act=None, bias=True):
    """wrapped_dense (layers.py:64-111): tf.layers.Dense on the last axis, kernel (in, out)."""
    y = F.linear(x, t(w[name + '/kernel']).T, t(w[name + '/bias']) if bias else None)
    return act(y) if act else y


def conv1d_same(x, kernel, bias):
    """tf.layers.conv1d(padding='SAME', strides=1) on (B, T, C): cross-correlation, k - 1 zeros split with the
    extra one on the RIGHT for even k; kernel (k, in, out)."""
    k = kernel.shape[0]
    left = (k - 1) // 2
    xp = F.pad(x.transpose(1, 2), (left, k - 1 - left))               # (B, C, T + k - 1)
    return F.conv1d(xp, t(kernel).permute(2, 1, 0), t(bias)).transpose(1, 2)


def batch_norm(x, w, scope, scale):
    g = t(w[scope + '/gamma']) if scale else None
    return F.batch_norm(x.transpose(1, 2), t(w[scope + '/moving_mean']), t(w[scope + '/moving_variance']), g,
                        t(w[scope + '/beta']), training=False, eps=BN_EPS).transpose(1, 2)


def pre_net(x, w, scope, layers):
    """layers.py:262-304 (dropout inactive outside training)."""
    for i, (units, _drop, act) in enumerate(layers):
        x = dense(x, w, '{}/{}-FC-{}'.format(scope, i + 1, units), torch.relu if act == 'relu' else None)
    return x


def gru_cell(x, h, w, scope):
    """tf.nn.rnn_cell.GRUCell [TF 1.8]: gates on [x ; h] ordered r | u, candidate on [x ; r * h].  When the weight
    dictionary holds the CudnnCompatibleGRUCell variables (force_cudnn, reference layers.py:560-577,
    model.py:226-229,257-262) the candidate is tanh(x W_ci + b_ci + r * (h W_ch + b_ch)) instead."""
    U = h.shape[-1]
    g = torch.sigmoid(F.linear(torch.cat([x, h], -1), t(w[scope + '/gates/kernel']).T, t(w[scope + '/gates/bias'])))
    r, u = g[..., :U], g[..., U:]
    if scope + '/candidate/kernel' in w:
        c = torch.tanh(F.linear(torch.cat([x, r * h], -1), t(w[scope + '/candidate/kernel']).T, t(w[scope + '/candidate/bias'])))
    else:
        ci = F.linear(x, t(w[scope + '/candidate/input_projection/kernel']).T, t(w[scope + '/candidate/input_projection/bias']))
        ch = F.linear(h, t(w[scope + '/candidate/hidden_projection/kernel']).T, t(w[scope + '/candidate/hidden_projection/bias']))
        c = torch.tanh(ci + r * ch)
    return u * h + (1.0 - u) * c


def bi_gru(x, w, scope, U):
    """bidirectional_dynamic_rnn without sequence_length (layers.py:579-592): both directions over all T."""
    B, T, _ = x.shape
    outs = []
    for d, order in (('fw', range(T)), ('bw', range(T - 1, -1, -1))):
        h = torch.zeros(B, U, dtype=T64)
        seq = [None] * T
        for i in order:
            h = gru_cell(x[:, i], h, w, '{}/gru/{}/gru_cell_{}'.format(scope, d, d))
            seq[i] = h
        outs.append(torch.stack(seq, 1))
    return torch.cat(outs, -1)


def cbhg(x, w, scope, hp_c):
    """layers.py:448-594."""
    banks = []
    for k in range(1, hp_c.n_banks + 1):
        name = '{}/convolution_banks/conv-{}-{}'.format(scope, k, hp_c.n_filters)
        y = torch.relu(conv1d_same(x, w[name + '/kernel'], w[name + '/bias']))
        bn = '{}/convolution_banks/batch_normalization{}'.format(scope, '' if k == 1 else '_{}'.format(k - 1))
        banks.append(batch_norm(y, w, bn, scale=False))                 # relu BEFORE the normalisation (layers.py:361-383)
    y = torch.cat(banks, -1)
    # max_pooling1d(pool 2, stride 1, 'SAME'): one pad position on the right that never wins
    y = F.max_pool1d(F.pad(y.transpose(1, 2), (0, 1), value=float('-inf')), 2, 1).transpose(1, 2)
    for i, (filters, ksize, act) in enumerate(hp_c.projections):
        name = '{}/projections/{}-conv-{}-{}'.format(scope, i + 1, ksize, filters)
        y = conv1d_same(y, w[name + '/conv1d/kernel'], w[name + '/conv1d/bias'])
        if act == 'relu':
            y = torch.relu(y)
        y = batch_norm(y, w, name + '/batch_normalization', scale=True)
    y = y + x                                                            # residual with the CBHG input
    y = dense(y, w, scope + '/lifter', torch.relu)
    for l in range(hp_c.n_highway_layers):
        hs = '{}/highway_network/highway_layer_{}'.format(scope, l)
        h = dense(y, w, hs + '/H', torch.relu)
        g = dense(y, w, hs + '/T', torch.sigmoid)
        y = h * g + y * (1.0 - g)
    return bi_gru(y, w, scope, hp_c.n_gru_units)


def decoder(memory, w, hp, n_steps):
    """model.py:175-334 with tf.contrib.seq2seq.AttentionWrapper(LuongAttention) semantics [TF 1.8]: keys =
    memory_layer(memory), values = memory, cell input = concat([inputs, previous attention]) (through the
    PrenetWrapper, wrappers.py:122-124), attention = attention_layer(concat([cell output, context])), the wrapper
    emits the attention; two residual GRU cells; output projection; feedback of the LAST n_mels values
    (helpers.py:161-205)."""
    B, Ts, _ = memory.shape
    dp = hp.decoder
    root = 'decoder2/decoder/output_projection_wrapper'
    aw = root + '/multi_rnn_cell/cell_0/attention_wrapper'
    keys = F.linear(memory, t(w['decoder2/memory_layer/kernel']).T)
    h_att = torch.zeros(B, dp.n_attention_units, dtype=T64)
    att = torch.zeros(B, dp.n_attention_units, dtype=T64)
    h_dec = [torch.zeros(B, dp.n_decoder_gru_units, dtype=T64) for _ in range(dp.n_gru_layers)]
    frame = torch.zeros(B, hp.n_mels, dtype=T64)                       # GO frame (helpers.py:108)
    outs, aligns = [], []
    for _ in range(n_steps):
        x = pre_net(torch.cat([frame, att], -1), w, aw + '/pre_net', dp.pre_net_layers)
        h_att = gru_cell(x, h_att, w, aw + '/gru_cell')
        score = torch.bmm(keys, h_att.unsqueeze(-1)).squeeze(-1)        # (B, Ts), no scaling, no mask
        a = torch.softmax(score, -1)
        ctx = torch.bmm(a.unsqueeze(1), memory).squeeze(1)
        att = F.linear(torch.cat([h_att, ctx], -1), t(w[aw + '/attention_layer/kernel']).T)
        y = att
        for l in range(dp.n_gru_layers):
            h_dec[l] = gru_cell(y, h_dec[l], w, '{}/multi_rnn_cell/cell_{}/gru_cell'.format(root, l + 1))
            y = y + h_dec[l]                                            # ResidualWrapper
        o = F.linear(y, t(w[root + '/kernel']).T, t(w[root + '/bias']))
        outs.append(o)
        aligns.append(a)
        frame = o[:, -hp.n_mels:]
    return torch.stack(outs, 1), torch.stack(aligns, 0)


def tacotron_forward(ids, w, hp, n_steps):
    emb = t(w['encoder/embedding'])[torch.as_tensor(ids, dtype=torch.long)]
    x = pre_net(emb, w, 'encoder/pre_net', hp.encoder.pre_net_layers)
    memory = cbhg(x, w, 'encoder', hp.encoder)
    reduced, align = decoder(memory, w, hp, n_steps)
    mel = reduced.reshape(reduced.shape[0], -1, hp.n_mels)              # (B, S, r * n_mels) -> (B, S * r, n_mels)
    post = cbhg(mel, w, 'post_process', hp.post)
    linear = dense(post, w, 'dense')
    return dict(memory=memory, reduced=reduced, mel=mel, linear=linear, alignments=align)


@pytest.mark.parametrize('B,Ts,S,seed,cudnn', [(2, 7, 3, 0, False), (3, 12, 4, 5, False), (2, 9, 3, 2, True)])
def test_whole_forward_pass_against_a_torch_restatement(B, Ts, S, seed, cudnn):
    hp = P.ModelParams()
    hp.force_cudnn = cudnn
    w = O.cast_weights(W.synthetic_weights(seed, hp), np.float64)
    rng = np.random.default_rng(100 + seed)
    ids = rng.integers(2, hp.vocabulary_size, (B, Ts)).astype(np.int32)
    ids[:, -1] = 1
    ids[0, Ts // 2:] = 0                                               # a padded row: no masking anywhere
    with torch.no_grad():
        got = tacotron_forward(ids, w, hp, S)
    ref = O.tacotron_predict(ids, w, hp, n_steps=S)

    def rel(a, b):
        a = a.numpy()
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))

    errs = {'mel': rel(got['mel'], ref['mel']), 'linear': rel(got['linear'], ref['linear']),
            'alignments': float(np.abs(got['alignments'].numpy() - ref['alignments']).max())}
    print('torch restatement vs oracle, B={} Ts={} S={}: {}'.format(B, Ts, S, errs))
    assert got['mel'].shape == (B, S * hp.reduction, hp.n_mels)
    assert got['linear'].shape == (B, S * hp.reduction, 1 + hp.n_fft // 2)
    for k, e in errs.items():
        assert e < 1e-10, (k, e)
    # every alignment row is a softmax over ALL memory positions, padding included
    np.testing.assert_allclose(got['alignments'].numpy().sum(-1), 1.0, atol=1e-12)


def test_griffin_lim_loop_against_a_torch_restatement():
    """The whole Griffin-Lim loop (reference audio/synthesis.py:85-123) on torch.istft / torch.stft in float64: the
    oracle follows librosa's mixed precision (float32 overlap-add buffer, complex64 spectra), so the two are compared
    after a few iterations, before the fixed-point iteration amplifies the rounding differences."""
    from oracle import audio_oracle as A
    n_fft, win, hop, T, n_iter = 2048, 1102, 275, 24, 3
    rng = np.random.default_rng(3)
    mag = (rng.random((1 + n_fft // 2, T)) ** 3 * 5).astype(np.float32)
    mag[5, 7] = 0.0
    init = rng.random(mag.shape)
    ref_wav, ref_mse = A.griffin_lim_v2(mag, win, hop, n_fft, n_iter, init_phase=init)

    window = torch.hann_window(win, periodic=True, dtype=T64)
    S = torch.as_tensor(mag, dtype=T64)
    angles = torch.exp(2j * np.pi * torch.as_tensor(init, dtype=T64))
    length = hop * (T - 1)
    mse = None
    for _ in range(n_iter):
        y = torch.istft(S * angles, n_fft, hop, win, window=window, center=True, length=length)
        D = torch.stft(y, n_fft, hop, win, window=window, center=True, pad_mode='reflect', return_complex=True)
        assert D.shape == S.shape                                      # 1 + length // hop frames: shape-stable
        angles = torch.exp(1j * torch.angle(D))                        # angle(0) = 0 -> 1 + 0j
        mse = float(((S - D.abs()) ** 2).mean())
    wav = torch.istft(S * angles, n_fft, hop, win, window=window, center=True, length=length).numpy()
    assert ref_wav.shape == wav.shape == (length,)
    err = float(np.linalg.norm(ref_wav - wav) / np.linalg.norm(wav))
    print('Griffin-Lim torch restatement vs oracle after {} iterations: rel-L2 {:.2e}, mse {:.6g} vs {:.6g}'.format(
        n_iter, err, ref_mse, mse))
    assert err < 1e-4
    assert abs(ref_mse - mse) < 1e-3 * mse
