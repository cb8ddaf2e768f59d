"""CPU oracle for the Tacotron inference network (numpy restatement).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The shipped path is the HIP library; it never routes through this file.

PARITY UNPINNED: the arithmetic of this path lives in TensorFlow 1.8.0
(requirements.txt:1 of the reference), which is neither under /root/reference
nor installable here, and the reference holds no golden vectors for it.  What
follows restates the published semantics of the TF-1.8 ops at the reference's
own call sites; it is cross-checked op by op against independent torch-CPU
formulations in tests/test_oracle_*.py and pinned by nothing else.

Every function cites the reference lines it follows.  All tensors are
batch-major, channels-last; ``dtype`` selects float64 (golden) or float32
(the timed CPU baseline).
"""
import numpy as np

BN_EPS = 1e-3  # tf.layers.batch_normalization default epsilon [TF-1.8]


def _act(x, name):
    if name is None:
        return x
    if name == 'relu':
        return np.maximum(x, 0)
    if name == 'sigmoid':
        return sigmoid(x)
    if name == 'tanh':
        return np.tanh(x)
    raise ValueError(name)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def dense(x, w, prefix, activation=None, use_bias=True):
    """tf.layers.Dense on the last axis: reference tacotron/layers.py:96-111 (S1)."""
    y = x @ w[prefix + '/kernel']
    if use_bias:
        y = y + w[prefix + '/bias']
    return _act(y, activation)


def conv1d_same(x, kernel, bias):
    """tf.layers.conv1d(padding='SAME', strides=1): cross-correlation, total padding
    k-1 with the smaller half on the left (S2).  x (B,T,Cin), kernel (k,Cin,Cout)."""
    k = kernel.shape[0]
    B, T, _ = x.shape
    pad_l = (k - 1) // 2
    pad_r = k - 1 - pad_l
    xp = np.pad(x, ((0, 0), (pad_l, pad_r), (0, 0)))
    y = np.zeros((B, T, kernel.shape[2]), dtype=x.dtype)
    for j in range(k):
        y += xp[:, j:j + T, :] @ kernel[j]
    return y + bias


def batch_norm_inference(x, w, prefix, scale):
    """tf.layers.batch_normalization(training=False): (x-mean)*rsqrt(var+eps)*gamma+beta,
    eps=1e-3; gamma absent when scale=False (S3).  reference layers.py:380-383,440-443."""
    inv = 1.0 / np.sqrt(w[prefix + '/moving_variance'] + BN_EPS)
    if scale:
        inv = inv * w[prefix + '/gamma']
    return (x - w[prefix + '/moving_mean']) * inv + w[prefix + '/beta']


def max_pool_2_1_same(x):
    """tf.layers.max_pooling1d(pool_size=2, strides=1, padding='SAME') (S4):
    y[t] = max(x[t], x[t+1]); the right pad is ignored.  reference layers.py:518-521."""
    y = x.copy()
    y[:, :-1, :] = np.maximum(x[:, :-1, :], x[:, 1:, :])
    return y


def pre_net(x, w, scope, layers):
    """reference layers.py:262-304 with dropout inactive (mode != TRAIN, model.py:119-122)."""
    for i, (units, _drop, act) in enumerate(layers):
        x = dense(x, w, '{}/{}-FC-{}'.format(scope, i + 1, units), act)
    return x


def highway_layer(x, w, scope):
    """reference layers.py:181-258: h*t + x*(1-t), h=relu(xW_H+b), t=sigmoid(xW_T+b)."""
    h = dense(x, w, scope + '/H', 'relu')
    t = dense(x, w, scope + '/T', 'sigmoid')
    return h * t + x * (1.0 - t)


def _bn_name(i):
    return 'batch_normalization' if i == 0 else 'batch_normalization_{}'.format(i)


def gru_cell(x, h, w, scope, cudnn=False):
    """One GRU step (S5 / S5').

    TF GRUCell [TF-1.8]: [r|u] = sigmoid([x;h] W_g + b_g); c = tanh([x; r*h] W_c + b_c);
    h' = u*h + (1-u)*c.  CudnnCompatibleGRUCell: c = tanh(x W_ci + b_ci + r*(h W_ch + b_ch)).
    Selected by force_cudnn at reference layers.py:560-577, model.py:226-229,257-262."""
    units = h.shape[-1]
    g = sigmoid(np.concatenate([x, h], -1) @ w[scope + '/gates/kernel'] + w[scope + '/gates/bias'])
    r, u = g[..., :units], g[..., units:]
    if cudnn:
        ci = x @ w[scope + '/candidate/input_projection/kernel'] + w[scope + '/candidate/input_projection/bias']
        ch = h @ w[scope + '/candidate/hidden_projection/kernel'] + w[scope + '/candidate/hidden_projection/bias']
        c = np.tanh(ci + r * ch)
    else:
        c = np.tanh(np.concatenate([x, r * h], -1) @ w[scope + '/candidate/kernel']
                    + w[scope + '/candidate/bias'])
    return u * h + (1.0 - u) * c


def bi_gru(x, w, scope, units, cudnn=False):
    """bidirectional_dynamic_rnn over the FULL padded length, zero initial state, no
    sequence_length (S6).  reference layers.py:579-592.  Returns (B,T,2*units) = [fw|bw]."""
    B, T, _ = x.shape
    out = np.zeros((B, T, 2 * units), dtype=x.dtype)
    h = np.zeros((B, units), dtype=x.dtype)
    for t in range(T):
        h = gru_cell(x[:, t], h, w, scope + '/fw/gru_cell_fw', cudnn)
        out[:, t, :units] = h
    h = np.zeros((B, units), dtype=x.dtype)
    for t in range(T - 1, -1, -1):
        h = gru_cell(x[:, t], h, w, scope + '/bw/gru_cell_bw', cudnn)
        out[:, t, units:] = h
    return out


def cbhg(x, w, scope, hp, cudnn=False, stages=None):
    """reference layers.py:448-594: bank -> maxpool -> projections -> residual -> lifter
    -> highway x4 -> bi-GRU.  ``stages`` (dict) optionally receives the intermediates."""
    inputs = x
    banks = []
    for k in range(1, hp.n_banks + 1):
        cs = '{}/convolution_banks/conv-{}-{}'.format(scope, k, hp.n_filters)
        y = conv1d_same(inputs, w[cs + '/kernel'], w[cs + '/bias'])
        y = np.maximum(y, 0)  # conv activation first, then BN (layers.py:361-383)
        y = batch_norm_inference(y, w, '{}/convolution_banks/{}'.format(scope, _bn_name(k - 1)), scale=False)
        banks.append(y)
    net = np.concatenate(banks, -1)
    if stages is not None:
        stages['bank'] = net
    net = max_pool_2_1_same(net)
    for i, (filters, ksize, act) in enumerate(hp.projections):
        ps = '{}/projections/{}-conv-{}-{}'.format(scope, i + 1, ksize, filters)
        net = conv1d_same(net, w[ps + '/conv1d/kernel'], w[ps + '/conv1d/bias'])
        net = _act(net, act)
        net = batch_norm_inference(net, w, ps + '/batch_normalization', scale=True)
        if stages is not None:
            stages['proj{}'.format(i + 1)] = net
    net = net + inputs
    net = dense(net, w, scope + '/lifter', 'relu')
    if stages is not None:
        stages['lifter'] = net
    for layer in range(hp.n_highway_layers):
        net = highway_layer(net, w, '{}/highway_network/highway_layer_{}'.format(scope, layer))
    if stages is not None:
        stages['highway'] = net
    out = bi_gru(net, w, scope + '/gru', hp.n_gru_units, cudnn)
    return out


def encoder(ids, w, hp, stages=None):
    """reference tacotron/model.py:124-173.  ids int (B,T_s) -> memory (B,T_s,256)."""
    dt = w['encoder/embedding'].dtype
    emb = w['encoder/embedding'][np.asarray(ids, dtype=np.int64)]
    net = pre_net(emb.astype(dt), w, 'encoder/pre_net', hp.encoder.pre_net_layers)
    if stages is not None:
        stages['prenet'] = net
    return cbhg(net, w, 'encoder', hp.encoder, bool(hp.force_cudnn), stages)


_ATT = 'decoder2/decoder/output_projection_wrapper/multi_rnn_cell/cell_0/attention_wrapper'
_MRC = 'decoder2/decoder/output_projection_wrapper/multi_rnn_cell'


def softmax_lastaxis(s):
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    return e / e.sum(-1, keepdims=True)


def local_luong_monotonic(query, keys, values, time, d, force_gaussian):
    """LocalLuongAttention (MONOTONIC + DOT, scale=False) for one decoder step.

    reference tacotron/attention.py:
      * 263-286  p = min(max(time, D), T_s - (D+1)); start = floor(p) - D; stop = floor(p) + D + 1,
                 clipped to the memory (window_start / window_stop) with zero padding up to 2D+1;
      * 308-328  the (zero-padded) key window is scored with the dot product and soft-maxed over the
                 2D+1 window positions -- padded positions score 0 and DO take softmax mass;
      * 52-71    context = window alignments @ (zero-padded) value window -- NOT gaussian weighted;
      * 73-92    reported alignments: window alignments * exp(-(pos - p)**2 / 2 * (D/2)**2) (the
                 expression as written there, i.e. multiplied by (D/2)^2) when force_gaussian, then
                 zero-padded back to the memory length;
      * 563      `time` is the AttentionWrapperState.time of the step (AdvancedAttentionWrapper).
    Returns (context (B, units), padded alignments (B, T_s))."""
    B, Ts, _ = keys.shape
    dt = keys.dtype
    p = min(max(int(time), d), Ts - (d + 1))
    start, stop = p - d, p + d + 1
    w_start, w_stop = max(0, start), min(Ts, stop)
    pre, post = abs(w_start - start), abs(w_stop - stop)
    kwin = np.pad(keys[:, w_start:w_stop], ((0, 0), (pre, post), (0, 0)))
    vwin = np.pad(values[:, w_start:w_stop], ((0, 0), (pre, post), (0, 0)))
    a = softmax_lastaxis(np.einsum('bd,bwd->bw', query, kwin))
    ctx = np.einsum('bw,bwd->bd', a, vwin)
    if force_gaussian:
        dist = np.arange(w_start, w_stop, dtype=dt) - dt.type(p)
        # shapes differ (and TF fails) when the window was padded; the reference never gets there with
        # T_s >= 2D+1 in monotonic mode
        a = a * np.exp(-(dist ** 2) / 2 * dt.type((d / 2) ** 2))
    full = np.pad(a, ((0, 0), (abs(start), abs(stop - Ts))))
    return ctx, full


def local_luong_predictive(query, keys, values, w_p, v_p, d, force_gaussian):
    """LocalLuongAttention (PREDICTIVE + DOT) for one decoder step: the window centre is predicted per utterance,
    p = T_s * sigmoid(v_p^T tanh(W_p h))  (reference tacotron/attention.py:246-258; tensordot(wp, query, [0, 1])
    transposed is query @ W_p, the second tensordot is tanh(.) @ v_p), the window is [floor(p) - D, floor(p) + D]
    (:273-286) and everything else is the monotonic path (:288-342, 52-92) with the real-valued p in the gaussian.

    Where a window leaves the memory the reference is not well defined: it pads the (2D+1)-wide alignments with
    abs(start) zeros in front and abs(stop - T_s) behind (:294-299), which only adds up to T_s for windows inside
    the memory -- otherwise the rows of the batch get different lengths and tf.stack / the gaussian product
    (range(window_start, window_stop) has fewer than 2D+1 entries) fail at run time.  That case raises here too.
    Returns (context (B, units), padded alignments (B, T_s), p (B,))."""
    B, Ts, _ = keys.shape
    dt = keys.dtype
    p = dt.type(Ts) * sigmoid(np.tanh(query @ w_p) @ v_p)[:, 0]          # (B,)
    centre = np.floor(p).astype(np.int64)
    start, stop = centre - d, centre + d + 1
    if (start < 0).any() or (stop > Ts).any():
        raise ValueError('LocalLuongAttention (predictive): a window leaves the memory; the reference pads such '
                         'windows inconsistently (tacotron/attention.py:288-304) and fails at run time')
    ctx = np.zeros((B, values.shape[-1]), dtype=dt)
    full = np.zeros((B, Ts), dtype=dt)
    for b in range(B):
        kwin, vwin = keys[b, start[b]:stop[b]], values[b, start[b]:stop[b]]
        a = softmax_lastaxis(kwin @ query[b])
        ctx[b] = a @ vwin
        if force_gaussian:
            dist = np.arange(start[b], stop[b], dtype=dt) - p[b]
            a = a * np.exp(-(dist ** 2) / 2 * dt.type((d / 2) ** 2))
        full[b, start[b]:stop[b]] = a
    return ctx, full, p


def decoder(memory, w, hp, n_steps=None, trace=None):
    """reference tacotron/model.py:175-334 in Mode.PREDICT (S7, S8).

    LuongAttention(scale=False, no memory_sequence_length): keys = memory W_mem (no bias),
    values = memory; AttentionWrapper(output_attention=True, attention_layer_size=256,
    cell_input_fn = concat([inputs, attention])); PrenetWrapper (wrappers.py:122-124);
    (hp.attention.mechanism == 'LocalLuongAttention': AdvancedAttentionWrapper + the windowed
    mechanism, model.py:210-245, see local_luong_monotonic);
    two ResidualWrapper(GRUCell); OutputProjectionWrapper(r*80); TacotronInferenceHelper
    (helpers.py:83-110,161-205): GO frame zeros, next input = last 80 of the 400 outputs,
    never finished -> maximum_iterations // reduction steps (model.py:309).

    Returns (reduced_mel (B,S,r*80), alignments (S,B,T_s))."""
    dec = hp.decoder
    cudnn = bool(hp.force_cudnn)
    B, Ts, _ = memory.shape
    dt = memory.dtype
    S = n_steps if n_steps is not None else dec.maximum_iterations // hp.reduction
    A = dec.n_attention_units
    U = dec.n_decoder_gru_units
    att_hp = getattr(hp, 'attention', None)
    local = att_hp is not None and att_hp.mechanism == 'LocalLuongAttention'
    if local and (att_hp.luong_local_mode not in ('monotonic', 'predictive') or att_hp.luong_local_score != 'dot'):
        raise NotImplementedError('LocalLuongAttention: the general / concat scores raise NotImplementedError in '
                                  'the reference too (tacotron/attention.py:436,464)')
    predictive = local and att_hp.luong_local_mode == 'predictive'
    keys = memory @ w['decoder2/memory_layer/kernel']
    x = np.zeros((B, dec.target_size), dtype=dt)
    att = np.zeros((B, A), dtype=dt)
    h_att = np.zeros((B, A), dtype=dt)
    hs = [np.zeros((B, U), dtype=dt) for _ in range(dec.n_gru_layers)]
    outs = np.zeros((B, S, dec.target_size * hp.reduction), dtype=dt)
    aligns = np.zeros((S, B, Ts), dtype=dt)
    for t in range(S):
        cell_in = np.concatenate([x, att], -1)
        p = pre_net(cell_in, w, _ATT + '/pre_net', dec.pre_net_layers)
        h_att = gru_cell(p, h_att, w, _ATT + '/gru_cell', cudnn)
        if predictive:
            score = None
            ctx, a, _ = local_luong_predictive(h_att, keys, memory, w[_ATT + '/local_luong_attention/local_w_p'],
                                               w[_ATT + '/local_luong_attention/local_v_p'],
                                               att_hp.luong_local_window_D, att_hp.luong_force_gaussian)
        elif local:
            score = None
            ctx, a = local_luong_monotonic(h_att, keys, memory, t, att_hp.luong_local_window_D,
                                           att_hp.luong_force_gaussian)
        else:
            score = np.einsum('bd,btd->bt', h_att, keys)
            a = softmax_lastaxis(score)
            ctx = np.einsum('bt,btd->bd', a, memory)
        att = np.concatenate([h_att, ctx], -1) @ w[_ATT + '/attention_layer/kernel']
        y = att
        for i in range(dec.n_gru_layers):
            hs[i] = gru_cell(y, hs[i], w, '{}/cell_{}/gru_cell'.format(_MRC, i + 1), cudnn)
            y = y + hs[i]
        out = y @ w['decoder2/decoder/output_projection_wrapper/kernel'] \
            + w['decoder2/decoder/output_projection_wrapper/bias']
        outs[:, t] = out
        aligns[t] = a
        x = out[:, -dec.target_size:]
        if trace is not None and t == 0:
            trace.update(dict(p=p, h_att=h_att.copy(), score=score, ctx=ctx, att=att.copy(), y=y))
    return outs, aligns


def post_process(mel, w, hp, stages=None):
    """reference tacotron/model.py:336-363, 394-398: post-net CBHG + Dense(1025)."""
    net = cbhg(mel, w, 'post_process', hp.post, bool(hp.force_cudnn), stages)
    if stages is not None:
        stages['gru'] = net
    return dense(net, w, 'dense')


def tacotron_predict(ids, w, hp, n_steps=None):
    """reference tacotron/model.py:365-401 (PREDICT outputs only).

    Returns dict(memory, reduced_mel (B,S,400), mel (B,S*r,80), alignments (S,B,T_s),
    linear (B,S*r,1025))."""
    memory = encoder(ids, w, hp)
    red, aligns = decoder(memory, w, hp, n_steps)
    B = red.shape[0]
    mel = red.reshape(B, -1, hp.n_mels)
    if hp.apply_post_processing:
        linear = post_process(mel, w, hp)
    else:
        linear = dense(mel, w, 'dense')
    return dict(memory=memory, reduced_mel=red, mel=mel, alignments=aligns, linear=linear)


def cast_weights(w, dtype):
    return {k: np.asarray(v, dtype=dtype) for k, v in w.items()}
