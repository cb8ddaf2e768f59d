"""Weight manifest, flat-blob (de)serialisation and a seeded synthetic initialiser.

The manifest keys follow the TensorFlow variable scopes the reference creates
(read off the ``variable_scope`` / ``name=`` arguments; no checkpoint ships with
the reference, README.md:357-361):

  * encoder            reference tacotron/model.py:143-173, layers.py:262-594
  * decoder2           reference tacotron/model.py:191-277, wrappers.py:122
  * post_process       reference tacotron/model.py:350-361
  * dense              reference tacotron/model.py:394-398 (serve.py:110 'dense/BiasAdd:0')

Layouts are TensorFlow's: Dense kernel (in, out); conv1d kernel (k, in, out);
GRUCell ``gates/kernel`` (in + units, 2*units) with columns [r | u] and rows
[input ; state]; ``candidate/kernel`` (in + units, units).
With ``force_cudnn`` (CudnnCompatibleGRUCell) the candidate is split into
``candidate/input_projection`` and ``candidate/hidden_projection``.
"""
from collections import OrderedDict

import numpy as np

from .params import ModelParams

_ATT = 'decoder2/decoder/output_projection_wrapper/multi_rnn_cell/cell_0/attention_wrapper'
_MRC = 'decoder2/decoder/output_projection_wrapper/multi_rnn_cell'


def _bn_name(i):
    return 'batch_normalization' if i == 0 else 'batch_normalization_{}'.format(i)


def _gru_entries(m, scope, n_in, units, cudnn):
    m[scope + '/gates/kernel'] = (n_in + units, 2 * units)
    m[scope + '/gates/bias'] = (2 * units,)
    if cudnn:
        m[scope + '/candidate/input_projection/kernel'] = (n_in, units)
        m[scope + '/candidate/input_projection/bias'] = (units,)
        m[scope + '/candidate/hidden_projection/kernel'] = (units, units)
        m[scope + '/candidate/hidden_projection/bias'] = (units,)
    else:
        m[scope + '/candidate/kernel'] = (n_in + units, units)
        m[scope + '/candidate/bias'] = (units,)


def _cbhg_entries(m, scope, n_in, hp, cudnn):
    for k in range(1, hp.n_banks + 1):
        m['{}/convolution_banks/conv-{}-{}/kernel'.format(scope, k, hp.n_filters)] = (k, n_in, hp.n_filters)
        m['{}/convolution_banks/conv-{}-{}/bias'.format(scope, k, hp.n_filters)] = (hp.n_filters,)
    for i in range(hp.n_banks):
        for v in ('beta', 'moving_mean', 'moving_variance'):
            m['{}/convolution_banks/{}/{}'.format(scope, _bn_name(i), v)] = (hp.n_filters,)
    c_in = hp.n_banks * hp.n_filters
    for i, (filters, ksize, _act) in enumerate(hp.projections):
        ps = '{}/projections/{}-conv-{}-{}'.format(scope, i + 1, ksize, filters)
        m[ps + '/conv1d/kernel'] = (ksize, c_in, filters)
        m[ps + '/conv1d/bias'] = (filters,)
        for v in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            m[ps + '/batch_normalization/' + v] = (filters,)
        c_in = filters
    m[scope + '/lifter/kernel'] = (c_in, hp.n_highway_units)
    m[scope + '/lifter/bias'] = (hp.n_highway_units,)
    for layer in range(hp.n_highway_layers):
        for g in ('H', 'T'):
            hs = '{}/highway_network/highway_layer_{}/{}'.format(scope, layer, g)
            m[hs + '/kernel'] = (hp.n_highway_units, hp.n_highway_units)
            m[hs + '/bias'] = (hp.n_highway_units,)
    for d in ('fw', 'bw'):
        _gru_entries(m, '{}/gru/{}/gru_cell_{}'.format(scope, d, d), hp.n_highway_units, hp.n_gru_units, cudnn)


def manifest(hp=None):
    """Ordered {name: shape} of every inference-time variable."""
    hp = hp or ModelParams()
    cudnn = bool(hp.force_cudnn)
    m = OrderedDict()
    enc, dec = hp.encoder, hp.decoder
    m['encoder/embedding'] = (hp.vocabulary_size, enc.embedding_size)
    n_in = enc.embedding_size
    for i, (units, _d, _a) in enumerate(enc.pre_net_layers):
        m['encoder/pre_net/{}-FC-{}/kernel'.format(i + 1, units)] = (n_in, units)
        m['encoder/pre_net/{}-FC-{}/bias'.format(i + 1, units)] = (units,)
        n_in = units
    _cbhg_entries(m, 'encoder', n_in, enc, cudnn)

    mem = 2 * enc.n_gru_units
    att = dec.n_attention_units
    m['decoder2/memory_layer/kernel'] = (mem, att)
    n_in = dec.target_size + att
    for i, (units, _d, _a) in enumerate(dec.pre_net_layers):
        m['{}/pre_net/{}-FC-{}/kernel'.format(_ATT, i + 1, units)] = (n_in, units)
        m['{}/pre_net/{}-FC-{}/bias'.format(_ATT, i + 1, units)] = (units,)
        n_in = units
    _gru_entries(m, _ATT + '/gru_cell', n_in, att, cudnn)
    m[_ATT + '/attention_layer/kernel'] = (att + mem, att)
    a_hp = getattr(hp, 'attention', None)
    if a_hp is not None and a_hp.mechanism == 'LocalLuongAttention' and a_hp.luong_local_mode == 'predictive':
        # tf.get_variable calls inside LocalLuongAttention.__call__ (reference tacotron/attention.py:247-250)
        m[_ATT + '/local_luong_attention/local_v_p'] = (att, 1)
        m[_ATT + '/local_luong_attention/local_w_p'] = (att, att)
    for i in range(dec.n_gru_layers):
        _gru_entries(m, '{}/cell_{}/gru_cell'.format(_MRC, i + 1),
                     att if i == 0 else dec.n_decoder_gru_units, dec.n_decoder_gru_units, cudnn)
    m['decoder2/decoder/output_projection_wrapper/kernel'] = (dec.n_decoder_gru_units,
                                                               dec.target_size * hp.reduction)
    m['decoder2/decoder/output_projection_wrapper/bias'] = (dec.target_size * hp.reduction,)

    # reference tacotron/model.py:388-398: without the post-processing CBHG the final Dense takes the mel frames
    if hp.apply_post_processing:
        _cbhg_entries(m, 'post_process', hp.n_mels, hp.post, cudnn)
    m['dense/kernel'] = (2 * hp.post.n_gru_units if hp.apply_post_processing else hp.n_mels, 1 + hp.n_fft // 2)
    m['dense/bias'] = (1 + hp.n_fft // 2,)
    return m


def n_parameters(hp=None):
    return int(sum(int(np.prod(s)) for s in manifest(hp).values()))


def synthetic_weights(seed=0, hp=None, dtype=np.float32):
    """Seeded random-init weights of the reference architecture.

    glorot-normal kernels; biases N(0, 0.05) except GRU gate bias 1 + N(0, 0.05)
    (TF GRUCell initialises it to 1) and highway T bias -1 + N(0, 0.05)
    (reference layers.py:255); BN beta ~ N(0, .1), gamma ~ U(.8, 1.2),
    mean ~ N(0, .1), variance ~ U(.5, 1.5); glorot-uniform embedding.
    """
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in manifest(hp).items():
        leaf = name.rsplit('/', 1)[1]
        if name == 'encoder/embedding':
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            w = rng.uniform(-lim, lim, shape)
        elif leaf in ('kernel', 'local_w_p', 'local_v_p'):
            fan_out = shape[-1]
            fan_in = int(np.prod(shape[:-1]))
            w = rng.normal(0.0, np.sqrt(2.0 / (fan_in + fan_out)), shape)
        elif leaf == 'bias':
            w = rng.normal(0.0, 0.05, shape)
            if name.endswith('gates/bias'):
                w += 1.0
            elif '/T/' in name:
                w -= 1.0
        elif leaf == 'beta' or leaf == 'moving_mean':
            w = rng.normal(0.0, 0.1, shape)
        elif leaf == 'gamma':
            w = rng.uniform(0.8, 1.2, shape)
        elif leaf == 'moving_variance':
            w = rng.uniform(0.5, 1.5, shape)
        else:
            raise KeyError(name)
        out[name] = np.ascontiguousarray(w, dtype=dtype)
    return out


def pack_blob(weights, hp=None):
    """Concatenate the weights in manifest order into one flat float32 blob."""
    m = manifest(hp)
    parts = []
    for name, shape in m.items():
        w = np.asarray(weights[name], dtype=np.float32)
        if tuple(w.shape) != tuple(shape):
            raise ValueError('weight {} has shape {}, manifest says {}'.format(name, w.shape, shape))
        parts.append(w.reshape(-1))
    return np.concatenate(parts)


def unpack_blob(blob, hp=None):
    """Inverse of :func:`pack_blob`; returns views into ``blob``."""
    m = manifest(hp)
    blob = np.asarray(blob, dtype=np.float32).reshape(-1)
    out = OrderedDict()
    off = 0
    for name, shape in m.items():
        n = int(np.prod(shape))
        out[name] = blob[off:off + n].reshape(shape)
        off += n
    if off != blob.size:
        raise ValueError('blob has {} floats, manifest needs {}'.format(blob.size, off))
    return out
