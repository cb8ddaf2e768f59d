"""Outputs of the REFERENCE'S OWN functions (tests/golden/conversion.{npz,json}, produced by
tests/golden/make_conversion_fixtures.py, which executes the function bodies of reference audio/conversion.py:5-136
and tacotron/inference.py:22-27) against

  * the oracle's restatements (CPU): this is what pins oracle/audio_oracle.py's conversion functions and the
    de-normalisation chain of rows a17 / a18 by the reference's arithmetic instead of by a second restatement;
  * the host-side mirror (CPU): ms_to_samples, pad_sentence;
  * the HIP kernels through the C ABI (-m gpu): tts_db_convert modes 0..3 and tts_denorm_power.

Tolerances are float32 rounding of the respective formula, written at each assertion."""
import json
import os

import numpy as np
import pytest

from conftest import pkg
from oracle import audio_oracle as A

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def fx():
    return np.load(os.path.join(GOLD, 'conversion.npz')), json.load(open(os.path.join(GOLD, 'conversion.json')))


def test_fixture_provenance(fx):
    arrays, meta = fx
    # the line ranges the generator cut out are the ones the oracle and the header cite
    assert meta['reference_lines']['decibel_to_magnitude'] == [32, 53]
    assert meta['reference_lines']['inv_normalize_decibel'] == [81, 102]
    assert meta['reference_lines']['pad_sentence'] == [22, 27]
    assert meta['constants'] == {'mel_mag_ref_db': 6.02, 'mel_mag_max_db': 99.89, 'magnitude_power': 1.3}
    # float32 in -> float32 out for every reference function (numpy keeps the array's type against python scalars)
    assert set(meta['dtypes'].values()) == {'float32'}


def test_oracle_conversions_equal_the_reference_bit_for_bit(fx):
    arrays, meta = fx
    assert np.array_equal(A.magnitude_to_decibel(arrays['m2d_in']), arrays['m2d_out'])
    assert np.array_equal(A.decibel_to_magnitude(arrays['d2m_in']), arrays['d2m_out'])
    for i, (r, m) in enumerate(meta['db_pairs']):
        assert np.array_equal(A.normalize_decibel(arrays['norm%d_in' % i], r, m), arrays['norm%d_out' % i])
        assert np.array_equal(A.inv_normalize_decibel(arrays['inv%d_in' % i], r, m), arrays['inv%d_out' % i])
    c = meta['constants']
    got = A.linear_to_magnitude(arrays['chain_in'], c['mel_mag_ref_db'], c['mel_mag_max_db'], c['magnitude_power'])
    assert got.dtype == np.float32 and np.array_equal(got, arrays['chain_pow'])
    assert np.array_equal(A.inv_normalize_decibel(arrays['chain_in'].T, c['mel_mag_ref_db'], c['mel_mag_max_db']), arrays['chain_db'])
    for case in meta['decibel_to_magnitude_assertion']:
        assert case['raises'] == 'AssertionError'
        with pytest.raises(AssertionError) as e:
            A.decibel_to_magnitude(np.float32(case['input']))
        assert str(e.value) == case['message']
    for case in meta['scalars']['ms_to_samples']:
        assert A.ms_to_samples(case['ms'], case['sr']) == case['out']


def test_host_mirror_equals_the_reference(fx):
    arrays, meta = fx
    conv = pkg('audio.conversion')
    inf = pkg('tacotron.inference')
    for case in meta['scalars']['ms_to_samples']:
        out = conv.ms_to_samples(case['ms'], case['sr'])
        assert out == case['out'] and isinstance(out, int)
    for case in meta['scalars']['samples_to_ms']:
        assert conv.samples_to_ms(case['samples'], case['sr']) == case['out']
    win = meta['scalars']['model_win']
    assert (win['win_len'], win['win_hop']) == (1102, 275)
    hp = pkg('tacotron.params').model_params
    assert (hp.win_len, hp.win_hop, hp.sampling_rate) == (win['win_len_ms'], win['win_hop_ms'], win['sampling_rate'])
    assert pkg('tacotron.params').dataset_params.vocabulary_dict['pad'] == meta['pad_token']
    for case in meta['pad_sentence']:
        out = inf.pad_sentence(np.int32(case['sentence']), case['max_len'])
        assert out.tolist() == case['out'] and str(out.dtype) == case['dtype']


@pytest.mark.gpu
def test_hip_db_convert_against_the_reference(engine, fx):
    arrays, meta = fx
    # mode 0: 20 log10(max(1e-5, x)): the kernel rounds the float64 result once; the reference computes in float32 (one
    # rounding in log10, one in the product): <= 2 ulp of the value apart (the inputs reach 3e38 = 770 dB)
    got = engine.db_convert(arrays['m2d_in'], 0)
    assert (np.abs(got - arrays['m2d_out']) <= 2.4e-7 * np.maximum(np.abs(arrays['m2d_out']), 64.0)).all()
    # mode 1: 10 ** (x / 20) as exp2: relative error of an fp32 exp2 with a 7-bit exponent argument
    got = engine.db_convert(arrays['d2m_in'], 1)
    assert np.abs(got / arrays['d2m_out'] - 1.0).max() <= 4e-6
    for i, (r, m) in enumerate(meta['db_pairs']):
        got = engine.db_convert(arrays['norm%d_in' % i], 2, r, m)
        ref = arrays['norm%d_out' % i]
        assert np.abs(got - ref).max() <= 3e-7                                    # values in [0, 1]: 2 ulp
        # ... and the same clip decisions away from the two edges (AT an edge the reference's float32 expression may
        # land one rounding above 0 or below 1 where the kernel's lands on it)
        inner = (ref > 3e-7) & (ref < 1.0 - 3e-7)
        assert np.all((got[inner] > 0.0) & (got[inner] < 1.0)) and got.min() >= 0.0 and got.max() <= 1.0
        got = engine.db_convert(arrays['inv%d_in' % i], 3, r, m)
        assert np.abs(got - arrays['inv%d_out' % i]).max() <= 2e-5                # dB values up to ~110: 2 ulp
    with pytest.raises(AssertionError):
        pkg('audio.conversion').decibel_to_magnitude(np.float32(meta['decibel_to_magnitude_assertion'][0]['input']), engine=engine)


@pytest.mark.gpu
def test_hip_denorm_power_against_the_reference(engine, fx):
    """tts_denorm_power = inference.py:93-101 + :175 in one kernel, against the chain as the reference's functions ran it."""
    arrays, meta = fx
    c = meta['constants']
    lin = arrays['chain_in'][None]                                                # (1, T, F)
    mag1 = engine.denorm_power(lin, c['mel_mag_ref_db'], c['mel_mag_max_db'], 1.0).to_host()[0]
    magp = engine.denorm_power(lin, c['mel_mag_ref_db'], c['mel_mag_max_db'], c['magnitude_power']).to_host()[0]
    assert mag1.shape == arrays['chain_mag'].shape == (1025, lin.shape[1])
    assert np.abs(mag1 / arrays['chain_mag'] - 1.0).max() <= 5e-6                 # one fp32 exp2 of |x| < 17
    assert np.abs(magp / arrays['chain_pow'] - 1.0).max() <= 8e-6                 # ... of |x| < 22


# ---- hyper-parameters: the mirror's defaults against the reference's own HParams calls (tests/golden/params.json, read
# out of tacotron/params/{model,inference,dataset}.py and datasets/lj_speech.py with ast by make_params_fixtures.py)
_NAMES = {'tf.nn.relu': 'relu', 'LuongAttention': 'LuongAttention', 'LocalLuongAttention': 'LocalLuongAttention',
          'AttentionScore.DOT': 'dot', 'AttentionMode.MONOTONIC': 'monotonic', 'AttentionMode.PREDICTIVE': 'predictive',
          'LJSpeechDatasetHelper': 'LJSpeechDatasetHelper'}


def _norm(v):
    """reference value as stored in the fixture -> the form the mirror keeps (names as strings, tuples as lists)"""
    if isinstance(v, dict) and set(v) == {'name'}:
        return _NAMES[v['name']]
    if isinstance(v, dict):
        return {k: _norm(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_norm(x) for x in v]
    return v


def _plain(v):
    import dataclasses
    if dataclasses.is_dataclass(v) and not isinstance(v, type):
        return {f.name: _plain(getattr(v, f.name)) for f in dataclasses.fields(v)}
    if isinstance(v, (tuple, list)):
        return [_plain(x) for x in v]
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    return v


def test_hyper_parameters_equal_the_reference():
    import json
    ref = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'params.json')))
    P = pkg('tacotron.params')
    mine = _plain(P.ModelParams())
    theirs = _norm(ref['model_params'])
    assert set(mine) == set(theirs), set(mine) ^ set(theirs)
    # the one documented difference: the reference ships force_cudnn=True (CudnnGRU, no CPU kernel in TF 1.8); the parity
    # target named by BASELINE.json is the TF CPU path = force_cudnn=False (params.py docstring)
    assert theirs.pop('force_cudnn') is True and mine.pop('force_cudnn') is False
    assert mine == theirs, {k: (mine[k], theirs[k]) for k in mine if mine[k] != theirs[k]}
    assert _plain(P.InferenceParams()) == _norm(ref['inference_params'])
    d_mine, d_ref = _plain(P.DatasetParams()), _norm(ref['dataset_params'])
    assert d_mine.pop('dataset_loader').__name__ in ('LJSpeechConstants', 'LJSpeechDatasetHelper') and d_ref.pop('dataset_loader') == 'LJSpeechDatasetHelper'
    assert d_mine == d_ref
    for k, v in ref['lj_speech_constants'].items():
        assert getattr(P.LJSpeechConstants, k) == v, k
    # ... and what the C ABI's default configuration says (tts_default_config) is what those parameters say
    import ctypes
    H = pkg('_hip')
    if os.path.exists(H.LIB_PATH):
        lib = ctypes.CDLL(H.LIB_PATH)
        cfg = H.TtsConfig()
        lib.tts_default_config(ctypes.byref(cfg))
        m = ref['model_params']
        assert (cfg.vocabulary_size, cfg.embedding_size, cfg.n_mels, cfg.reduction, cfg.n_fft) == \
               (m['vocabulary_size'], m['encoder']['embedding_size'], m['n_mels'], m['reduction'], m['n_fft'])
        assert [cfg.enc_prenet_units[0], cfg.enc_prenet_units[1]] == [l[0] for l in m['encoder']['pre_net_layers']]
        assert [cfg.dec_prenet_units[0], cfg.dec_prenet_units[1]] == [l[0] for l in m['decoder']['pre_net_layers']]
        assert (cfg.enc_n_banks, cfg.enc_n_filters, cfg.post_n_banks, cfg.post_n_filters) == \
               (m['encoder']['n_banks'], m['encoder']['n_filters'], m['post']['n_banks'], m['post']['n_filters'])
        assert [cfg.enc_proj_filters[0], cfg.enc_proj_filters[1]] == [p[0] for p in m['encoder']['projections']]
        assert [cfg.post_proj_filters[0], cfg.post_proj_filters[1]] == [p[0] for p in m['post']['projections']]
        assert (cfg.n_highway_layers, cfg.n_highway_units, cfg.n_gru_units) == \
               (m['encoder']['n_highway_layers'], m['encoder']['n_highway_units'], m['encoder']['n_gru_units'])
        assert (cfg.n_attention_units, cfg.n_decoder_gru_units, cfg.n_decoder_gru_layers) == \
               (m['decoder']['n_attention_units'], m['decoder']['n_decoder_gru_units'], m['decoder']['n_gru_layers'])
        assert cfg.luong_local_window_d == m['attention']['luong_local_window_D'] and cfg.luong_force_gaussian == 1
