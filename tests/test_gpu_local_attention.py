"""GPU parity for the reference's experimental LocalLuongAttention (tacotron/attention.py:109-342,
enabled by model_params.attention.mechanism, params/model.py:112-128) in its default sub-mode
MONOTONIC + DOT: a 2D+1 window of the memory around the decoder step index is scored, the context
comes from the plain window softmax, the reported alignments are gaussian-weighted and zero outside
the window.  Checked end to end against the oracle's restatement."""
import copy

import numpy as np
import pytest

from conftest import pkg, rel_l2
from oracle import tacotron_oracle as O

pytestmark = pytest.mark.gpu


def _setup(hparams, weights, D, gaussian):
    hp = copy.deepcopy(hparams)
    hp.attention.mechanism = 'LocalLuongAttention'
    hp.attention.luong_local_window_D = D
    hp.attention.luong_force_gaussian = gaussian
    eng = pkg().Engine(hp)
    eng.load_weights(weights)          # same variables as the global mechanism (monotonic mode adds none)
    return hp, eng


def _decoder_form(eng, form):
    """form = (persistent_decoder, pd_ws, rows per cluster of the weight-stationary kernel) or just persistent_decoder:
    the launch-per-layer decoder (decoder.hip), decoder_ws.hip with 32 / 16 utterances per cluster (round 6: it scores the
    local window too), decoder_persistent.hip."""
    pd, pd_ws, rows = form if isinstance(form, tuple) else (form, 1, 0)
    eng.set_option('persistent_decoder', pd)
    eng.set_option('pd_ws', pd_ws)
    eng.set_option('debug_hooks', 1)
    eng.set_option('pd_rows', rows)


@pytest.mark.parametrize('B,Ts,S,D,gaussian', [
    (3, 21, 6, 10, True),      # window == memory
    (4, 60, 40, 10, True),     # window slides: steps 0..10 pinned left, then moves with t
    (2, 33, 45, 10, False),    # runs past the end: window pinned at the right edge
    (5, 30, 12, 3, True),      # small D, 7 positions split over 4 attention slices
    (2, 150, 30, 1, False),    # 3 positions: some slices empty
])
@pytest.mark.parametrize('pd', [(0, 1, 0), (2, 1, 32), (2, 1, 16), (2, 0, 0)],
                         ids=['launch-per-layer', 'weight-stationary-32', 'weight-stationary-16', 'streamed-weights'])
def test_local_attention_decoder(hparams, weights, weights64, B, Ts, S, D, gaussian, pd):
    hp, eng = _setup(hparams, weights, D, gaussian)
    _decoder_form(eng, pd)
    try:
        rng = np.random.default_rng(100 * B + D)
        memory = rng.standard_normal((B, Ts, 256)).astype(np.float32) * 0.5
        ref_mel, ref_al = O.decoder(memory.astype(np.float64), weights64, hp, n_steps=S)
        mel, al = eng.decoder_forward(memory, S)
        al = al.to_host()
        e_mel = rel_l2(mel.to_host(), ref_mel)
        e_al = float(np.abs(al - ref_al).max())
        print('local attention B={} Ts={} S={} D={} gaussian={}: mel {:.2e} align {:.2e}'.format(
            B, Ts, S, D, gaussian, e_mel, e_al))
        assert e_mel < 1e-3 and e_al < 1e-4
        # structure: nothing outside the window, window follows the step index
        for t in range(S):
            p = min(max(t, D), Ts - (D + 1))
            outside = np.ones(Ts, bool)
            outside[p - D:p + D + 1] = False
            assert np.all(al[t][:, outside] == 0.0)
        if not gaussian:
            np.testing.assert_allclose(al.sum(-1), 1.0, atol=1e-5)
    finally:
        eng.close()


def test_local_differs_from_global(hparams, weights, engine):
    """Sanity: with Ts > 2D+1 the windowed mechanism is a different computation."""
    hp, eng = _setup(hparams, weights, 4, True)
    try:
        memory = np.random.default_rng(3).standard_normal((2, 40, 256)).astype(np.float32)
        a, _ = eng.decoder_forward(memory, 8)
        b, _ = engine.decoder_forward(memory, 8)
        assert rel_l2(a.to_host(), b.to_host()) > 1e-3
    finally:
        eng.close()


def test_local_window_equal_memory_matches_global_context(hparams, weights, engine):
    """Ts == 2D+1: the window is the whole memory, so the mel output equals the global mechanism's
    (only the reported alignments differ, by the gaussian weighting)."""
    hp, eng = _setup(hparams, weights, 6, True)
    try:
        memory = np.random.default_rng(4).standard_normal((3, 13, 256)).astype(np.float32)
        a, al_l = eng.decoder_forward(memory, 10)
        b, al_g = engine.decoder_forward(memory, 10)
        assert rel_l2(a.to_host(), b.to_host()) < 1e-5
        g = np.exp(-((np.arange(13) - 6.0) ** 2) / 2 * 3.0 ** 2)
        np.testing.assert_allclose(al_l.to_host(), al_g.to_host() * g, atol=1e-6)
    finally:
        eng.close()


@pytest.mark.parametrize('gaussian', [True, False])
def test_local_short_memory_is_refused(hparams, weights, weights64, gaussian):
    """T_s < 2D+1 in monotonic mode.  The reference computes a CONTEXT for it (the window is zero-padded in front,
    tacotron/attention.py:288-321) but not alignments it can carry: the window's 2D+1 alignments are padded with
    abs(start) + abs(stop - T_s) = 2D+1 - T_s zeros (:294-299, :85-92) to 4D+2 - T_s != T_s entries -- the state of the
    attention wrapper changes shape between steps -- and with luong_force_gaussian the 2D+1 alignments meet T_s gaussian
    weights (:73-80).  TensorFlow fails there; the oracle's restatement raises on the same shapes (both settings), and the
    library refuses the call instead of inventing alignments: same error behaviour."""
    hp, eng = _setup(hparams, weights, 10, gaussian)
    try:
        ids = np.random.default_rng(3).integers(2, 39, (1, 20)).astype(np.int32)   # 20 < 2D+1 = 21
        with pytest.raises(ValueError):
            O.tacotron_predict(ids, weights64, hp, n_steps=3)
        memory = np.zeros((1, 20, 256), np.float32)
        for pd in (0, 2):
            _decoder_form(eng, pd)
            with pytest.raises(pkg('_hip').TtsError) as ei:
                eng.decoder_forward(memory, 3)
            assert ei.value.code == -5
    finally:
        eng.close()


def test_local_end_to_end_synthesize(hparams, weights, weights64):
    """tts_synthesize with the local mechanism: spectrograms against the oracle."""
    hp, eng = _setup(hparams, weights, 10, True)
    try:
        rng = np.random.default_rng(9)
        ids = rng.integers(2, 39, (2, 30)).astype(np.int32)
        ids[:, -1] = 1
        ref = O.tacotron_predict(ids, weights64, hp, n_steps=8)
        out = eng.synthesize(ids, n_steps=8, n_iter=2, ref_db=35.66, max_db=100.0, power=1.3,
                             win_length=1102, hop_length=275, want_mel=True, want_linear=True,
                             want_alignments=True)
        assert rel_l2(out['mel'].to_host(), ref['mel']) < 1e-3
        assert rel_l2(out['linear'].to_host(), ref['linear']) < 1e-3
        assert float(np.abs(out['alignments'].to_host() - ref['alignments']).max()) < 1e-4
    finally:
        eng.close()


def _setup_predictive(hparams, D, gaussian, seed=11, vp_scale=1.0, vp_shift=0.0):
    hp = copy.deepcopy(hparams)
    hp.attention.mechanism = 'LocalLuongAttention'
    hp.attention.luong_local_mode = 'predictive'
    hp.attention.luong_local_window_D = D
    hp.attention.luong_force_gaussian = gaussian
    w = pkg('tacotron.weights').synthetic_weights(seed, hp)
    vp = 'decoder2/decoder/output_projection_wrapper/multi_rnn_cell/cell_0/attention_wrapper/local_luong_attention/local_v_p'
    w[vp] = (w[vp] * vp_scale + vp_shift).astype(np.float32)
    eng = pkg().Engine(hp)
    eng.load_weights(w)
    return hp, eng, w


@pytest.mark.parametrize('B,Ts,S,D,gaussian,vp_scale', [(3, 60, 12, 10, True, 1.0), (4, 90, 20, 5, False, 4.0),
                                                        (2, 150, 9, 10, True, 8.0)])
@pytest.mark.parametrize('pd', [(0, 1, 0), (2, 1, 32), (2, 1, 16), (2, 0, 0)],
                         ids=['launch-per-layer', 'weight-stationary-32', 'weight-stationary-16', 'streamed-weights'])
def test_predictive_local_attention_decoder(hparams, B, Ts, S, D, gaussian, vp_scale, pd):
    """LocalLuongAttention in PREDICTIVE mode (reference tacotron/attention.py:246-258): the window centre
    p = T_s sigmoid(v_p^T tanh(W_p h)) is predicted per utterance and step; larger v_p spreads the centres."""
    hp, eng, w = _setup_predictive(hparams, D, gaussian, vp_scale=vp_scale)
    _decoder_form(eng, pd)
    try:
        assert len(eng.manifest()) == len(pkg('tacotron.weights').manifest(hp))
        rng = np.random.default_rng(7 * B + D)
        memory = rng.standard_normal((B, Ts, 256)).astype(np.float32) * 0.5
        w64 = {k: v.astype(np.float64) for k, v in w.items()}
        ref_mel, ref_al = O.decoder(memory.astype(np.float64), w64, hp, n_steps=S)
        mel, al = eng.decoder_forward(memory, S)
        al = al.to_host()
        e_mel, e_al = rel_l2(mel.to_host(), ref_mel), float(np.abs(al - ref_al).max())
        centres = np.array([[np.flatnonzero(al[t, b]).mean() for b in range(B)] for t in range(S)])
        print('predictive local attention B={} Ts={} S={} D={}: mel {:.2e} align {:.2e}; window centres {:.1f}..{:.1f}'.format(
            B, Ts, S, D, e_mel, e_al, centres.min(), centres.max()))
        assert e_mel < 1e-3 and e_al < 1e-4
        assert np.all((al != 0).sum(-1) <= 2 * D + 1)
        if not gaussian:
            np.testing.assert_allclose(al.sum(-1), 1.0, atol=1e-5)
    finally:
        eng.close()


@pytest.mark.parametrize('pd', [(0, 1, 0), (2, 1, 32), (2, 1, 16), (2, 0, 0)],
                         ids=['launch-per-layer', 'weight-stationary-32', 'weight-stationary-16', 'streamed-weights'])
def test_predictive_window_leaving_the_memory_is_an_error(hparams, pd):
    """Where the predicted window leaves the memory the reference's padding arithmetic (attention.py:288-304)
    breaks and TensorFlow fails at run time; the library reports TTS_ERR_UNSUPPORTED, the oracle raises."""
    # T_s = 2D+1: only floor(p) == D keeps the window inside, and a large v_p spreads p = 21 sigmoid(.) well beyond
    hp, eng, w = _setup_predictive(hparams, 10, True, vp_scale=8.0)
    _decoder_form(eng, pd)
    try:
        memory = np.random.default_rng(1).standard_normal((2, 21, 256)).astype(np.float32)
        w64 = {k: v.astype(np.float64) for k, v in w.items()}
        with pytest.raises(ValueError):
            O.decoder(memory.astype(np.float64), w64, hp, n_steps=3)
        with pytest.raises(pkg('_hip').TtsError) as ei:
            eng.decoder_forward(memory, 3)
        assert ei.value.code == -5
    finally:
        eng.close()
