"""GPU parity on edge shapes: single token / single frame, tile-edge batch sizes, empty attention slices,
the shortest signal Griffin-Lim accepts, chunk-boundary frame counts, and argument validation."""
import numpy as np
import pytest

from conftest import pkg, rel_l2
from oracle import audio_oracle as A
from oracle import tacotron_oracle as O

pytestmark = pytest.mark.gpu
N_FFT, WIN, HOP = 2048, 1102, 275


@pytest.mark.parametrize('B,Ts', [(1, 1), (1, 2), (65, 3), (2, 129)])
def test_encoder_edge_shapes(engine, hparams, weights64, B, Ts):
    rng = np.random.default_rng(B * 100 + Ts)
    ids = rng.integers(0, 39, (B, Ts)).astype(np.int32)     # includes pad (0) and EOS (1) ids anywhere
    ref = O.encoder(ids, weights64, hparams)
    got = engine.encoder_forward(ids).to_host()
    assert got.shape == ref.shape and rel_l2(got, ref) < 1e-3


@pytest.mark.parametrize('B,Ts,S', [(1, 1, 1), (1, 2, 3), (3, 3, 2), (65, 5, 2), (2, 130, 2)])
def test_decoder_edge_shapes(engine, hparams, weights64, B, Ts, S):
    """Ts < 4 leaves some of the 4 attention slices empty; B = 65 crosses the 16-row tile edge."""
    rng = np.random.default_rng(B * 1000 + Ts * 10 + S)
    memory = rng.standard_normal((B, Ts, 256)).astype(np.float32)
    ref_mel, ref_al = O.decoder(memory.astype(np.float64), weights64, hparams, n_steps=S)
    mel, al = engine.decoder_forward(memory, S)
    assert rel_l2(mel.to_host(), ref_mel) < 1e-3
    assert np.abs(al.to_host() - ref_al).max() < 1e-4
    mel2, none = engine.decoder_forward(memory, S, want_alignments=False)
    assert none is None and np.array_equal(mel2.to_host(), mel.to_host())


@pytest.mark.parametrize('B,T', [(1, 1), (1, 2), (2, 129)])
def test_postnet_edge_shapes(engine, hparams, weights64, B, T):
    rng = np.random.default_rng(B * 7 + T)
    mel = rng.random((B, T, 80)).astype(np.float32)
    ref = O.post_process(mel.astype(np.float64), weights64, hparams)
    got = engine.postnet_forward(mel).to_host()
    assert got.shape == ref.shape == (B, T, 1025) and rel_l2(got, ref) < 1e-3


@pytest.mark.parametrize('T', [5, 6, 31, 32, 33, 64, 65])
def test_griffin_lim_chunk_boundaries(engine, T):
    rng = np.random.default_rng(T)
    n = HOP * (T - 1)
    y = (0.2 * np.sin(2 * np.pi * 300 * np.arange(n) / 22050) + 0.05 * rng.standard_normal(n)).astype(np.float32)
    mag = np.abs(A.stft(y, N_FFT, HOP, WIN)).astype(np.float32)
    assert mag.shape == (1025, T)
    init = rng.random((1, 1025, T)).astype(np.float32)
    wav, mse = engine.griffin_lim(mag[None], 2, WIN, HOP, N_FFT, init_phase=init)
    ref_wav, ref_mse = A.griffin_lim_v2(mag, WIN, HOP, N_FFT, 2, init_phase=init[0])
    assert wav.shape == (1, n)
    assert rel_l2(wav.to_host()[0], ref_wav) < 1e-4
    assert abs(mse.to_host()[0] - ref_mse) <= 1e-3 * ref_mse


def test_griffin_lim_rejects_too_short_signals(engine):
    sstts = pkg()
    with pytest.raises(sstts.TtsError) as e:
        engine.griffin_lim(np.ones((1, 1025, 4), np.float32), 1, WIN, HOP, N_FFT)     # 825 samples <= n_fft / 2
    assert e.value.code == -1
    with pytest.raises(sstts.TtsError):
        engine.griffin_lim(np.ones((1, 1025, 8), np.float32), 1, 4096, HOP, N_FFT)    # window longer than n_fft


def test_generic_window_and_hop(engine):
    """A window/hop other than the reference's 1102/275 takes the run-time-parameter instantiation."""
    rng = np.random.default_rng(9)
    win, hop, T = 1024, 256, 40
    n = hop * (T - 1)
    y = rng.standard_normal(n).astype(np.float32) * 0.1
    mag = np.abs(A.stft(y, N_FFT, hop, win)).astype(np.float32)
    init = rng.random((2, 1025, T)).astype(np.float32)
    wav, mse = engine.griffin_lim(np.stack([mag, 0.5 * mag]), 3, win, hop, N_FFT, init_phase=init)
    for b, m in enumerate((mag, 0.5 * mag)):
        ref_wav, ref_mse = A.griffin_lim_v2(m, win, hop, N_FFT, 3, init_phase=init[b])
        assert rel_l2(wav.to_host()[b], ref_wav) < 3e-4
        assert abs(mse.to_host()[b] - ref_mse) <= 1e-3 * ref_mse


def test_argument_validation(engine):
    sstts = pkg()
    lib, h = engine.lib, engine.handle
    assert lib.tts_encoder_forward(h, None, 1, 1, None) == -1
    assert lib.tts_encoder_forward(h, engine.to_device(np.zeros((1, 1), np.int32)).ptr, 0, 1, engine.empty((1,)).ptr) == -1
    assert lib.tts_decoder_forward(h, None, 1, 1, 1, None, None) == -1
    assert lib.tts_postnet_forward(h, None, 1, 1, None) == -1
    assert b'bad arguments' in lib.tts_last_error(h)
    with pytest.raises(sstts.TtsError):
        engine.set_option('no_such_option', 1)
    assert lib.tts_profile_get(h, b'no_such_stage', None, None) == -1
