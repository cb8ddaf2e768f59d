"""GPU parity of the GEMM / conv1d kernel on its own (tts_debug_gemm) against numpy float64: the implicit im2col
of TF 'SAME' conv1d (reference tacotron/layers.py:361-367, 432-437), the fused max-pool(2,1,SAME)
loader (layers.py:518-521; S4), the tap-inner k order, the XCD-aware tile map (M not a multiple of 8 tiles) and
the split-K path of long-K layers."""
import numpy as np
import pytest

from conftest import pkg, rel_l2

pytestmark = pytest.mark.gpu


def _conv_ref(x, w, ktaps, T, pool):
    """x [M][Cin] = B sequences of length T; w [N][ktaps*Cin] (tap-major k); TF SAME padding."""
    M, Cin = x.shape
    xs = x.reshape(M // T, T, Cin).astype(np.float64)
    if pool:
        nxt = np.concatenate([xs[:, 1:], xs[:, -1:]], 1)
        xs = np.maximum(xs, nxt)
    padl = (ktaps - 1) // 2
    xp = np.pad(xs, ((0, 0), (padl, ktaps - 1 - padl), (0, 0)))
    cols = np.concatenate([xp[:, j:j + T] for j in range(ktaps)], -1)       # [B][T][ktaps*Cin]
    return (cols @ w.astype(np.float64).T).reshape(M, -1)


@pytest.mark.parametrize('B,T,Cin,ktaps,N,pool', [
    (3, 50, 128, 1, 256, 0),        # plain dense, M = 150 (two M tiles, one partly empty)
    (2, 77, 256, 3, 80, 0),         # conv3, N not a multiple of the tile, tap-inner k order (256 % 32 == 0)
    (2, 77, 256, 3, 128, 1),        # ... with the max-pool loader
    (4, 40, 80, 5, 128, 0),         # channel count not a multiple of the tile depth: linear k order
    (5, 30, 2048, 3, 128, 1),       # K = 6144: the split-K shape of the encoder's first projection
    (9, 150, 128, 1, 1025, 0),      # 11 M tiles x 9 N tiles: exercises the XCD tile map with padding
])
def test_gemm_conv_matches_numpy(engine, B, T, Cin, ktaps, N, pool):
    rng = np.random.default_rng(B * 1000 + Cin)
    M = B * T
    x = rng.standard_normal((M, Cin)).astype(np.float32)
    w = (rng.standard_normal((N, ktaps * Cin)) * 0.05).astype(np.float32)
    ref = _conv_ref(x, w, ktaps, T, pool)
    dx, dw = engine.to_device(x), engine.to_device(w)
    dc = engine.empty((M, N))
    hip = pkg('_hip')
    engine._check(engine.lib.tts_debug_gemm(engine.handle, dx.data_ptr(), dw.data_ptr(), dc.data_ptr(), M, N, Cin, ktaps,
                                            T, pool))
    got = dc.to_host()
    e = rel_l2(got, ref)
    print('gemm B={} T={} Cin={} k={} N={} pool={}: rel-L2 {:.2e}'.format(B, T, Cin, ktaps, N, pool, e))
    assert e < 1e-5
    dx.free(); dw.free(); dc.free()


@pytest.mark.parametrize('option', ['gemm_ps', 'gemm_presplit'])
def test_gemm_variant_forms_match_numpy(engine, option):
    """Round 5's two GEMM variants (measured, not faster: profiles/r05_experiment_gemm_presplit.txt) -- producer / consumer
    waves (`gemm_ps`) and pre-split weight images (`gemm_presplit`): the shipped library does not carry their kernels and
    refuses the options; a tools build (-DGEMM_EXPERIMENTS, SSTTS_HIP_LIB) keeps them correct: dense, conv3 with the
    max-pool loader, and the split-K shape."""
    H = pkg('_hip')
    try:
        try:
            engine.set_option(option, 1)
        except H.TtsError as e:
            assert e.code == H.TTS_ERR_UNSUPPORTED and 'GEMM_EXPERIMENTS' in str(e)
            return
        for B, T, Cin, ktaps, N, pool in [(3, 50, 128, 1, 256, 0), (2, 77, 256, 3, 128, 1), (5, 30, 2048, 3, 128, 1), (4, 40, 80, 5, 128, 0)]:
            rng = np.random.default_rng(B * 1000 + Cin)
            M = B * T
            x = rng.standard_normal((M, Cin)).astype(np.float32)
            w = (rng.standard_normal((N, ktaps * Cin)) * 0.05).astype(np.float32)
            ref = _conv_ref(x, w, ktaps, T, pool)
            dx, dw = engine.to_device(x), engine.to_device(w)
            dc = engine.empty((M, N))
            engine._check(engine.lib.tts_debug_gemm(engine.handle, dx.data_ptr(), dw.data_ptr(), dc.data_ptr(), M, N, Cin, ktaps, T, pool))
            assert rel_l2(dc.to_host(), ref) < 1e-5, (option, B, T, Cin, ktaps, N, pool)
            dx.free(); dw.free(); dc.free()
    finally:
        engine.set_option(option, 0)
