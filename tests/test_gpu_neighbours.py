"""Every stage of the library beside MFMA GEMM launches of ANOTHER handle (another set of streams): same bits as alone.

Round 6 found the general Griffin-Lim kernels (csrc/griffin_lim_generic.hip) storing wrong values -- the low dword of a packed-f32
result in lanes 48-63 of one wave, single frames of a waveform -- whenever waves of gemm_f32_kernel shared their compute unit:
129 of 200 calls, none alone (profiles/r06_experiment_packed_f32_beside_mfma.txt).  Built without packed-f32 instructions they
are right 3100 of 3100 times.  The pipelined tts_synthesize puts exactly such neighbours side by side (post-net and encoder GEMMs
of one call, Griffin-Lim of the call before), so every stage is run here under that neighbour and compared bit for bit with its
quiet result.  Reference stages: tacotron/model.py:93-160 (encoder), :205-324 (decoder), :326-391 (post-net),
audio/synthesis.py:43-125 (Griffin-Lim)."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu
F = 1025


def _ids(B, Ts, seed):
    rng = np.random.default_rng(seed)
    ids = np.zeros((B, Ts), np.int32)
    for b in range(B):
        L = int(np.clip(round(rng.normal(Ts * 0.7, Ts * 0.15)), 5, Ts - 1))
        ids[b, :L] = rng.integers(2, 39, L)
        ids[b, L] = 1
    return ids


@pytest.fixture(scope='module')
def neighbour(hparams, weights):
    """A second handle whose only job is to keep MFMA GEMM waves on the chip."""
    eng2 = pkg().Engine(hparams)
    eng2.load_weights(weights)
    rng = np.random.default_rng(7)
    x = eng2.to_device(rng.standard_normal((9600, 256)).astype(np.float32))
    w = eng2.to_device(rng.standard_normal((256, 256)).astype(np.float32))
    c = eng2.empty((9600, 256))

    def launch(n=30):
        for _ in range(n):
            eng2._check(eng2.lib.tts_debug_gemm(eng2.handle, x.data_ptr(), w.data_ptr(), c.data_ptr(), 9600, 256, 256, 1, 150, 0))

    yield eng2, launch
    eng2.synchronize()
    for a in (x, w, c):
        a.free()
    eng2.close()


def _first(r):
    return r[0] if isinstance(r, tuple) else r


STAGES = ['encoder', 'decoder_launch', 'decoder_ws', 'postnet', 'gl_general_2048', 'gl_general_1024', 'gl_general_512', 'gl_stream',
          'gl_stream_800']


@pytest.mark.parametrize('stage', STAGES)
def test_stage_beside_gemm_launches_of_another_handle(engine, neighbour, stage):
    eng2, launch = neighbour
    rng = np.random.default_rng(11)
    B = 16
    keep = []

    def dev(a):
        keep.append(engine.to_device(a))
        return keep[-1]

    if stage == 'encoder':
        ids = dev(_ids(B, 60, 5))
        run = lambda: engine.encoder_forward(ids)
    elif stage in ('decoder_launch', 'decoder_ws'):
        mem = dev((rng.standard_normal((B, 60, 256)) * 0.5).astype(np.float32))
        pd = 0 if stage == 'decoder_launch' else 2

        def run():
            engine.set_option('persistent_decoder', pd)
            return _first(engine.decoder_forward(mem, 10))
    elif stage == 'postnet':
        mel = dev(rng.standard_normal((B, 100, 80)).astype(np.float32) * 0.1)
        run = lambda: _first(engine.postnet_forward(mel))
    else:
        n_fft, win, hop, T, n_iter = {'gl_general_2048': (2048, 1200, 300, 100, 2), 'gl_general_1024': (1024, 800, 200, 100, 2),
                                      'gl_general_512': (512, 400, 100, 100, 2), 'gl_stream': (2048, 1102, 275, 200, 3),
                                      'gl_stream_800': (2048, 800, 200, 200, 3)}[stage]
        mag = dev((rng.random((B, 1 + n_fft // 2, T), dtype=np.float32) ** 4) * 10)
        run = lambda: _first(engine.griffin_lim(mag, n_iter, win, hop, n_fft, seed=3, want_mse=False))
    try:
        quiet = run()
        engine.synchronize()
        ref = quiet.to_host().copy()
        assert np.isfinite(ref).all()
        bad = n = 0
        for _ in range(25):
            launch()
            outs = [run() for _ in range(2)]
            engine.synchronize()
            eng2.synchronize()
            for o in outs:
                n += 1
                bad += not np.array_equal(o.to_host(), ref)
        assert bad == 0, '%d of %d results differ from the quiet run' % (bad, n)
    finally:
        engine.set_option('persistent_decoder', 1)
        for a in keep:
            a.free()


@pytest.mark.parametrize('form', [(0, 1), (2, 1), (2, 0)], ids=['launch-per-layer', 'weight-stationary', 'streamed-weights'])
@pytest.mark.parametrize('variant', ['local-attention', 'cudnn-gru'])
def test_other_decoder_variants_beside_gemm_launches(hparams, weights, neighbour, variant, form):
    """The decoder kernels of the configurations the session's engine does not run -- LocalLuongAttention (attention.py:109-342;
    pd_attention_local has 139 packed-f32 instructions) and the CudnnCompatibleGRUCell form (layers.py:562-568) -- under the same
    neighbour."""
    import copy
    eng2, launch = neighbour
    hp = copy.deepcopy(hparams)
    if variant == 'local-attention':
        hp.attention.mechanism = 'LocalLuongAttention'
        hp.attention.luong_local_window_D = 6
        hp.attention.luong_force_gaussian = True
        w = weights
    else:
        hp.force_cudnn = True
        w = pkg('tacotron.weights').synthetic_weights(0, hp)
    eng = pkg().Engine(hp)
    eng.load_weights(w)
    pd, pd_ws = form
    eng.set_option('persistent_decoder', pd)
    eng.set_option('debug_hooks', 1)
    eng.set_option('pd_ws', pd_ws)
    try:
        mem = eng.to_device((np.random.default_rng(5).standard_normal((16, 60, 256)) * 0.5).astype(np.float32))
        run = lambda: _first(eng.decoder_forward(mem, 10))
        quiet = run()
        eng.synchronize()
        ref = quiet.to_host().copy()
        assert np.isfinite(ref).all()
        bad = n = 0
        for _ in range(20):
            launch()
            outs = [run() for _ in range(2)]
            eng.synchronize()
            eng2.synchronize()
            for o in outs:
                n += 1
                bad += not np.array_equal(o.to_host(), ref)
        assert bad == 0, '%d of %d results differ from the quiet run' % (bad, n)
    finally:
        eng.close()
