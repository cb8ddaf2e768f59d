"""GPU parity of the persistent decoder (decoder_persistent.hip: clusters of co-resident workgroups handing
activations to each other, reference tacotron/model.py:191-331).

It replaces launch boundaries by bounded waits between workgroups; a hand-off that loses a race shows up here as a
mismatch, a wait that never ends as TTS_ERR_HIP from tts_synchronize."""
import numpy as np
import pytest

from conftest import rel_l2
from oracle import tacotron_oracle as O

pytestmark = pytest.mark.gpu

WIN, HOP, NFFT = 1102, 275, 2048


@pytest.fixture(params=[(1, 32), (1, 16), (0, 0)], ids=['weight-stationary-32', 'weight-stationary-16', 'streamed-weights'])
def persistent(engine, request):
    """the persistent kernels: decoder_ws.hip (clusters of 16 workgroups, the weights resident in registers; the default) in
    both of its forms -- 32 utterances per cluster (round 5) and 16 (round 6: twice the compute units, a shorter phase) -- and
    decoder_persistent.hip (8 workgroups x 16 utterances, the weights streamed from L2 in every step)"""
    pd_ws, rows = request.param
    engine.set_option('persistent_decoder', 2)   # also outside the call pipeline
    engine.set_option('pd_ws', pd_ws)
    engine.set_option('debug_hooks', 1)
    engine.set_option('pd_rows', rows)
    yield engine
    engine.set_option('pd_rows', 0)
    engine.set_option('debug_hooks', 0)
    engine.set_option('pd_ws', 1)
    engine.set_option('persistent_decoder', 1)


@pytest.mark.parametrize('B,Ts,S', [(1, 5, 3), (3, 37, 10), (17, 150, 6), (33, 64, 4), (16, 1, 3), (2, 401, 3), (80, 33, 3)])
def test_persistent_decoder_vs_oracle(persistent, hparams, weights64, B, Ts, S):
    """Row counts that are no multiple of the 16- / 32-row cluster tile, one to five clusters (80 utterances = 40 / 48 workgroups),
    memories of 1 and of 401 positions (the score / context loops run 1 and 13 passes)."""
    rng = np.random.default_rng(200 + B)
    memory = (rng.standard_normal((B, Ts, 256)) * 1.5).astype(np.float32)
    ref_mel, ref_al = O.decoder(memory.astype(np.float64), weights64, hparams, n_steps=S)
    mel, al = persistent.decoder_forward(memory, S)
    persistent.synchronize()
    e_mel, e_al = rel_l2(mel.to_host(), ref_mel), float(np.abs(al.to_host() - ref_al).max())
    print('persistent decoder B={} Ts={} S={}: mel rel-L2 {:.3e}, align max-abs {:.3e}'.format(B, Ts, S, e_mel, e_al))
    assert e_mel < 1e-3
    assert e_al < 1e-4
    assert np.allclose(al.to_host().sum(-1), 1.0, atol=1e-5)


def test_persistent_decoder_200_steps_b64(persistent, hparams, weights64):
    """Config 3 through the persistent kernel: 2000 hand-offs per cluster, same bars as the launch path."""
    rng = np.random.default_rng(7)
    memory = (rng.standard_normal((64, 150, 256)) * 0.5).astype(np.float32)
    dev = persistent.to_device(memory)
    mel, al = persistent.decoder_forward(dev, 200)
    persistent.synchronize()
    mel, al = mel.to_host(), al.to_host()
    persistent.set_option('persistent_decoder', 0)
    mel0, al0 = persistent.decoder_forward(dev, 200)
    mel0, al0 = mel0.to_host(), al0.to_host()
    # equal to the launch-per-layer path to fp32 rounding (the K slices and the softmax are summed in another order)
    print('persistent vs launch path: mel rel-L2 {:.3e}, align max-abs {:.3e}'.format(rel_l2(mel, mel0), float(np.abs(al - al0).max())))
    assert rel_l2(mel, mel0) < 1e-5
    assert np.abs(al - al0).max() < 1e-5
    # ... and to the oracle on four of the rows, one from each cluster (the whole batch is test_config3's job)
    rows = [0, 21, 42, 63]
    ref_mel, ref_al = O.decoder(memory[rows].astype(np.float64), weights64, hparams)
    e = rel_l2(mel[rows], ref_mel)
    print('persistent decoder B=64 S=200 vs oracle rows {}: mel rel-L2 {:.3e}, align max-abs {:.3e}'.format(
        rows, e, float(np.abs(al[:, rows] - ref_al).max())))
    assert e < 1e-3
    assert np.abs(al[:, rows] - ref_al).max() < 1e-4


@pytest.mark.parametrize('B,Ts,S,reruns', [(20, 50, 12, 3), (64, 150, 200, 4)])
def test_persistent_decoder_reruns_are_bit_identical(persistent, B, Ts, S, reruns):
    """Every hand-off is a race the protocol has to win: a stale read anywhere in the up to 2000 hops per cluster would
    show up as a difference between runs of the same input (the arithmetic itself has a fixed order)."""
    rng = np.random.default_rng(11)
    memory = persistent.to_device(rng.standard_normal((B, Ts, 256)).astype(np.float32))
    mel, al = persistent.decoder_forward(memory, S)
    persistent.synchronize()
    a, b = mel.to_host(), al.to_host()
    for _ in range(reruns):
        persistent.decoder_forward(memory, S, mel=mel, alignments=al)
        persistent.synchronize()
        assert np.array_equal(a, mel.to_host()) and np.array_equal(b, al.to_host())


def test_pipelined_full_size_calls_repeat_bit_identically(engine):
    """The bench configuration (64 utterances, 200 decoder steps, 60 Griffin-Lim iterations) as back-to-back pipelined
    calls on the same input: the persistent decoder of call k + 1 runs beside the Griffin-Lim launches of call k, at
    full memory load.  Every call must reproduce the first one bit for bit (mel, alignments and waveform)."""
    rng = np.random.default_rng(3)
    ids = rng.integers(2, 39, (64, 150)).astype(np.int32)
    ids[:, -1] = 1
    d_ids = engine.to_device(ids)
    init = engine.to_device(rng.random((64, 1025, 1000), dtype=np.float32))
    kw = dict(n_steps=200, ref_db=6.02, max_db=99.89, power=1.3, n_iter=60, win_length=WIN, hop_length=HOP,
              init_phase=init, want_mel=True, want_alignments=True)
    engine.synthesize(d_ids, **kw)                      # first call of the shape: unpipelined, sizes the workspaces
    outs = [engine.synthesize(d_ids, **kw) for _ in range(4)]
    engine.synchronize()
    ref = {k: v.to_host() for k, v in outs[0].items() if v is not None}
    assert np.isfinite(ref['wav']).all() and np.abs(ref['wav']).max() > 0
    for o in outs[1:]:
        for k, v in ref.items():
            assert np.array_equal(v, o[k].to_host()), k


@pytest.mark.parametrize('B,Ts,S', [(1, 9, 4), (5, 37, 6), (16, 64, 5), (33, 150, 6), (64, 150, 12)])
def test_weight_stationary_decoder_same_bits_at_16_and_32_rows_per_cluster(engine, B, Ts, S):
    """decoder_ws.hip with 16 and with 32 utterances per cluster: the same K slices in the same order, four waves per
    attention row in both, independent MFMA rows -- mel spectrograms and alignments equal BIT FOR BIT, which is what lets the
    library choose the form by what else is running (csrc/api_stages.hip, decoder_impl)."""
    rng = np.random.default_rng(900 + B)
    memory = engine.to_device((rng.standard_normal((B, Ts, 256)) * 1.2).astype(np.float32))
    out = {}
    try:
        engine.set_option('persistent_decoder', 2)
        engine.set_option('debug_hooks', 1)
        for rows in (32, 16):
            engine.set_option('pd_rows', rows)
            mel, al = engine.decoder_forward(memory, S)
            engine.synchronize()
            out[rows] = (mel.to_host(), al.to_host())
    finally:
        engine.set_option('pd_rows', 0)
        engine.set_option('debug_hooks', 0)
        engine.set_option('persistent_decoder', 1)
    assert np.isfinite(out[32][0]).all() and np.abs(out[32][0]).max() > 0
    assert np.array_equal(out[16][0], out[32][0])
    assert np.array_equal(out[16][1], out[32][1])
