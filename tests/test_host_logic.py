"""CPU: host logic of the package -- C-ABI surface, text front-end, weights manifest, sharding."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, pkg

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_library_loads_and_exports_every_declared_symbol():
    """The .so must load without a GPU and export exactly what include/sstts_hip.h declares."""
    header = open(os.path.join(ROOT, 'include', 'sstts_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(tts_[a-z0-9_]+)\s*\(', header))
    sstts = pkg()
    lib = sstts.load_library()
    assert declared == set(sstts.exported_symbols())
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.tts_version().startswith(b'sstts_hip')
    nm = subprocess.run(['nm', '-D', '--defined-only', sstts.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (tts_[a-z0-9_]+)', nm))
    assert declared <= exported


def test_config_struct_of_another_size_is_refused():
    """tts_config_t starts with its own size (tts_default_config fills it): a zero-initialised struct, or one laid out by
    another version of the header, is refused by tts_create before anything behind the first field is read (no GPU needed)."""
    import ctypes
    H = pkg('_hip')
    lib = H.load_library()
    cfg = H.TtsConfig()
    lib.tts_default_config(ctypes.byref(cfg))
    assert cfg.struct_size == ctypes.sizeof(H.TtsConfig) and cfg.apply_post_processing == 1
    for bad in (0, cfg.struct_size - 4, cfg.struct_size + 4):
        c2 = H.TtsConfig()
        ctypes.memmove(ctypes.byref(c2), ctypes.byref(cfg), ctypes.sizeof(cfg))
        c2.struct_size = bad
        h = ctypes.c_void_p()
        assert lib.tts_create(ctypes.byref(c2), 0, ctypes.byref(h)) == H.TTS_ERR_INVALID
        assert b'struct_size' in lib.tts_last_error(None)


def test_missing_library_fails_loudly(tmp_path):
    H = pkg('_hip')
    with pytest.raises(OSError):
        H.load_library(str(tmp_path / 'nope.so'))


def test_no_product_module_imports_the_oracle():
    bad = []
    for dp, _dn, fn in os.walk(os.path.join(ROOT, 'single-speaker-tts_amd')):
        for f in fn:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M):
                    bad.append(f)
    assert not bad, bad


def test_text_frontend_matches_reference_run():
    g = json.load(open(os.path.join(GOLD, 'text_frontend.json')))
    LJ = pkg('datasets.lj_speech').LJSpeechDatasetHelper
    P = pkg('tacotron.params')
    assert P.dataset_params.vocabulary_dict == g['vocabulary']
    ds = LJ('/nonexistent', dict(P.dataset_params.vocabulary_dict), False)
    assert list(ds._abbreviations.items()) == [tuple(x) for x in g['abbreviations']]   # order matters
    ids, lens = ds.process_sentences(g['sentences'])
    assert lens == g['lengths']
    assert [np.frombuffer(b, dtype=np.int32).tolist() for b in ids] == g['ids']
    assert [ds.replace_abbreviations(s.lower()) for s in g['sentences']] == g['folded']
    assert all(seq[-1] == 1 for seq in g['ids'])                                      # trailing EOS
    assert ds.idx2sent(g['ids'][0][:-1]) == "tis a test!"
    with pytest.raises(KeyError) as e:
        ds.process_sentences([g['known_error']['sentence']])
    assert e.value.args[0] == g['known_error']['key']
    rp = g['reduction_padding']
    mel = np.arange(7 * 3, dtype=np.float32).reshape(7, 3)
    lin = np.arange(7 * 5, dtype=np.float32).reshape(7, 5)
    rm, rl = ds.apply_reduction_padding(mel, lin, 5)
    assert list(rm.shape) == rp['mel_shape'] and np.array_equal(rm, np.array(rp['mel'], np.float32))
    assert list(rl.shape) == rp['lin_shape'] and np.array_equal(rl, np.array(rp['lin'], np.float32))


def test_reference_stale_known_answers_documented():
    """reference datasets/tests/lj_speech.py:104-112,192-201 expect commas to be dropped, but the
    shipped abbreviation table (lj_speech.py:37-60) keeps them and ',' is in the vocabulary
    (params/dataset.py:19-31): the shipped CODE is the behaviour reproduced here."""
    LJ = pkg('datasets.lj_speech').LJSpeechDatasetHelper
    P = pkg('tacotron.params')
    ds = LJ('/nonexistent', dict(P.dataset_params.vocabulary_dict), False)
    s = 'Neild gives, on the authority of Mr. Burchell, the under sheriff of Middlesex,'
    out = ds.replace_abbreviations(s.lower())
    assert out == 'neild gives, on the authority of mister burchell, the under sheriff of middlesex,'
    assert out.replace(',', '') == 'neild gives on the authority of mister burchell the under sheriff of middlesex'


def test_pad_sentence_and_batch():
    S = pkg('sharding')
    b = S.pad_batch([[5, 6, 1], [7, 1]], pad_token=0)
    assert b.dtype == np.int32 and b.tolist() == [[5, 6, 1], [7, 1, 0]]
    assert S.pad_batch([[5, 1]], max_len=4).tolist() == [[5, 1, 0, 0]]


def test_shard_range_partitions():
    S = pkg('sharding')
    for n, w in [(512, 8), (64, 1), (10, 3), (3, 8)]:
        spans = [S.shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert S.shard_range(512, 8, 3) == (192, 256)


def test_params_mirror_reference_defaults():
    P = pkg('tacotron.params')
    m = P.ModelParams()
    assert (m.vocabulary_size, m.sampling_rate, m.n_fft, m.win_len, m.win_hop) == (39, 22050, 2048, 50.0, 12.5)
    assert (m.n_mels, m.reduction, m.magnitude_power, m.reconstruction_iterations) == (80, 5, 1.3, 50)
    assert m.decoder.maximum_iterations // m.reduction == 200
    assert (m.encoder.n_banks, m.post.n_banks, m.post.projections[0][0], m.post.projections[1][0]) == (16, 8, 256, 80)
    assert P.inference_params.n_synthesis_threads == 6
    assert (P.LJSpeechConstants.mel_mag_ref_db, P.LJSpeechConstants.mel_mag_max_db) == (6.02, 99.89)


def test_fft_decomposition_emulation():
    """numpy emulation of the wave-level FFT the HIP kernel implements (64 lanes x 16 points:
    radix-16 -> transpose -> radix-4 -> transpose -> radix-16) and of the real<->complex
    split/merge formulas, against numpy.fft."""
    M, N = 1024, 2048
    Wn = lambda n, k: np.exp(-2j * np.pi * k / n)
    rng = np.random.default_rng(0)
    z = rng.standard_normal(M) + 1j * rng.standard_normal(M)
    v = z.reshape(16, 64).T                                       # v[l][j] = z[l + 64 j]
    j = np.arange(16)
    Y = v @ Wn(16, np.outer(j, j)) * Wn(1024, np.outer(np.arange(64), j))   # radix-16 + twiddle
    E1 = Y.T                                                      # [k2][n1]
    P = np.zeros((64, 4, 4), complex)
    for lam in range(64):
        a, kq = lam & 15, lam >> 4
        for i in range(4):
            x = E1[kq + 4 * i, a + 16 * np.arange(4)]
            P[lam, i] = (x @ Wn(4, np.outer(np.arange(4), np.arange(4)))) * Wn(64, a * np.arange(4))
    E2 = np.zeros((64, 16), complex)
    for lam in range(64):
        a, kq = lam & 15, lam >> 4
        for i in range(4):
            for d in range(4):
                E2[16 * d + kq + 4 * i, a] = P[lam, i, d]
    out = E2 @ Wn(16, np.outer(j, j))                             # out[mu][c] = X[mu + 64 c]
    assert np.abs(out.T.reshape(-1) - np.fft.fft(z)).max() < 1e-10
    x = rng.standard_normal(N)
    Zf = np.fft.fft(x[0::2] + 1j * x[1::2])
    k = np.arange(M)
    Zr = np.conj(Zf[(M - k) % M])
    X = 0.5 * (Zf + Zr) - 0.5j * np.exp(-2j * np.pi * k / N) * (Zf - Zr)
    ref = np.fft.rfft(x)
    assert np.abs(X - ref[:M]).max() < 1e-10 and abs(Zf[0].real - Zf[0].imag - ref[M]) < 1e-10
    Xs = ref.copy()
    Xs[0] += 0.3j
    Xs[M] -= 0.7j                                                 # imaginary DC / Nyquist must drop out
    Xk = Xs[:M].copy()
    Xk[0] = Xs[0].real
    Xmk = np.conj(Xs[M - k])
    Xmk[0] = Xs[M].real
    Zi = 0.5 * (Xk + Xmk) + 0.5j * np.exp(2j * np.pi * k / N) * (Xk - Xmk)
    z2 = np.conj(np.fft.fft(np.conj(Zi))) / M
    xx = np.empty(N)
    xx[0::2], xx[1::2] = z2.real, z2.imag
    assert np.abs(xx - np.fft.irfft(Xs, n=N)).max() < 1e-12


def test_permlane_swap_transpose_emulation():
    """The first FFT exchange is done in registers: swapping register-index bit 0 with lane bit 4
    (v_permlane16_swap) and register-index bit 1 with lane bit 5 (v_permlane32_swap) must equal the LDS
    transpose it replaced: new v[4 i + b] at lane (a, kq) = old v[4 i + kq] at lane (a, b)."""
    lanes = np.arange(64)
    old = np.arange(64 * 16).reshape(64, 16)                       # old[lane][reg] unique tags

    def swap32(A, B):        # lanes 32-63 of A swap with lanes 0-31 of B
        A2, B2 = A.copy(), B.copy()
        A2[32:], B2[:32] = B[:32], A[32:]
        return A2, B2

    def swap16(A, B):        # odd rows (16 lanes) of A swap with even rows of B
        A2, B2 = A.copy(), B.copy()
        for row in (0, 2):
            A2[16 * (row + 1):16 * (row + 2)] = B[16 * row:16 * (row + 1)]
            B2[16 * row:16 * (row + 1)] = A[16 * (row + 1):16 * (row + 2)]
        return A2, B2

    v = [old[:, r].copy() for r in range(16)]
    for i in range(4):
        v[4 * i + 0], v[4 * i + 1] = swap16(v[4 * i + 0], v[4 * i + 1])
        v[4 * i + 2], v[4 * i + 3] = swap16(v[4 * i + 2], v[4 * i + 3])
        v[4 * i + 0], v[4 * i + 2] = swap32(v[4 * i + 0], v[4 * i + 2])
        v[4 * i + 1], v[4 * i + 3] = swap32(v[4 * i + 1], v[4 * i + 3])
    a, kq = lanes & 15, lanes >> 4
    for i in range(4):
        for b in range(4):
            assert np.array_equal(v[4 * i + b], old[a + 16 * b, 4 * i + kq])


def _gl_plan(T, B, win, hop, workers):
    lib = pkg('_hip').load_library()
    import ctypes
    cap = 16384
    buf = (ctypes.c_int * (4 * cap))()
    ring = ctypes.c_int(0)
    n = lib.tts_debug_gl_plan(T, B, win, hop, workers, buf, cap, ctypes.byref(ring))
    assert 1 <= n <= cap, n
    return [(buf[4 * k], buf[4 * k + 1], buf[4 * k + 2], buf[4 * k + 3]) for k in range(n)], ring.value


@pytest.mark.parametrize('T,B,win,hop,workers', [
    (1000, 64, 1102, 275, 256), (1000, 64, 1102, 275, 224), (1000, 1, 1102, 275, 256), (5, 2, 1102, 275, 256),
    (37, 3, 1102, 275, 248), (1000, 512, 1102, 275, 256), (400, 16, 2048, 512, 256), (333, 7, 400, 160, 256),
    (1000, 64, 551, 275, 224), (1000, 64, 800, 200, 224), (1, 1, 1102, 275, 256), (813, 5, 1102, 275, 224),
    (1000, 63, 1102, 275, 224), (200, 300, 1102, 275, 256),
])
def test_griffin_lim_item_plan_covers_every_frame_once(T, B, win, hop, workers):
    """Host planner of the streaming Griffin-Lim kernel (gl_plan_items): the runs tile [0, T) of every utterance exactly, a
    run's slot is its ordinal inside the utterance, the utterance's last run carries the number of slots other utterances
    have beyond it, the table starts with one run per worker (the runs that follow a worker's first come after them), no
    worker's frames + per-run cost exceed the average by more than a run's rounding, and the LDS ring holds enough frames."""
    items, ring = _gl_plan(T, B, win, hop, workers)
    assert ring >= 8
    runs = {}
    for b, t0, n, w in items:
        assert 0 <= b < B and n >= 1
        runs.setdefault(b, []).append((t0, n, w))
    assert sorted(runs) == list(range(B))
    spu = max(len(r) for r in runs.values())
    for b, r in runs.items():
        r.sort()
        t = 0
        for k, (t0, n, w) in enumerate(r):
            assert t0 == t
            t += n
            assert (w & 0xffff) == k
            assert (w >> 16) == (spu - len(r) if k == len(r) - 1 else 0)
        assert t == T
    # same inputs -> same cut
    assert _gl_plan(T, B, win, hop, workers)[0] == items
    if len(items) > workers:
        # the frames a worker gets: a first run and the runs dealt after the first `workers` entries, in table order to the
        # worker that comes free first -- the list schedule the kernel's item counter produces
        import heapq
        heap = [(n, k) for k, (_, _, n, _) in enumerate(items[:workers])]
        heapq.heapify(heap)
        for _, _, n, _ in items[workers:]:
            load, k = heapq.heappop(heap)
            heapq.heappush(heap, (load + n + 11, k))
        worst = max(l for l, _ in heap)
        assert worst <= B * T / workers * 1.06 + 24, (worst, B * T / workers)
    if (T, B, win, hop, workers) == (1000, 64, 1102, 275, 224):
        # 2 utterances on 7 workgroups: 288 + 286 + 286 + (140 | 140) + 286 + 286 + 288 -- the workgroup in the middle takes the
        # tail of one utterance and the head of the next (until round 6: 3 x 296 + 112 for every utterance, 32 of the 224
        # workgroups idle for a fifth of every launch)
        assert sorted({n for _, _, n, _ in items}) == [140, 286, 288]
        assert [n for b, t0, n, w in items[224:]] == [140] * 32


def _ring_frames(win, hop, n_stage=1):
    """gl_stream_ring_frames (griffin_lim.hip) restated."""
    wpad = (2048 - win) >> 1
    c_lo = wpad >> 7
    S = 128 * (((wpad + win - 1) >> 7) - c_lo + 1)
    acc = S - hop
    if acc < 0:
        return 0
    halo = -(-win // hop) - 1
    lag = halo if (halo + 1) * hop > 2 * (1024 - wpad) else halo + 1
    budget = (160 * 1024 - 8 * 1088 * 8 - 64) // 4 // n_stage - acc - 132
    need = max(9 + lag + -(-S // hop) + 1, -(-(S + win + 2 * hop) // hop), 8)
    R = min(budget // hop, max(64, need))
    return R if R >= need else 0


@pytest.mark.parametrize('win,hop,T,n_stage', [
    (1102, 275, 200, 1), (1102, 275, 75, 3), (1102, 275, 40, 2), (1024, 256, 40, 1), (1024, 256, 130, 3), (2048, 512, 100, 2),
    (400, 100, 150, 1), (64, 8, 300, 1), (2048, 1024, 30, 1), (1000, 250, 90, 2), (1101, 275, 50, 1), (2, 1, 700, 1),
    (1500, 1400, 20, 1), (2047, 256, 60, 1), (1200, 300, 70, 3), (800, 200, 130, 3), (800, 200, 45, 1),
])
def test_stream_ring_emulation(win, hop, T, n_stage):
    """The index arithmetic of gl_stream_kernel, emulated in numpy with the frames' windowed signals as random vectors:
    span slots, accumulate / store split, guard and fold at the lap end, the `lag` between a frame's overlap-add and its
    forward transform, linear against index-mapped reads (reflect padding; wrapped spans before and after the fold) --
    every frame's transform input must equal the directly overlap-added, reflect-padded signal, and the samples the final
    iSTFT writes out must tile [0, hop (T - 1)) exactly once, for whole utterances and for utterances cut into runs.
    (Only the last stage of a multi-iteration launch is emulated here -- the stages are identical machines with shifted
    index ranges; the ring is sized for n_stage of them.)"""
    NFFT, MH = 2048, 1024
    R = _ring_frames(win, hop, n_stage)
    if R == 0:
        pytest.skip('this window / hop pair is refused (TTS_ERR_UNSUPPORTED)')
    ncol = -(-win // hop)
    halo = ncol - 1
    wpad = (NFFT - win) >> 1
    c_lo = wpad >> 7
    n_sl = ((wpad + win - 1) >> 7) - c_lo + 1
    S, fs = 128 * n_sl, 128 * c_lo
    acc, ring_len, w2 = S - hop, hop * R, MH - wpad
    lag = halo if (halo + 1) * hop > 2 * w2 else halo + 1
    Ltot = hop * (T - 1)
    rng = np.random.default_rng(win + hop)
    fr = np.zeros((T, NFFT))
    fr[:, wpad:wpad + win] = rng.standard_normal((T, win))
    full = np.zeros(hop * (T - 1) + NFFT)
    for t in range(T):
        full[t * hop:t * hop + NFFT] += fr[t]
    ytrim = full[MH:MH + Ltot]

    def frame_input(tm):
        y = tm * hop + np.arange(wpad, wpad + win) - MH
        y = np.where(y < 0, -y, y)
        y = np.where(y >= Ltot, 2 * (Ltot - 1) - y, y)
        out = np.zeros(NFFT)
        out[wpad:wpad + win] = ytrim[y]
        return out

    def run(t0, Lrun):
        ring = np.full(ring_len + acc + 128, np.nan)
        ring[ring_len:] = 0
        n_idx = Lrun + halo + lag
        y_base = (t0 - halo) * hop - MH + fs
        emitted = {}
        for i in range(n_idx):
            t, s = t0 - halo + i, i % R
            v = fr[t] if 0 <= t < T else np.zeros(NFFT)
            wr = hop * s
            rd = wr + (ring_len if s == 0 else 0)
            old = np.zeros(S)
            old[:acc] = ring[rd:rd + acc]
            new = old + v[fs:fs + S]
            ring[wr:wr + S] = new
            q_fin = wpad - fs
            if (t >= t0 or t0 == 0) and (t < t0 + Lrun or t0 + Lrun == T):
                for q in range(q_fin, q_fin + hop):
                    y = t * hop + fs - MH + q
                    if 0 <= y < Ltot:
                        assert y not in emitted
                        emitted[y] = new[q]
            if halo + lag <= i < halo + lag + Lrun:
                m, tm, sm = i - lag, t - lag, (s - lag) % R
                ylo = tm * hop + wpad - MH
                edge = ylo < 0 or ylo + win > Ltot
                got = np.zeros(NFFT)
                if not edge and (hop * sm + S <= ring_len or R - sm > lag):
                    got[fs:fs + S] = ring[hop * sm:hop * sm + S]
                    got[:wpad] = 0
                    got[wpad + win:] = 0
                else:
                    lap0 = m // R
                    for f in range(wpad, wpad + win):
                        y = ylo + (f - wpad)
                        y = -y if y < 0 else y
                        y = 2 * (Ltot - 1) - y if y >= Ltot else y
                        off, lap = (y - y_base) - lap0 * ring_len, lap0
                        if off >= ring_len:
                            off, lap = off - ring_len, lap + 1
                        elif off < 0:
                            off, lap = off + ring_len, lap - 1
                        assert 0 <= off < ring_len
                        got[f] = ring[ring_len + off if (off < acc and lap * R > i) else off]
                assert np.allclose(got, frame_input(tm), atol=1e-12), (tm, i, edge)
        assert all(abs(v - ytrim[y]) < 1e-12 for y, v in emitted.items())
        return set(emitted)

    assert run(0, T) == set(range(Ltot))
    L1 = max(8, T // 3 // 8 * 8)
    seen = set()
    for t0 in range(0, T, L1):
        part = run(t0, min(L1, T - t0))
        assert not (seen & part)
        seen |= part
    assert seen == set(range(Ltot))


def test_phasor_code_emulation():
    """The Griffin-Lim state between iterations is a 32-bit code per bin (griffin_lim.hip, gl_pack_phasor /
    gl_unpack_phasor): the point where the phasor's ray meets the diamond |Re| + |Im| = 1, stored as its imaginary part
    p = Im / (|Re| + |Im|), a float whose lowest mantissa bit carries the sign of Re; decoded with |Re| = 1 - |p| and one
    reciprocal square root.  numpy emulation of exactly those operations (with a random relative error of one ulp on the
    reciprocal and the reciprocal square root, as v_rcp_f32 / v_rsq_f32 have): the decoded phasor is within 4e-7 of the
    exact unit phasor for random angles at any scale, on the axes, the diagonals and next to them, and the zero bin
    decodes to (1, 0)."""
    rng = np.random.default_rng(5)
    ang = np.concatenate([rng.uniform(-np.pi, np.pi, 200000), np.arange(-8, 9) * (np.pi / 4),
                          np.arange(-8, 9) * (np.pi / 4) + 1e-6, np.arange(-8, 9) * (np.pi / 4) - 3e-4])
    scale = np.exp(rng.uniform(-20, 20, ang.size))                 # the FFT output is not normalised
    x = (np.cos(ang) * scale).astype(np.float32)
    y = (np.sin(ang) * scale).astype(np.float32)
    one = np.float32(1.0)

    def hw(v):   # a transcendental unit's result: one ulp
        return (v * (1 + rng.uniform(-1, 1, v.shape) * 2.0 ** -24)).astype(np.float32)

    def pack(x, y):
        den = np.maximum((np.abs(x) + np.abs(y)).astype(np.float32), np.float32(1e-30))
        p = (y * hw((one / den).astype(np.float32))).astype(np.float32)
        return (p.view(np.uint32) & ~np.uint32(1)) | (x.view(np.uint32) >> np.uint32(31))

    def unpack(code):
        p = code.view(np.float32)   # the flag bit is not stripped: one ulp of p, what stripping costs too
        q = (one - np.abs(p)).astype(np.float32)
        xs = (q.view(np.uint32) | (code << np.uint32(31))).view(np.float32)
        s2 = (p * p + (q * q).astype(np.float32)).astype(np.float32)
        n = hw((one / np.sqrt(s2)).astype(np.float32))
        return (xs * n).astype(np.float32), (p * n).astype(np.float32)

    code = pack(x, y)
    ux, uy = unpack(code)
    err = np.maximum(np.abs(ux - np.cos(ang)), np.abs(uy - np.sin(ang)))
    assert err.max() < 4e-7, err.max()
    assert np.all(np.abs((code & ~np.uint32(1)).view(np.float32)) <= 1.0)          # the stored value lies on the diamond
    assert np.abs(ux * ux + uy * uy - 1.0).max() < 6e-7
    # zero bins (angle(0) = 0 -> 1 + 0j, reference audio/synthesis.py:109) and the codes the kernel writes directly
    z = np.zeros(1, np.float32)
    def near(v, want):   # a flag bit left in a zero ratio is a denormal (1e-45), not a phase
        return abs(float(v[0][0]) - want[0]) < 2e-7 and abs(float(v[1][0]) - want[1]) < 1e-40

    assert near(unpack(pack(z, z)), [1.0, 0.0])
    assert near(unpack(pack(-z, z)), [-1.0, 0.0])                                  # np.angle(-0.0 + 0j) = pi
    for c, want in ((np.uint32(0), [1.0, 0.0]), (np.uint32(1), [-1.0, 0.0])):      # Nyquist / DC phasors
        assert near(unpack(np.array([c], np.uint32)), want)


def test_bench_cpu_baseline_runs_in_a_child_without_the_gpu():
    """bench.py times the numpy oracle in a child interpreter (`--cpu-baseline-only`): its fork()ed worker pools must not be
    made by the process that holds the HIP runtime.  The child, on a one-utterance sample: a JSON object with the fields the
    bench line carries, and no GPU call on the way (this suite runs where there is none)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--cpu-baseline-only', '--cpu-baseline-utts', '1',
                        '--cpu-baseline-repeats', '1'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')][-1])
    assert out['kind'] == 'port' and out['unit'] == 'mel-frames/s' and out['value'] > 0 and out['cores'] >= 1
    assert out['gl_workers_reference'] == 1 and out['repeats'] == 1
    # and the parent's wrapper reports a failing child instead of waiting for it
    import bench
    src = open(os.path.join(root, 'bench.py')).read()
    assert "cpu_baseline_in_child()" in src and "out['cpu_baseline'] = cpu_baseline(" not in src
    assert callable(bench.cpu_baseline_in_child)
