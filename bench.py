#!/usr/bin/env python3
"""Benchmark of the MI355X-native Tacotron inference hot path.

    python bench.py --gpus N --steps K --warmup W

One "step" = one end-to-end pass of the hot path over one 64-utterance LJ-Speech-shaped
synthetic batch per GPU: ids (64,150) -> CBHG encoder -> 200-step attention decoder (r=5,
1000 mel frames) -> post-net CBHG + 1025-bin linear spectrogram -> de-normalise, ^1.3 ->
60-iteration Griffin-Lim -> peak-normalised waveform (64, 274725).  This is the
configuration BASELINE.json's metric ("mel-frames/sec + Griffin-Lim real-time-factor, 64-utt
LJ-Speech batch") is quoted on.  Inputs (ids, initial phases) are resident in HBM before the
timed region.  For N > 1 every rank runs its own 64-utterance shard (weak scaling, no data-path
collective); weights are generated on rank 0 and broadcast once over RCCL before timing.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 64
TS = 150
N_STEPS = 200
N_ITER = 60
WIN, HOP, N_FFT = 1102, 275, 2048
REF_DB, MAX_DB, POWER = 6.02, 99.89, 1.3
SR = 22050
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-input MFMA (v_mfma_f32_32x32x2_f32), same guide


def synthetic_ids(B, Ts, seed):
    """lengths ~ clip(round(N(100,30)),20,149), ids U[2,38], EOS=1, pad=0."""
    rng = np.random.default_rng(seed)
    ids = np.zeros((B, Ts), np.int32)
    for b in range(B):
        L = int(np.clip(round(rng.normal(100, 30)), 20, Ts - 1))
        ids[b, :L] = rng.integers(2, 39, L)
        ids[b, L] = 1
    return ids


def cpu_baseline(weights, hp, n_utts=6):
    """Times the numpy oracle (a restated CPU path -- NOT TensorFlow, which cannot run here)
    on a bounded sample: n_utts utterances, network in one process (float32), Griffin-Lim in
    an n_utts-process pool the way the reference fans it out (tacotron/inference.py:185-188,
    params/inference.py:34: 6 workers)."""
    from multiprocessing import get_context
    from oracle import tacotron_oracle as O
    w32 = {k: np.asarray(v, np.float32) for k, v in weights.items()}
    ids = synthetic_ids(n_utts, TS, 4321)
    t0 = time.perf_counter()
    out = O.tacotron_predict(ids, w32, hp, n_steps=N_STEPS)
    t_net = time.perf_counter() - t0
    lin = out['linear'].astype(np.float32)
    jobs = [(lin[b], b) for b in range(n_utts)]
    t0 = time.perf_counter()
    with get_context('fork').Pool(n_utts) as pool:
        wavs = pool.map(_cpu_gl_job, jobs)
    t_gl = time.perf_counter() - t0
    assert all(w.shape == (HOP * (N_STEPS * hp.reduction - 1),) for w in wavs)
    frames = n_utts * N_STEPS * hp.reduction
    return dict(value=frames / (t_net + t_gl), unit='mel-frames/s', cores=n_utts, kind='port',
                sample='{} utterances end-to-end (Ts={}, {} decoder steps, {} GL iterations): numpy oracle, '
                       'network {:.2f} s in 1 process + Griffin-Lim {:.2f} s in a {}-process pool'.format(
                           n_utts, TS, N_STEPS, N_ITER, t_net, t_gl, n_utts),
                griffin_lim_rtf=t_gl / (n_utts * HOP * (N_STEPS * hp.reduction - 1) / SR))


def _cpu_gl_job(args):
    from oracle import audio_oracle as A
    lin, seed = args
    mag = A.linear_to_magnitude(lin, REF_DB, MAX_DB, POWER)
    init = np.random.default_rng(seed).random(mag.shape)
    return A.peak_normalize(A.spectrogram_to_wav(mag, WIN, HOP, N_FFT, N_ITER, init_phase=init))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes, one rank per GPU, wired up
    the way `python -m torch.distributed.run --nproc-per-node N` would (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  The parent never touches the GPU (no HIP call, no torch.cuda): the children are
    plain subprocesses, nothing is exec'ed over a process that holds a device."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r else None))
    rc = 0
    for r, pr in enumerate(procs):
        pr.wait()
        rc = rc or pr.returncode
    return rc


def dist_selftest(rank, local_rank, world, dist):
    """--dist-selftest: everything of the N > 1 path except the GPU work (process group, the one weight broadcast,
    shard ranges, the max-over-ranks reduction), so that the launcher can be rehearsed on a box without GPUs."""
    shard = importlib.import_module('single-speaker-tts_amd.sharding')
    Wm = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
    n = Wm.n_parameters()
    blob = Wm.pack_blob(Wm.synthetic_weights(0)) if rank == 0 else np.zeros(n, np.float32)
    got = shard.broadcast_blob(blob, src=0, device='cpu')
    import torch
    t = torch.tensor([float(np.abs(got).sum()), float(rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    lo, hi = shard.shard_range(world * B_PER_GPU, world, rank)
    ok = abs(float(np.abs(got).sum()) - float(t[0])) < 1e-6 and hi - lo == B_PER_GPU
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        print(json.dumps({'selftest': 'dist', 'n_gpus': world, 'world_size_seen': dist.get_world_size(),
                          'backend': dist.get_backend(), 'ok': all(flags)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # 200 steps = 4.5 s: long enough that the one-off pipeline fill (the first timed call's encoder + decoder have nothing
    # to overlap with: ~16 ms once) is below 0.1 ms per step; 40 steps report 0.4 ms per step more than the steady state
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--pipeline', type=int, default=None, help='override the library default (stream pipelining)')
    ap.add_argument('--reserve-cus', type=int, default=None)
    ap.add_argument('--hold-lds-kb', type=int, default=None)
    ap.add_argument('--persistent-decoder', type=int, default=None, help='override the library default (0 never, 1 pipelined, 2 always)')
    ap.add_argument('--gl-fused', type=int, default=None, help='override the library default (all Griffin-Lim iterations in one launch)')
    ap.add_argument('--dist-selftest', action='store_true', help='rehearse the N > 1 launch path without GPU work')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world > 1:
        raise SystemExit('WORLD_SIZE ({}) != --gpus ({})'.format(world, args.gpus))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        # one process per GPU over RCCL ("nccl" IS RCCL on ROCm).  SSTTS_DIST_BACKEND=gloo lets the
        # N > 1 path be rehearsed on a box with fewer GPUs than ranks (ranks then share devices).
        backend = os.environ.get('SSTTS_DIST_BACKEND', 'nccl')
        n_dev = max(1, torch.cuda.device_count())
        if backend != 'nccl':
            local_rank = local_rank % n_dev
        if not args.dist_selftest:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
        if rank == 0:
            print('bench: process group up, backend {} world size {}'.format(dist.get_backend(), dist.get_world_size()),
                  file=sys.stderr, flush=True)
    if args.dist_selftest:
        if dist is None:
            raise SystemExit('--dist-selftest needs --gpus N > 1')
        return dist_selftest(rank, local_rank, world, dist)

    sstts = importlib.import_module('single-speaker-tts_amd')
    P = importlib.import_module('single-speaker-tts_amd.tacotron.params')
    Wm = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
    shard = importlib.import_module('single-speaker-tts_amd.sharding')
    hp = P.ModelParams()

    # ---- weights: rank 0 generates, one RCCL broadcast of the flat blob, then no communication
    n_floats = Wm.n_parameters(hp)
    weights = None
    if rank == 0:
        weights = Wm.synthetic_weights(0, hp)
        blob = Wm.pack_blob(weights, hp)
    else:
        blob = np.empty(n_floats, np.float32)
    if world > 1:
        bdev = 'cuda:{}'.format(local_rank) if dist.get_backend() == 'nccl' else 'cpu'
        blob = shard.broadcast_blob(blob, src=0, device=bdev)
    eng = sstts.Engine(hp, device_id=local_rank)
    eng.load_weights_blob(blob)
    if args.pipeline is not None:
        eng.set_option('pipeline', args.pipeline)
    if args.reserve_cus is not None:
        eng.set_option('reserve_cus', args.reserve_cus)
    if args.hold_lds_kb is not None:
        eng.set_option('hold_lds_kb', args.hold_lds_kb)
    if args.persistent_decoder is not None:
        eng.set_option('persistent_decoder', args.persistent_decoder)
    if args.gl_fused is not None:
        eng.set_option('gl_fused', args.gl_fused)

    # ---- this rank's shard of the synthetic batch, resident in HBM
    lo, hi = shard.shard_range(world * B_PER_GPU, world, rank)
    ids_all = synthetic_ids(world * B_PER_GPU, TS, 1234)
    ids = eng.to_device(ids_all[lo:hi])
    B = hi - lo
    T = N_STEPS * hp.reduction
    F = 1 + N_FFT // 2
    init = eng.to_device(np.random.default_rng(42 + rank).random((B, F, T), dtype=np.float32))
    wav = eng.empty((B, HOP * (T - 1)))

    def step():
        eng.synthesize(ids, N_STEPS, REF_DB, MAX_DB, POWER, N_ITER, WIN, HOP, init_phase=init,
                       peak_normalize=True, wav=wav)

    def barrier():
        eng.synchronize()
        if dist is not None:
            dist.barrier()
        eng.synchronize()

    for _ in range(args.warmup):
        step()
    eng.set_option('profile', 1)
    eng.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device='cuda:{}'.format(local_rank) if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stage_ms = {}
    launches = {}
    for st in ('encoder', 'decoder', 'postnet', 'gl_iter', 'gl_final'):
        ms, n = eng.profile_get(st)
        stage_ms[st] = ms / args.steps
        launches[st] = n // max(1, args.steps)
    w = wav.to_host()
    assert np.isfinite(w).all()
    # the dominant kernel alone (no overlapping stream), for reference next to the in-pipeline figure
    eng.synchronize()
    eng.profile_reset()
    # (non-zero magnitudes: an all-zero spectrogram would send every frame down the kernel's exact
    #  zero-bin path, which the synthesised spectrograms of the timed steps never take)
    mag_alone = eng.to_device((np.random.default_rng(1).random((B, F, T), dtype=np.float32) ** 4) * 10 + 1e-3)
    eng.griffin_lim(mag_alone, 8, WIN, HOP, N_FFT, init_phase=init, want_mse=False)
    ms_alone, n_alone = eng.profile_get('gl_iter')
    gl_alone_ms = ms_alone / max(1, n_alone)
    mag_alone.free()
    # second roofline: the MFMA GEMM kernel on the largest post-net layer (first projection: conv1d k=3,
    # 1024 -> 256 channels, max-pool fused into the loader, M = B*T rows), HIP events around tts_debug_gemm
    M, N, CIN, KT = B * T, hp.post.projections[0][0], hp.post.n_banks * hp.post.n_filters, 3
    rng = np.random.default_rng(3)
    ga = eng.to_device(rng.standard_normal((M + 8, CIN), dtype=np.float32))
    gw = eng.to_device(rng.standard_normal((N, KT * CIN), dtype=np.float32) * 0.05)
    gc = eng.empty((M, N))
    eng._check(eng.lib.tts_debug_gemm(eng.handle, ga.data_ptr(), gw.data_ptr(), gc.data_ptr(), M, N, CIN, KT, T, 1))
    eng.synchronize()
    eng.profile_reset()
    for _ in range(10):
        eng._check(eng.lib.tts_debug_gemm(eng.handle, ga.data_ptr(), gw.data_ptr(), gc.data_ptr(), M, N, CIN, KT, T, 1))
    gemm_ms, gemm_n = eng.profile_get('debug_gemm')
    gemm_ms /= max(1, gemm_n)
    gemm_flop = 2.0 * M * N * KT * CIN
    for d in (ga, gw, gc):
        d.free()

    if rank == 0:
        frames_total = world * B_PER_GPU * T
        ms_per_step = 1e3 * elapsed / args.steps
        audio_s = B_PER_GPU * HOP * (T - 1) / SR
        # dominant kernel: gl_iter_kernel<0>, one launch = one Griffin-Lim iteration over the batch.
        # Algorithmic bytes per launch: 20 B per bin = |S| 4 + phase in 8 + phase out 8.
        gl_launch_ms = stage_ms['gl_iter'] / max(1, launches['gl_iter'])
        alg_bytes = 20.0 * F * T * B_PER_GPU
        achieved = alg_bytes / (gl_launch_ms * 1e-3) / 1e9 if gl_launch_ms > 0 else 0.0
        # HBM bytes per launch from the PMC passes of the same build (FETCH_SIZE / WRITE_SIZE cannot be read
        # inside this process): newest profiles/*gl_iter_hbm_bytes_per_launch.json
        traffic = None
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*gl_iter_hbm_bytes_per_launch.json')))
        if pmcs:
            with open(pmcs[-1]) as f:
                traffic = json.load(f).get('hbm_bytes_per_launch')
        out = {
            'metric': 'mel-frames/sec (end-to-end text->waveform incl. 60-iter Griffin-Lim, 64-utt LJ-Speech-shaped batch per GPU)',
            'value': frames_total * args.steps / elapsed,
            'unit': 'mel-frames/s',
            'n_gpus': world,
            'world_size_seen': dist.get_world_size() if dist is not None else 1,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'end-to-end Tacotron inference, B=64/GPU, T_sent=150, 200 decoder steps (r=5, 1000 frames), '
                                   '1025-bin linear, Griffin-Lim 60 iterations, n_fft 2048 / win 1102 / hop 275',
                       'global_batch': world * B_PER_GPU, 'parallelism': 'utterance shards, dp{}'.format(world)},
            'griffin_lim_rtf': (stage_ms['gl_iter'] + stage_ms['gl_final']) * 1e-3 / audio_s,
            'end_to_end_rtf': ms_per_step * 1e-3 / audio_s,
            'mel_frames_per_sec_encoder_decoder': B_PER_GPU * T / ((stage_ms['encoder'] + stage_ms['decoder']) * 1e-3),
            'stage_ms': stage_ms,
            'roofline': {'kernel': 'gl_iter_kernel<0> (one Griffin-Lim iteration, iSTFT+STFT fused)', 'bound': 'hbm',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'launch_ms': gl_launch_ms, 'launch_ms_alone': gl_alone_ms,
                         'algorithmic_bytes_per_launch': alg_bytes,
                         'note': 'achieved = algorithmic bytes (20 B per bin and iteration, SURVEY.md 8(d)) / launch time; the kernel '
                                 'itself moves 12 B per bin (32-bit phasor code in and out, 4 B magnitude in), which is what traffic shows'},
            'roofline_mfma': {'kernel': 'gemm_f32_kernel (post-net projection 1: conv1d k=3, 1024 -> 256, max-pool in the loader, M = {})'.format(M),
                              'bound': 'mfma', 'achieved': gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0,
                              'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                              'frac': (gemm_flop / (gemm_ms * 1e-3) / 1e12) / MFMA_F32_PEAK_TFLOPS if gemm_ms > 0 else 0.0,
                              'traffic': None, 'launch_ms': gemm_ms, 'flop_per_launch': gemm_flop},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(weights, hp)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
