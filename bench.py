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
GL_PER_LAUNCH = 3   # the library's default (tts_set_option "gl_pair")
WIN, HOP, N_FFT = 1102, 275, 2048
REF_DB, MAX_DB, POWER = 6.02, 99.89, 1.3
SR = 22050
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-input MFMA (v_mfma_f32_32x32x2_f32), same guide
MFMA_BF16_PEAK_TFLOPS = 16 * 157.3   # dense bf16 MFMA = 16 x the f32 rate (same guide, matrix cores table): ~2.5 PFLOP/s
GEMM_PRODUCTS = 6              # bf16 MFMAs per f32-equivalent product block (csrc/gemm_f32.hip: hi/mid/lo split, six products)
DEC_MFLOP_PER_UTT_STEP = 3.02  # decoder loop, MFLOP per step and utterance
DEC_PHASES_PER_STEP = 10       # hand-off phases of the persistent decoder per step (GRUCell form)
DEC_KERNELS = {   # tts_decoder_kernel_choice
    0: 'dec_gemm_kernel / dec_attention_kernel (launch per layer: 200 steps x {} dependent launches)',
    1: 'dec_persistent_kernel (200 steps x {} hand-off phases, clusters of 8 workgroups x 16 utterances, weights streamed from L2 every step)',
    2: 'dec_ws_kernel (200 steps x {} hand-off phases, clusters of 16 workgroups x 32 utterances, each workgroup\'s share of the '
       'weights resident in registers: 172 per lane)',
}


def _sha16(paths):
    import hashlib
    h = hashlib.sha256()
    for q in paths:
        with open(q, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def gl_kernel_sha16():
    """content hash of the sources the Griffin-Lim kernel is built from: what a counter file must have been measured on"""
    c = os.path.join(ROOT, 'single-speaker-tts_amd', 'csrc')
    return _sha16([os.path.join(c, 'griffin_lim.hip'), os.path.join(c, 'griffin_lim.h')])


def _newest_profile(pattern, per_launch):
    """newest profiles/<pattern> whose record was measured on the Griffin-Lim kernel of THIS tree (its `kernel_sha16`
    equals gl_kernel_sha16()) at the same iterations per launch; (record or None, reason)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))
    if not files:
        return None, 'no file profiles/' + pattern
    with open(files[-1]) as f:
        rec = json.load(f)
    name = os.path.basename(files[-1])
    if rec.get('iterations_per_launch', 1) != per_launch:
        return None, name + ': measured at {} iterations per launch'.format(rec.get('iterations_per_launch'))
    if rec.get('kernel_sha16') != gl_kernel_sha16():
        return None, name + ': measured on other kernel sources (kernel_sha16 {} != {} of this tree)'.format(
            rec.get('kernel_sha16'), gl_kernel_sha16())
    rec = dict(rec, file='profiles/' + name)
    return rec, None


def synthetic_ids(B, Ts, seed):
    """lengths ~ clip(round(N(100,30)),20,149), ids U[2,38], EOS=1, pad=0."""
    rng = np.random.default_rng(seed)
    ids = np.zeros((B, Ts), np.int32)
    for b in range(B):
        L = int(np.clip(round(rng.normal(100, 30)), 20, Ts - 1))
        ids[b, :L] = rng.integers(2, 39, L)
        ids[b, L] = 1
    return ids


def _median_time(fn, repeats):
    """one untimed warm-up run, then the median wall time of `repeats` runs; returns (seconds, last result)"""
    out = fn()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), out


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return [{'lib': i.get('internal_api'), 'threads': i.get('num_threads')} for i in threadpool_info()]
    except Exception:   # noqa: BLE001 -- a report field, never a reason to fail the bench
        return None


def cpu_baseline(weights, hp, n_utts=6, repeats=5):
    """the numpy oracle (a restated CPU path -- NOT TensorFlow, which cannot run here) on a bounded
    sample of the bench workload, on this box's host cores: (i) the network on n_utts utterances in one process
    (float32, the BLAS library's default thread count, reported); (ii) Griffin-Lim the way the reference fans it out
    -- 6 worker processes, one utterance each (tacotron/inference.py:185-188, params/inference.py:34) -- and also
    with one worker per CPU this process may run on (os.sched_getaffinity: os.cpu_count() on a host of its own; on a GPU box
    one GPU's share, where more worker processes than that are refused) on as many utterances.  One warm-up run each, then
    the median of `repeats` (>= 5).
    `value` = frames / (network + Griffin-Lim with the reference's 6 workers)."""
    from multiprocessing import get_context
    from oracle import tacotron_oracle as O
    cores = os.cpu_count() or 1
    w32 = {k: np.asarray(v, np.float32) for k, v in weights.items()}
    ids = synthetic_ids(n_utts, TS, 4321)
    t_net, out = _median_time(lambda: O.tacotron_predict(ids, w32, hp, n_steps=N_STEPS), repeats)
    lin = out['linear'].astype(np.float32)
    n_samples = HOP * (N_STEPS * hp.reduction - 1)

    def gl(n_workers):
        jobs = [(lin[b % n_utts], b) for b in range(n_workers)]
        with get_context('fork').Pool(n_workers) as pool:
            t, wavs = _median_time(lambda: pool.map(_cpu_gl_job, jobs), repeats)
        assert all(w.shape == (n_samples,) for w in wavs)
        return t

    t_gl6 = gl(n_utts)
    # "all cores": every CPU this process may run on -- os.cpu_count() unless the box restricts the process (the GPU boxes
    # report 256 logical CPUs for eight GPUs and give one GPU's job 16 of them; worker pools beyond a job's share are
    # killed there)
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = cores
    share = max(1, min(cores, usable, int(os.environ.get('SSTTS_CPU_WORKERS', '16' if cores > 64 else str(cores)))))
    t_glc = gl(share) if share != n_utts else t_gl6
    frames = n_utts * N_STEPS * hp.reduction
    return dict(value=frames / (t_net + t_gl6), unit='mel-frames/s', cores=cores, cpus_usable=usable, repeats=repeats, kind='port',
                sample='{} utterances end-to-end (Ts={}, {} decoder steps, {} GL iterations), numpy oracle = a restated CPU '
                       'path, not TensorFlow; 1 warm-up + median of {}: network {:.2f} s in 1 process (BLAS threads: {}), '
                       'Griffin-Lim {:.2f} s in the reference\'s {}-process pool; with {} workers (one GPU\'s share of the {} host CPUs) on {} '
                       'utterances {:.2f} s'.format(
                           n_utts, TS, N_STEPS, N_ITER, repeats, t_net, _blas_threads(), t_gl6, n_utts, share, cores, share, t_glc),
                network_s=t_net, griffin_lim_s_6_workers=t_gl6, griffin_lim_s_cpu_share=t_glc, gl_workers_reference=n_utts,
                gl_workers_cpu_share=share,
                blas=_blas_threads(),
                griffin_lim_rtf=t_gl6 / (n_utts * n_samples / SR),
                griffin_lim_rtf_cpu_share=t_glc / (share * n_samples / SR))


def cpu_baseline_in_child(timeout_s=900):
    """cpu_baseline() in a fresh interpreter that never touches the GPU: its worker pools are made by fork(), and forking a
    process that holds a HIP runtime (its helper threads, their locks) now and then leaves a child waiting for a lock nobody
    in it will release -- the pool never answers and the bench hangs behind a finished measurement.  The child is a plain
    subprocess of this one (nothing is exec'ed over the process that holds the device); a child that fails or overruns is
    reported in the field, never waited for."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only']
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, check=False)
    except subprocess.TimeoutExpired:
        return {'value': None, 'error': 'CPU baseline child overran {} s'.format(timeout_s)}
    lines = [ln for ln in r.stdout.decode(errors='replace').splitlines() if ln.startswith('{')]
    if r.returncode != 0 or not lines:
        return {'value': None, 'error': 'CPU baseline child failed (rc {}): {}'.format(
            r.returncode, r.stderr.decode(errors='replace')[-400:])}
    return json.loads(lines[-1])


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


# What a rank process needs from its environment, whoever launched it (this file's self_launch or the driver's
# `python -m torch.distributed.run ... bench.py --gpus N`): applied by rank_env() at the top of main(), BEFORE torch
# or the HIP runtime is loaded, so the two launch paths cannot differ.
#   HSA_ENABLE_IPC_MODE_LEGACY=0 -- the host driver of this pool only supports dmabuf IPC; with the legacy mode RCCL's
#   (and torch's) cross-process buffer sharing fails with `hipIpcGetMemHandle: invalid argument` when the ranks of one
#   node set up their communicator.  A value the launcher exported is kept.
RANK_ENV = {'HSA_ENABLE_IPC_MODE_LEGACY': '0'}


def rank_cpu_share(cpus, local_rank, local_world):
    """the contiguous share of the CPUs this process may use that belongs to local rank `local_rank` of `local_world`"""
    cpus = sorted(cpus)
    n = len(cpus)
    if local_world <= 1 or n < local_world:
        return cpus
    return cpus[local_rank * n // local_world:(local_rank + 1) * n // local_world]


def rank_env():
    """Environment and CPU binding of a rank process; returns what was applied (reported in the JSON line and compared
    between the two launch paths by tests/test_distributed_cpu.py).  The call path blocks in hipEventSynchronize and
    hipStreamSynchronize (csrc/api_pipeline.hip, tts_synthesize): eight ranks' host threads wandering over all cores of the node
    wake each other's caches; each rank keeps to its share."""
    applied = {}
    if 'RANK' in os.environ:
        for k, v in RANK_ENV.items():
            os.environ.setdefault(k, v)
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1')))
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        if local_world > 1 and hasattr(os, 'sched_setaffinity'):
            share = rank_cpu_share(os.sched_getaffinity(0), local_rank, local_world)
            try:
                os.sched_setaffinity(0, share)
            except OSError:
                pass
    applied.update({k: os.environ.get(k) for k in RANK_ENV})
    applied['n_cpus_bound'] = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None
    return applied


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes, one rank per GPU, wired up
    the way `python -m torch.distributed.run --nproc-per-node N` would (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  The parent never touches the GPU (no HIP call, no torch.cuda): the children are
    plain subprocesses, nothing is exec'ed over a process that holds a device."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        # (exactly the variables torch.distributed.run sets; everything else a rank needs it sets itself: rank_env())
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        # only rank 0 prints the result; the others' stdout is dropped (an unread pipe would block a chatty child)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # the first rank that fails ends the run: its siblings would otherwise sit in a collective until the backend's
    # timeout (they are the exact processes started above, ended by PID)
    rc = 0
    live = dict(enumerate(procs))
    while live and rc == 0:
        for r, pr in list(live.items()):
            code = pr.poll()
            if code is not None:
                del live[r]
                rc = rc or code
        if live and rc == 0:
            time.sleep(0.05)
    for pr in live.values():
        pr.terminate()
    for pr in live.values():
        try:
            pr.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pr.kill()
            pr.wait()
    return rc


def timed_broadcasts(dist, shard, blob, bdev, rank):
    """The ONE collective of the path, twice, behind a barrier that takes the communicator set-up (the first collective of
    the process group): the second broadcast is what the 27.4 MB cost on the wire once everything is warm.
    Returns (blob, setup_ms, first_ms, steady_ms)."""
    t0 = time.perf_counter()
    dist.barrier()
    setup_ms = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter()
    blob = shard.broadcast_blob(blob, src=0, device=bdev)
    dist.barrier()
    first_ms = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter()
    again = shard.broadcast_blob(blob, src=0, device=bdev)
    dist.barrier()
    steady_ms = 1e3 * (time.perf_counter() - t0)
    if not np.array_equal(again, blob):
        raise SystemExit('rank {}: the second weight broadcast differs from the first'.format(rank))
    return blob, setup_ms, first_ms, steady_ms


def gather_rank_devices(dist, world, rank, dev_uuid, dev_cus):
    """every rank's (host, device uuid, compute units): under "nccl" no two ranks may sit on one device (one process per
    GPU); under gloo (rehearsals on a box with fewer GPUs than ranks) sharing is allowed and reported"""
    import socket
    devs = [None] * world
    dist.all_gather_object(devs, (socket.gethostname(), dev_uuid, dev_cus))
    if dist.get_backend() == 'nccl' and len({(d[0], d[1]) for d in devs}) != world:
        raise SystemExit('rank {}: two ranks share a device: {}'.format(rank, devs))
    return devs


def check_shards(dist, world, rank, padded_len, n_utts):
    """every shard is padded to the GLOBAL sentence length (the reference masks nothing: sharding.py) and holds B_PER_GPU
    utterances: the same on all ranks, or the ranks do not compute what one process would"""
    shapes = [None] * world
    dist.all_gather_object(shapes, (int(padded_len), int(n_utts)))
    if len({sh[0] for sh in shapes}) != 1 or any(sh[1] != B_PER_GPU for sh in shapes):
        raise SystemExit('rank {}: shards differ in padded length or size: {}'.format(rank, shapes))
    return shapes[0][0]


def gather_rank_ms(dist, world, own_ms):
    """each rank's own ms per step (its clock stopped when ITS device was idle, before the closing barrier)"""
    own = [None] * world
    dist.all_gather_object(own, float(own_ms))
    return own


def rank_fields(rank_ms, rank_devices, first_ms, steady_ms, padded_len, setup_ms=None):
    """the N > 1 diagnostics of the JSON line (None / one entry for a single process)"""
    return {
        'communicator_setup_ms': setup_ms,            # the first barrier = the first collective of the process group
        'weight_broadcast_ms': first_ms,              # the weight broadcast of the run (first use of the broadcast)
        'weight_broadcast_ms_steady': steady_ms,      # the same 27.4 MB again
        'rank_ms_per_step': rank_ms,
        'rank_ms_per_step_min': min(rank_ms) if rank_ms else None,
        'rank_ms_per_step_max': max(rank_ms) if rank_ms else None,
        'straggler_rank': int(np.argmax(rank_ms)) if rank_ms else None,
        'rank_devices': [{'host': d[0], 'uuid': d[1], 'compute_units': d[2]} for d in rank_devices],
        'padded_sentence_length': padded_len,
    }


def gather_rank_env(dist, world, env_applied):
    """every rank's rank_env() record: one entry per rank, all equal in their environment part"""
    envs = [None] * world
    dist.all_gather_object(envs, env_applied)
    return envs


def dist_selftest(rank, local_rank, world, dist, env_applied=None):
    """--dist-selftest: everything of the N > 1 path except the GPU work (process group, the one weight broadcast,
    shard ranges, the max-over-ranks reduction), so that the launcher can be rehearsed on a box without GPUs."""
    shard = importlib.import_module('single-speaker-tts_amd.sharding')
    Wm = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
    n = Wm.n_parameters()
    blob = Wm.pack_blob(Wm.synthetic_weights(0)) if rank == 0 else np.zeros(n, np.float32)
    got, setup_ms, first_ms, steady_ms = timed_broadcasts(dist, shard, blob, 'cpu', rank)
    import torch
    t = torch.tensor([float(np.abs(got).sum()), float(rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    lo, hi = shard.shard_range(world * B_PER_GPU, world, rank)
    ok = abs(float(np.abs(got).sum()) - float(t[0])) < 1e-6 and hi - lo == B_PER_GPU
    # the diagnostics of the real run, on stand-in values: a device id per rank, the shard check, per-rank times
    devs = gather_rank_devices(dist, world, rank, 'selftest-rank{}'.format(rank), 0)
    padded = check_shards(dist, world, rank, synthetic_ids(world * B_PER_GPU, TS, 1234)[lo:hi].shape[1], hi - lo)
    rank_ms = gather_rank_ms(dist, world, 1.0 + rank)   # (rank r "took" 1 + r ms: the last rank is the straggler)
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    envs = gather_rank_env(dist, world, env_applied or rank_env())
    if rank == 0:
        rec = {'selftest': 'dist', 'n_gpus': world, 'world_size_seen': dist.get_world_size(),
               'backend': dist.get_backend(), 'ok': all(flags)}
        rec.update(rank_fields(rank_ms, devs, first_ms, steady_ms, padded, setup_ms))
        rec['rank_env'] = envs
        print(json.dumps(rec), flush=True)
    dist.barrier()
    dist.destroy_process_group()


TRAINED_SPEC = os.path.join(ROOT, 'tests', 'golden', 'reference_linear_spec_post_215k.npz')


def gl_beside_decoder(sstts, eng, hp, blob, ids, mags, B, T, dev_cus, reserve, per_launch, n_steps, n_launches=5, reps=4,
                      with_decoder=True, decoder_stream=None):
    """Griffin-Lim launches of the bench's form (per_launch iterations each, planned for dev_cus - reserve workgroups) on
    the magnitude spectrograms `mags` ({name: (B, F, T) device array}), each BESIDE a running persistent decoder: the decoder
    runs on a SECOND handle (its own stream) on the `reserve` compute units Griffin-Lim leaves free, as it does under the
    call pipeline; n_launches launches fit under one decoder (8.9 ms).  Returns {name: ms per iteration} from the library's
    HIP events around the launches (tts_profile_get "gl_iter")."""
    eng2 = sstts.Engine(hp, device_id=eng.device_id)
    out = {}
    try:
        if decoder_stream is not None:
            eng2.set_stream(decoder_stream)
        eng2.load_weights_blob(blob)
        eng2.set_option('persistent_decoder', 2)
        eng2.set_option('debug_hooks', 1)
        eng2.set_option('pd_rows', 32)   # the pipeline's form: 32 workgroups on the `reserve` units (a stand-alone call would take 64)
        mem2 = eng2.encoder_forward(ids)
        mel2, al2 = eng2.decoder_forward(mem2, n_steps)
        eng2.synchronize()
        eng.set_option('profile', 1)
        eng.set_option('debug_hooks', 1)
        eng.set_option('gl_workers', dev_cus - reserve)
        init = eng.to_device(np.random.default_rng(7).random((B, 1 + N_FFT // 2, T), dtype=np.float32))
        n_iter = per_launch * n_launches
        eng.griffin_lim(next(iter(mags.values())), n_iter, WIN, HOP, N_FFT, init_phase=init, want_mse=False)   # (workspaces, the cut)
        eng.synchronize()
        tot = {name: [0.0, 0] for name in mags}
        for _ in range(reps):   # the spectra take turns inside every repetition: drift of the box's clocks hits all alike
            for name, mag in mags.items():
                eng.profile_reset()
                if with_decoder:
                    eng2.decoder_forward(mem2, n_steps, mel=mel2, alignments=al2)   # enqueued first: resident on the free units
                eng.griffin_lim(mag, n_iter, WIN, HOP, N_FFT, init_phase=init, want_mse=False)
                eng.synchronize()
                eng2.synchronize()
                ms, n = eng.profile_get('gl_iter')
                tot[name][0] += ms
                tot[name][1] += n
        out = {name: t[0] / max(1, t[1]) for name, t in tot.items()}
        init.free()
    finally:
        eng.set_option('gl_workers', 0)
        eng.set_option('debug_hooks', 0)
        eng2.close()
    return out


def trained_spectrum_batch(B, T):
    """The one model output the reference ships (a (1, 1025, 1000, 1) linear-spectrogram dump after 215k training steps,
    reference visualization/data/ljspeech/v1.1/post-processing/, copied as data to tests/golden/) as B magnitude
    spectrograms: every row rolled in time by another offset and scaled by another gain (exp(U(-0.7, 0.7)))."""
    if not os.path.exists(TRAINED_SPEC):
        return None
    spec = np.load(TRAINED_SPEC)['linear_spec'][0, :, :, 0].astype(np.float32)   # (F, T0)
    if spec.shape[1] < T:
        spec = np.tile(spec, (1, -(-T // spec.shape[1])))
    spec = spec[:, :T]
    rng = np.random.default_rng(215)
    gains = np.exp(rng.uniform(-0.7, 0.7, B)).astype(np.float32)
    return np.stack([np.roll(spec, 37 * b, axis=1) * gains[b] for b in range(B)])



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
    ap.add_argument('--pd-ws', type=int, default=None, help='override the library default (1: the weight-stationary persistent decoder, 0: decoder_persistent.hip)')
    ap.add_argument('--set', action='append', default=[], metavar='KEY=VALUE', help='any other tts_set_option (A/B runs)')
    ap.add_argument('--gl-pair', type=int, default=None, help='override the library default (Griffin-Lim iterations per launch, 1..3)')
    ap.add_argument('--enc-stream', type=int, default=None, help='override the library default (1: the encoder on a stream of its own, a gap ahead of its decoder)')
    ap.add_argument('--through-facade', action='store_true',
                    help='also time the reference-shaped host API (tacotron.inference.synthesize_stream: host ids in, host '
                         'waveforms out, upload and download inside the timed region) and report facade_ms_per_step')
    ap.add_argument('--no-aux-outputs', action='store_true', help='A/B: do not write the linear spectrograms and alignments')
    ap.add_argument('--dist-selftest', action='store_true', help='rehearse the N > 1 launch path without GPU work')
    ap.add_argument('--cpu-baseline-only', action='store_true',
                    help='print the cpu_baseline object and exit (no GPU call is made: how the bench runs it, in a child)')
    ap.add_argument('--cpu-baseline-utts', type=int, default=6, help='with --cpu-baseline-only: utterances of the sample (tests: 1)')
    ap.add_argument('--cpu-baseline-repeats', type=int, default=5, help='with --cpu-baseline-only: timed repeats (tests: 1)')
    args = ap.parse_args()

    # a bench that stops answering says where: every 4 minutes without an exit, all thread stacks on stderr
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)

    if args.cpu_baseline_only:
        P = importlib.import_module('single-speaker-tts_amd.tacotron.params')
        Wm = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
        hp = P.ModelParams()
        print(json.dumps(cpu_baseline(Wm.synthetic_weights(0, hp), hp, n_utts=args.cpu_baseline_utts,
                                      repeats=args.cpu_baseline_repeats)), flush=True)
        return

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    env_applied = rank_env()   # before torch / the HIP runtime: the same for both launch paths
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world > 1:
        raise SystemExit('WORLD_SIZE ({}) != --gpus ({})'.format(world, args.gpus))
    dist = None
    # SSTTS_DIST_SINGLE=1: bring the process group up for a single rank too (a one-GPU box cannot hold two RCCL ranks; this
    # is how far the RCCL path -- communicator, broadcast, barrier, all-reduce beside the library's streams -- can be
    # rehearsed there: tests/test_gpu_bench_launch.py)
    if world > 1 or (os.environ.get('SSTTS_DIST_SINGLE') == '1' and 'RANK' in os.environ):
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
        return dist_selftest(rank, local_rank, world, dist, env_applied)

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
    setup_ms = None              # the first collective (a barrier): communicator set-up
    broadcast_ms = None          # the weight broadcast
    broadcast_ms_steady = None   # the same broadcast again: what the 27.4 MB cost on the wire
    if dist is not None:
        # the ONE collective of the path (RCCL over xGMI under "nccl"); nothing is exchanged afterwards
        if dist.get_world_size() != args.gpus:
            raise SystemExit('process group has {} ranks, --gpus says {}'.format(dist.get_world_size(), args.gpus))
        bdev = 'cuda:{}'.format(local_rank) if dist.get_backend() == 'nccl' else 'cpu'
        blob, setup_ms, broadcast_ms, broadcast_ms_steady = timed_broadcasts(dist, shard, blob, bdev, rank)
    eng = sstts.Engine(hp, device_id=local_rank)
    if dist is not None and dist.get_backend() == 'nccl':
        import torch
        # one process per GPU: this rank's handle and its RCCL communicator sit on the same device, a different one per rank
        if eng.device_id != local_rank or torch.cuda.current_device() != local_rank:
            raise SystemExit('rank {}: handle on device {}, communicator on {}, LOCAL_RANK {}'.format(
                rank, eng.device_id, torch.cuda.current_device(), local_rank))
    dev_uuid, dev_cus = eng.device_info()
    rank_devices = [(None, dev_uuid, dev_cus)]
    rank_envs = [env_applied]
    if dist is not None:
        rank_devices = gather_rank_devices(dist, world, rank, dev_uuid, dev_cus)
        rank_envs = gather_rank_env(dist, world, env_applied)
        if len({e.get('HSA_ENABLE_IPC_MODE_LEGACY') for e in rank_envs}) != 1:
            raise SystemExit('rank {}: the ranks differ in HSA_ENABLE_IPC_MODE_LEGACY: {}'.format(rank, rank_envs))
    eng.load_weights_blob(blob)
    if args.pipeline is not None:
        eng.set_option('pipeline', args.pipeline)
    if args.reserve_cus is not None:
        eng.set_option('reserve_cus', args.reserve_cus)
    if args.hold_lds_kb is not None:
        eng.set_option('hold_lds_kb', args.hold_lds_kb)
    if args.persistent_decoder is not None:
        eng.set_option('persistent_decoder', args.persistent_decoder)
    if args.pd_ws is not None:
        eng.set_option('pd_ws', args.pd_ws)
    for kv in args.set:
        k, v = kv.split('=')
        eng.set_option(k, int(v))
    if args.gl_pair is not None:
        eng.set_option('gl_pair', args.gl_pair)
    if args.enc_stream is not None:
        eng.set_option('enc_stream', args.enc_stream)

    # ---- this rank's shard of the synthetic batch, resident in HBM
    lo, hi = shard.shard_range(world * B_PER_GPU, world, rank)
    ids_all = synthetic_ids(world * B_PER_GPU, TS, 1234)
    ids = eng.to_device(ids_all[lo:hi])
    B = hi - lo
    padded_len = int(ids_all[lo:hi].shape[1])
    if dist is not None:
        padded_len = check_shards(dist, world, rank, padded_len, B)
    T = N_STEPS * hp.reduction
    F = 1 + N_FFT // 2
    init = eng.to_device(np.random.default_rng(42 + rank).random((B, F, T), dtype=np.float32))
    wav = eng.empty((B, HOP * (T - 1)))
    # every output north_star names is written to HBM in every timed step: the waveforms, the linear spectrograms
    # (262 MB, an extra store of the final Dense) and the alignments into caller buffers; the mel spectrograms into the
    # library's own pair of buffers (one per call parity -- a caller's single mel buffer would make the decoder of call
    # k + 1 wait for the post-net of call k)
    lin_out = None if args.no_aux_outputs else eng.empty((B, T, F))
    ali_out = None if args.no_aux_outputs else eng.empty((N_STEPS, B, TS))

    # Initial phases: drawn per call inside the path, as the reference does (np.random.rand in audio/synthesis.py:91) -- here
    # a counter-based draw from a per-call seed, made by the first Griffin-Lim launch itself (csrc/griffin_lim.hip,
    # gl_seed_phasor).  `init` (an explicit (B, F, T) array, what the parity tests pass) is only used by the stand-alone
    # Griffin-Lim measurement below.
    calls = [0]

    def step():
        calls[0] += 1
        eng.synthesize(ids, N_STEPS, REF_DB, MAX_DB, POWER, N_ITER, WIN, HOP, seed=1000 * (rank + 1) + calls[0],
                       peak_normalize=True, wav=wav, want_linear=lin_out, want_alignments=ali_out)

    own_elapsed = [0.0]
    t_start = [0.0]

    def barrier():
        eng.synchronize()
        own_elapsed[0] = time.perf_counter() - t_start[0]   # (meaningful for the closing barrier: own work done)
        if dist is not None:
            dist.barrier()
        eng.synchronize()

    for _ in range(args.warmup):
        step()
    eng.set_option('profile', 1)
    eng.profile_reset()
    barrier()
    t0 = time.perf_counter()
    t_start[0] = t0
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = None
    if dist is not None:
        import torch
        # this rank's own time: the clock stopped when ITS device was idle, before the closing barrier (what `elapsed`
        # includes); gathered so that a straggler is named, not just suffered
        rank_ms = gather_rank_ms(dist, world, 1e3 * own_elapsed[0] / args.steps)
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
    eng.griffin_lim(mag_alone, 3 * GL_PER_LAUNCH, WIN, HOP, N_FFT, init_phase=init, want_mse=False)
    ms_alone, n_alone = eng.profile_get('gl_iter')
    gl_alone_ms = ms_alone / max(1, n_alone)
    mag_alone.free()
    # The contract's synthetic weights put half of the linear-spectrogram bins on the clip floor (below): Griffin-Lim beside a
    # running decoder on spectra with a TRAINED model's dynamic range, and -- same harness, same box -- on the spectra of
    # the timed workload.  Reported beside the contract's figure, never instead of it.
    floor_frac = None
    gl_harness = {}
    if rank == 0 and lin_out is not None:
        reserve = args.reserve_cus if args.reserve_cus is not None else 32
        per_launch_h = args.gl_pair if args.gl_pair is not None else GL_PER_LAUNCH
        lin_host = lin_out.to_host()
        floor_frac = float((lin_host <= 0.0).mean())
        del lin_host
        mags = {'contract': eng.denorm_power(lin_out, REF_DB, MAX_DB, POWER)}
        trained = trained_spectrum_batch(B, T)
        if trained is not None:
            mags['trained'] = eng.to_device(trained)
        if dev_cus - reserve >= 16:
            gl_harness = gl_beside_decoder(sstts, eng, hp, blob, ids, mags, B, T, dev_cus, reserve, per_launch_h, N_STEPS)
        for m in mags.values():
            m.free()
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

    facade_ms = None
    if args.through_facade:
        # the same batch through the reference-shaped functions: ids from host memory, waveforms back in host memory
        # (pinned buffers of the library), two calls in flight; initial phases from the seed as the reference draws them
        Tm = importlib.import_module('single-speaker-tts_amd.tacotron.model')
        Inf = importlib.import_module('single-speaker-tts_amd.tacotron.inference')
        model = Tm.Tacotron(inputs=Tm.Tacotron.model_placeholders(), mode=Tm.Mode.PREDICT, engine=eng, hparams=hp)
        ids_host = ids_all[lo:hi]
        n_f = args.steps
        for _ in Inf.synthesize_stream(model, (ids_host for _ in range(max(2, args.warmup))), n_steps=N_STEPS, n_iter=N_ITER,
                                       peak_normalize=True):
            pass
        barrier()
        t0 = time.perf_counter()
        checksum = 0.0
        for wavs in Inf.synthesize_stream(model, (ids_host for _ in range(n_f)), n_steps=N_STEPS, n_iter=N_ITER, peak_normalize=True):
            checksum += float(wavs[0, 1000])   # the host really reads what came back
        barrier()
        facade_ms = 1e3 * (time.perf_counter() - t0) / n_f
        assert np.isfinite(checksum)

    if rank == 0:
        frames_total = world * B_PER_GPU * T
        ms_per_step = 1e3 * elapsed / args.steps
        audio_s = B_PER_GPU * HOP * (T - 1) / SR
        # dominant kernel: gl_stream_kernel<0, 1102, 275, false, 3>, one launch = GL_PER_LAUNCH Griffin-Lim iterations over
        # the batch (the library counts the stage in iterations).  Algorithmic bytes per iteration: 20 B
        # per bin = |S| 4 + phase in 8 + phase out 8.
        gl_iter_ms = stage_ms['gl_iter'] / max(1, launches['gl_iter'])
        per_launch = args.gl_pair if args.gl_pair is not None else GL_PER_LAUNCH
        gl_launch_ms = gl_iter_ms * per_launch
        alg_bytes = 20.0 * F * T * B_PER_GPU * per_launch
        achieved = alg_bytes / (gl_launch_ms * 1e-3) / 1e9 if gl_launch_ms > 0 else 0.0
        # HBM bytes per launch and the SQ counters of the kernel come from PMC passes (rocprofv3 cannot run inside this
        # process): the newest profiles/*gl_iter_hbm_bytes_per_launch.json / *gl_iter_valu.json -- used ONLY when the record
        # says it was measured on the Griffin-Lim kernel sources of this tree (kernel_sha16), reported as stale otherwise
        traffic_rec, traffic_why = _newest_profile('*gl_iter_hbm_bytes_per_launch.json', per_launch)
        traffic = traffic_rec.get('hbm_bytes_per_launch') if traffic_rec else None
        valu_rec, valu_why = _newest_profile('*gl_iter_valu.json', per_launch)
        # VALU issue: a SIMD issues one VALU wave-instruction per 4 cycles at best (MI355X_MICROARCH.md, cycle constants);
        # fraction = wave-instructions x 4 / (SIMDs the launch ran on x cycles of the launch) at the clock the counter record
        # measured (shader_clock_mhz), on the n_cus - reserve_cus compute units Griffin-Lim is planned for
        valu_issue_frac = None
        if valu_rec and valu_rec.get('valu_wave_insts_per_launch') and gl_launch_ms > 0:
            gl_cus = max(1, dev_cus - (args.reserve_cus if args.reserve_cus is not None else 32))
            clock_hz = 1e6 * (valu_rec.get('shader_clock_mhz') or 2400.0)
            valu_issue_frac = valu_rec['valu_wave_insts_per_launch'] * 4.0 / (4 * gl_cus * gl_launch_ms * 1e-3 * clock_hz)
        dec_choice = eng.decoder_kernel_choice(B_PER_GPU, TS, pipelined=(args.pipeline is None or args.pipeline != 0))
        dec_ms = stage_ms['decoder']
        dec_flop = DEC_MFLOP_PER_UTT_STEP * 1e6 * B_PER_GPU * N_STEPS
        gemm_tflops = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        out = {
            'metric': 'mel-frames/sec (end-to-end text->waveform incl. 60-iter Griffin-Lim, 64-utt LJ-Speech-shaped batch per GPU)',
            'value': frames_total * args.steps / elapsed,
            'unit': 'mel-frames/s',
            'n_gpus': world,
            'world_size_seen': dist.get_world_size() if dist is not None else 1,
            # N > 1 diagnostics: the first broadcast (communicator set-up included) against the same 27.4 MB again; every
            # rank's own ms per step and the slowest rank; the devices the ranks sat on; the global padded length
            **rank_fields(rank_ms, rank_devices, broadcast_ms, broadcast_ms_steady, padded_len, setup_ms),
            'rank_env': rank_envs,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            # f32 throughout; the GEMMs form their f32 products from exact three-way bf16 splits of both operands (6 bf16 MFMAs,
            # f32 accumulation: same error against the float64 oracle as the f32 MFMA, tests/test_gpu_gemm.py)
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'end-to-end Tacotron inference, B=64/GPU, T_sent=150, 200 decoder steps (r=5, 1000 frames), '
                                   '1025-bin linear, Griffin-Lim 60 iterations, n_fft 2048 / win 1102 / hop 275',
                       'global_batch': world * B_PER_GPU, 'parallelism': 'utterance shards, dp{}'.format(world)},
            'griffin_lim_rtf': (stage_ms['gl_iter'] + stage_ms['gl_final']) * 1e-3 / audio_s,
            'end_to_end_rtf': ms_per_step * 1e-3 / audio_s,
            'mel_frames_per_sec_encoder_decoder': B_PER_GPU * T / ((stage_ms['encoder'] + stage_ms['decoder']) * 1e-3),
            'stage_ms': stage_ms,
            'facade_ms_per_step': facade_ms,
            'outputs_per_step': 'wav (peak-normalised)' + ('' if args.no_aux_outputs else ', linear spectrograms, alignments') +
                                ', mel (library-owned double buffer); all resident in HBM, none copied to the host in the timed region',
            # SURVEY.md 8(d) / the bench contract: `achieved` = ALGORITHMIC bytes (20 B per bin and iteration) / launch time.
            # The kernel no longer moves those bytes (three iterations per launch pass the spectrum on in registers): what
            # the memory system really carries is `traffic` / `achieved_traffic` / `frac_traffic`, and what bounds the launch
            # is VALU issue (`limiter`, `roofline_valu`), neither roof.
            'roofline': {'kernel': 'gl_stream_kernel<0, 1102, 275, false, {0}> ({0} Griffin-Lim iterations per launch: iSTFT + STFT '
                                   'of every iteration fused, the spectrum passed from one iteration to the next in registers)'.format(per_launch),
                         # what the counters show (roofline_valu): the launch is bound by VALU issue, not by either roof of the
                         # contract; achieved / peak / frac stay the contract's HBM figures (algorithmic bytes)
                         'bound': 'valu-issue',
                         'valu_issue_frac': valu_issue_frac,
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic,
                         'traffic_source': ({'file': traffic_rec['file'], 'commit': traffic_rec.get('commit'),
                                             'kernel_sha16': traffic_rec.get('kernel_sha16')} if traffic_rec else None),
                         'traffic_stale': traffic_why,
                         'achieved_traffic': (traffic / (gl_launch_ms * 1e-3) / 1e9) if (traffic and gl_launch_ms > 0) else None,
                         'frac_traffic': (traffic / (gl_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and gl_launch_ms > 0) else None,
                         'limiter': 'VALU issue (roofline_valu), not HBM: the launch moves about a third of the algorithmic bytes',
                         'launch_ms': gl_launch_ms, 'iterations_per_launch': per_launch,
                         'iteration_ms': gl_iter_ms, 'iteration_ms_alone': gl_alone_ms,
                         # Griffin-Lim on a trained model's dynamic range (the spectrogram the reference ships, tiled to the
                         # batch with per-row gain and time offsets) and on the timed workload's own spectra, both as
                         # launches of the bench's form on n_cus - reserve_cus workgroups BESIDE a running persistent decoder
                         # (second handle); `floor_bin_frac_timed_workload`: share of the timed steps' linear-spectrogram bins
                         # on the clip floor (<= 0) -- the contract's random weights, SURVEY.md 8(d)
                         'iteration_ms_trained_spectrum': gl_harness.get('trained'),
                         'iteration_ms_contract_spectrum_same_harness': gl_harness.get('contract'),
                         'floor_bin_frac_timed_workload': floor_frac,
                         'algorithmic_bytes_per_launch': alg_bytes,
                         'note': 'achieved / frac = algorithmic bytes (20 B per bin and iteration, SURVEY.md 8(d)) / launch time, as the '
                                 'contract defines them; achieved_traffic / frac_traffic = measured HBM bytes of the same launch (a launch '
                                 'requests 4 B phasor code + 4 B |S| per bin in its first iteration, 4 B |S| in each further one and writes '
                                 '4 B code in its last)'},
            # what bounds the dominant kernel: SQ counters of the same kernel sources (separate --pmc passes, tools/gl_pmc.sh)
            'roofline_valu': ({'kernel': 'gl_stream_kernel<0, 1102, 275, false, {}>'.format(per_launch),
                               'valu_wave_insts_per_launch': valu_rec.get('valu_wave_insts_per_launch'),
                               'valu_busy_frac': valu_rec.get('valu_busy_frac'),
                               'fp_share': valu_rec.get('fp_share'),
                               'shader_clock_mhz': valu_rec.get('shader_clock_mhz'),
                               'lds_bank_conflict_frac': valu_rec.get('lds_bank_conflict_frac'),
                               'source': {'file': valu_rec['file'], 'commit': valu_rec.get('commit'),
                                          'kernel_sha16': valu_rec.get('kernel_sha16')}} if valu_rec else
                              {'kernel': 'gl_stream_kernel', 'stale': valu_why}),
            # the latency-bound loop ("report achieved MFMA fraction and steps/s")
            'roofline_decoder': {'kernel': DEC_KERNELS[dec_choice].format(DEC_PHASES_PER_STEP),
                                 'kernel_choice': dec_choice,
                                 'bound': 'latency',
                                 'ms': dec_ms,
                                 'steps_per_s': N_STEPS / (dec_ms * 1e-3) if dec_ms > 0 else None,
                                 'utterance_steps_per_s': B_PER_GPU * N_STEPS / (dec_ms * 1e-3) if dec_ms > 0 else None,
                                 'tflops': dec_flop / (dec_ms * 1e-3) / 1e12 if dec_ms > 0 else None,
                                 'mfma_frac': (dec_flop / (dec_ms * 1e-3) / 1e12) / MFMA_F32_PEAK_TFLOPS if dec_ms > 0 else None,
                                 'mfma_peak': MFMA_F32_PEAK_TFLOPS,
                                 'us_per_step': 1e3 * dec_ms / N_STEPS,
                                 'us_per_phase': 1e3 * dec_ms / (N_STEPS * DEC_PHASES_PER_STEP),
                                 'flop_per_call': dec_flop,
                                 'note': 'measured beside the Griffin-Lim launches of the previous call (stage_ms.decoder)'},
            # f32 GEMM on the bf16 matrix pipe: every f32 operand split exactly into three bf16 terms, six bf16 MFMAs per
            # product block, f32 accumulation (csrc/gemm_f32.hip).  `achieved` counts the f32-equivalent flops 2 M N K;
            # `peak` = dense bf16 MFMA peak / 6; `frac_of_f32_mfma_peak` prices the same flops against the f32-input MFMA
            # the kernel used until round 3 (> 1 is possible: that is the point of the split)
            'roofline_mfma': {'kernel': 'gemm_f32_pool_kernel (post-net projection 1: conv1d k=3, 1024 -> 256, max-pool in the loader, '
                                        'M = {}; f32 operands as 3 x bf16, 6 x v_mfma_f32_32x32x16_bf16 per 16 k)'.format(M),
                              'bound': 'mfma', 'achieved': gemm_tflops,
                              'peak': MFMA_BF16_PEAK_TFLOPS / GEMM_PRODUCTS, 'unit': 'TFLOP/s (f32-equivalent)',
                              'frac': gemm_tflops / (MFMA_BF16_PEAK_TFLOPS / GEMM_PRODUCTS),
                              'frac_of_f32_mfma_peak': gemm_tflops / MFMA_F32_PEAK_TFLOPS,
                              'bf16_mfma_tflops': gemm_tflops * GEMM_PRODUCTS, 'bf16_mfma_peak': MFMA_BF16_PEAK_TFLOPS,
                              'traffic': None, 'launch_ms': gemm_ms, 'flop_per_launch': gemm_flop},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_in_child()
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
