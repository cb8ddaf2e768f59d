"""Builds libsstts_hip.so (hand-written HIP, gfx950) in-tree with hipcc.

    python single-speaker-tts_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libsstts_hip.so')
SOURCES = ['gemm_f32.hip', 'cbhg_tail.hip', 'gru.hip', 'decoder.hip', 'decoder_persistent.hip', 'decoder_ws.hip', 'griffin_lim.hip', 'griffin_lim_generic.hip', 'reserve.hip', 'api_handle.hip', 'api_stages.hip', 'api_pipeline.hip']
HEADERS = ['tts_common.h', 'decoder.h', 'griffin_lim.h', 'api_internal.h', os.path.join('..', '..', 'include', 'sstts_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-Wno-unused-result']
# Packed f32 VALU ops (v_pk_add/mul/fma_f32) issue slower than the two scalar ops they replace on gfx950 and
# need aligned register pairs (extra v_mov); the SLP vectoriser forms them from complex arithmetic.  Measured
# on the wave-level FFT: 2.70 -> 2.02 us, 92 -> 67 VGPRs (tools/fft_microbench.hip).
# griffin_lim_generic.hip: no packed-f32 selection at all (its complex type is two scalars; see the note there -- packed results
# were stored wrong when MFMA waves of another stream shared the compute unit)
EXTRA_FLAGS = {'griffin_lim.hip': ['-fno-slp-vectorize'], 'griffin_lim_generic.hip': ['-fno-slp-vectorize']}


def _digest(paths, extra=()):
    """sha256 over the CONTENT of the inputs (and the command line): what decides whether an object is rebuilt.
    Modification times say nothing in a fresh checkout or on a box the tree was copied to."""
    h = hashlib.sha256()
    for e in extra:
        h.update(e.encode() + b'\0')
    for p in paths:
        h.update(os.path.basename(p).encode() + b'\0')
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_ok(target, digest):
    stamp = target + '.sha256'
    if not (os.path.exists(target) and os.path.exists(stamp)):
        return False
    with open(stamp) as f:
        return f.read().strip() == digest


def _write_stamp(target, digest):
    with open(target + '.sha256', 'w') as f:
        f.write(digest + '\n')


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', 'hipcc')
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    digests = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        flags = FLAGS + EXTRA_FLAGS.get(src, [])
        d = _digest([s] + hdrs, flags)
        objs.append(o)
        digests.append(d)
        if force or not _stamp_ok(o, d):
            cmd = [hipcc] + flags + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, o, d, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, o, d, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors='replace'))
            raise RuntimeError('hipcc failed on ' + src)
        _write_stamp(o, d)
    # objects nobody lists any more (renamed / removed sources) must not travel to the GPU box
    keep = {os.path.basename(o) for o in objs} | {os.path.basename(o) + '.sha256' for o in objs}
    for f in os.listdir(objdir):
        if f not in keep:
            os.remove(os.path.join(objdir, f))
    lib_digest = hashlib.sha256('\n'.join(digests).encode()).hexdigest()
    if force or procs or not _stamp_ok(LIB, lib_digest):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        _write_stamp(LIB, lib_digest)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
