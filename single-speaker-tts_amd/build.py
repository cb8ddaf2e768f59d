"""Builds libsstts_hip.so (hand-written HIP, gfx950) in-tree with hipcc.

    python single-speaker-tts_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libsstts_hip.so')
SOURCES = ['gemm_f32.hip', 'gru.hip', 'decoder.hip', 'decoder_persistent.hip', 'griffin_lim.hip', 'reserve.hip', 'api.hip']
HEADERS = ['tts_common.h', 'decoder.h', 'griffin_lim.h', os.path.join('..', '..', 'include', 'sstts_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-Wno-unused-result']
# Packed f32 VALU ops (v_pk_add/mul/fma_f32) issue slower than the two scalar ops they replace on gfx950 and
# need aligned register pairs (extra v_mov); the SLP vectoriser forms them from complex arithmetic.  Measured
# on the wave-level FFT: 2.70 -> 2.02 us, 92 -> 67 VGPRs (tools/fft_microbench.hip).
EXTRA_FLAGS = {'griffin_lim.hip': ['-fno-slp-vectorize']}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', 'hipcc')
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors='replace'))
            raise RuntimeError('hipcc failed on ' + src)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
