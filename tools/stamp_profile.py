#!/usr/bin/env python3
"""Copy a counter record from gpurun_out/ into profiles/ with the commit it was measured on:

    python tools/stamp_profile.py gpurun_out/r04/x.json profiles/r04_x.json

`commit` = git HEAD of this checkout (the GPU box has no .git); refuses when the record's kernel_sha16 is not the hash of
the Griffin-Lim sources of this tree (the record would be refused by bench.py anyway)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
rec = json.load(open(src))
if rec.get('kernel_sha16') != bench.gl_kernel_sha16():
    raise SystemExit('{}: kernel_sha16 {} != {} of this tree'.format(src, rec.get('kernel_sha16'), bench.gl_kernel_sha16()))
rec['commit'] = subprocess.check_output(['git', '-C', ROOT, 'rev-parse', 'HEAD']).decode().strip()
if subprocess.check_output(['git', '-C', ROOT, 'status', '--porcelain', '--', 'single-speaker-tts_amd/csrc']).decode().strip():
    rec['commit'] += ' + uncommitted changes under csrc/'
json.dump(rec, open(dst, 'w'), indent=1)
print(dst, rec['commit'], rec['kernel_sha16'])
