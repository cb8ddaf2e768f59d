#!/usr/bin/env python3
"""profiles/<tag>_gl_iter_valu.json from the text summary of tools/gl_pmc.sh (SQ passes of the Griffin-Lim kernel alone).

    python tools/gl_counters_json.py gpurun_out/r04_gl_pmc.txt profiles/r04_gl_iter_valu.json [iterations per launch = 3]

Derived fields (what bench.py reports as roofline_valu):
  valu_wave_insts_per_launch = SQ_INSTS_VALU
  shader_clock_mhz           = GRBM_GUI_ACTIVE / 8 XCDs / launch duration
  valu_busy_frac             = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
                               (SQ_ACTIVE_INST_VALU counts in units of four cycles, MI355X_MICROARCH.md)
  fp_share                   = (ADD + MUL + FMA + TRANS)_F32 / SQ_INSTS_VALU
  lds_bank_conflict_frac     = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
kernel_sha16 = content hash of csrc/griffin_lim.hip + griffin_lim.h the passes ran on (bench.py refuses a record whose
hash differs from the tree's); `commit` is stamped when the file is copied into profiles/ (tools/stamp_profile.py).
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    want = 'false, {}, false>'.format(per)
    cur = None
    vals = {}
    dur_us = None
    for line in open(src):
        if line.startswith('kernel:'):
            cur = line.strip()
            continue
        if cur is None or want not in cur:
            continue
        m = re.match(r'\s+duration under the counter passes: mean ([0-9.]+) us', line)
        if m:
            dur_us = float(m.group(1))
        m = re.match(r'\s+([A-Za-z_0-9]+)\s+mean\s+([0-9.]+)', line)
        if m:
            vals[m.group(1)] = float(m.group(2))
    if not vals or dur_us is None:
        raise SystemExit('no counters of a kernel matching "{}" in {}'.format(want, src))
    import bench
    cycles = vals['GRBM_GUI_ACTIVE'] / 8.0
    fp = sum(vals.get(k, 0.0) for k in ('SQ_INSTS_VALU_ADD_F32', 'SQ_INSTS_VALU_MUL_F32', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_TRANS_F32'))
    out = {
        'kernel': 'gl_stream_kernel<0, 1102, 275, false, {}, false>'.format(per),
        'iterations_per_launch': per,
        'workload': 'tools/gl_bench.py: B = 64, T = 1000, alone on 256 compute units',
        'launch_us_under_counters': dur_us,
        'valu_wave_insts_per_launch': vals['SQ_INSTS_VALU'],
        'valu_busy_frac': 4.0 * vals['SQ_ACTIVE_INST_VALU'] / (1024.0 * cycles),
        'fp_share': fp / vals['SQ_INSTS_VALU'],
        'shader_clock_mhz': cycles / dur_us,
        'lds_bank_conflict_frac': vals.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(1.0, vals.get('SQ_LDS_IDX_ACTIVE', 0.0)),
        'lds_wave_insts_per_launch': vals.get('SQ_INSTS_LDS'),
        'vmem_rd_wave_insts_per_launch': vals.get('SQ_INSTS_VMEM_RD'),
        'vmem_wr_wave_insts_per_launch': vals.get('SQ_INSTS_VMEM_WR'),
        'counters': vals,
        'kernel_sha16': bench.gl_kernel_sha16(),
        'commit': None,
        'note': 'separate rocprofv3 --pmc passes (tools/gl_pmc.sh), one counter group per run, kernel alone',
    }
    with open(dst, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in out if k != 'counters'}))


if __name__ == '__main__':
    main()
