#!/usr/bin/env python3
"""Experiment: remove the `s_nop 0` that hipcc's hazard recogniser puts between a packed-f32 VALU instruction (or an asm
statement) and a directly following instruction that reads its result, inside the gl_stream_kernel functions of a device
assembly file.  (The recogniser treats op_sel_hi[0] of a VOP3P instruction as the VOP3 destination select and assumes the
destination-select forwarding hazard of gfx940 for it, and for every asm statement; v_pk_*_f32 write whole registers.)

    python tools/strip_nops.py in.s out.s
"""
import re
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = open(src).read().split('\n')
    out = []
    in_kernel = False
    removed = 0
    prev_real = ''
    for i, ln in enumerate(lines):
        m = re.match(r'^(_ZN3tts16gl_stream_kernel\w+):', ln)
        if m:
            in_kernel = True
        if in_kernel and re.match(r'^\s*s_endpgm', ln):
            in_kernel = False
        st = ln.strip()
        if in_kernel and st == 's_nop 0':
            # next real instruction
            j = i + 1
            while j < len(lines) and (not lines[j].strip() or lines[j].strip().startswith(';') or lines[j].strip().startswith('.')):
                j += 1
            nxt = lines[j].strip() if j < len(lines) else ''
            if prev_real.startswith('v_pk_') and nxt.startswith('v_') and 'permlane' not in nxt and 'readlane' not in nxt and 'readfirstlane' not in nxt:
                removed += 1
                continue
        if st and not st.startswith(';') and not st.startswith('.'):
            prev_real = st
        out.append(ln)
    open(dst, 'w').write('\n'.join(out))
    print('removed', removed, 's_nop 0')


if __name__ == '__main__':
    main()
