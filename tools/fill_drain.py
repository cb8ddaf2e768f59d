#!/usr/bin/env python3
"""Where the pipeline's fill and drain go in the driver's 20-step form: from a rocprofv3 kernel trace of
`bench.py --steps 20 --warmup 5 --no-cpu-baseline`, the start / end of every persistent decoder and of every
Griffin-Lim phase (first to last launch of a call), relative to the first timed decoder.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fd -o st -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline
    python3 tools/fill_drain.py gpurun_out/fd
"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
dec = [r for r in rows if 'dec_persistent' in r['Kernel_Name']]
fin = [r for r in rows if 'gl_stream_kernel<1' in r['Kernel_Name']]
seed = [r for r in rows if 'gl_stream_kernel<0' in r['Kernel_Name'] and '3, true' in r['Kernel_Name']]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dec, fin, seed = dec[-n:], fin[-n:], seed[-n:]
t0 = int(dec[0]['Start_Timestamp'])
ms = lambda r, k: (int(r[k]) - t0) / 1e6
print('call  decoder start..end      Griffin-Lim start..end   (ms after the first timed decoder started)')
for i in range(n):
    print('%3d   %8.2f %8.2f       %8.2f %8.2f' % (i, ms(dec[i], 'Start_Timestamp'), ms(dec[i], 'End_Timestamp'), ms(seed[i], 'Start_Timestamp'), ms(fin[i], 'End_Timestamp')))
print('total: first decoder start -> last Griffin-Lim end %.2f ms = %.2f ms per call' % (ms(fin[-1], 'End_Timestamp'), ms(fin[-1], 'End_Timestamp') / n))
