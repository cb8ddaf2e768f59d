#!/usr/bin/env python3
"""Per-dispatch averages (and sums) of rocprofv3 --pmc counter CSVs for kernels whose name contains a pattern.

    python tools/pmc_summary.py <dir with p1/, p2/, ...> <kernel-name substring> [more substrings]

Prints one line per counter (mean over the dispatches of the kernel, and the sum over them) plus the kernel-trace
duration seen in the same passes, so that the numbers quoted in profiles/README.md can be re-derived from the
committed file."""
import collections
import csv
import glob
import sys


def main():
    root, pats = sys.argv[1], sys.argv[2:]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    meta = {}
    for f in sorted(glob.glob(root + '/**/*counter_collection.csv', recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            for p in pats:
                if p in name:
                    vals[name][r['Counter_Name']].append(float(r['Counter_Value']))
                    meta[name] = {k: r.get(k) for k in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size', 'Workgroup_Size', 'Grid_Size')}
    npass = max(1, len(glob.glob(root + '/**/*kernel_trace.csv', recursive=True)))
    for f in sorted(glob.glob(root + '/**/*kernel_trace.csv', recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            for p in pats:
                if p in name:
                    durs[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
    for name in sorted(vals):
        print('kernel:', name)
        print('  launch config:', meta.get(name))
        if durs[name]:
            d = sorted(durs[name])
            print('  duration under the counter passes: mean {:.1f} us, median {:.1f} us over {} dispatches; '
                  'sum per pass {:.1f} us'.format(sum(d) / len(d), d[len(d) // 2], len(d), sum(d) / npass))
        for c in sorted(vals[name]):
            v = vals[name][c]
            print('  {:32s} mean {:16.1f}   sum {:18.1f}   (n={})'.format(c, sum(v) / len(v), sum(v), len(v)))


if __name__ == '__main__':
    main()
