#!/usr/bin/env python
"""Run-length-compressed kernel timeline from a rocprofv3 --kernel-trace CSV: which helper kernels and copies sit between
the GEMMs of one step.

usage: timeline_prof.py <rocprof -d dir> <out.txt> [last_n_events]
Each output line: <count> x <short kernel name>  total <us>  (consecutive launches of the same kernel are merged).
"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)
    name = name.replace('void ', '').replace('tfep::', '')
    name = re.sub(r'at::native::(\(anonymous namespace\)::)?', '', name)
    return name[:90]


def main(src, out, last_n=3000):
    files = glob.glob(os.path.join(src, '**', '*_kernel_trace.csv'), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    rows = rows[-last_n:]
    with open(out, 'w') as fh:
        i = 0
        t0 = rows[0][0] if rows else 0
        while i < len(rows):
            j = i
            tot = 0
            while j < len(rows) and rows[j][2] == rows[i][2]:
                tot += rows[j][1] - rows[j][0]
                j += 1
            fh.write('%9.3f ms  %5d x %-90s total %10.1f us\n' % ((rows[i][0] - t0) / 1e6, j - i, short(rows[i][2]), tot / 1e3))
            i = j


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3000)
