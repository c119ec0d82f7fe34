#!/usr/bin/env python
"""Per-launch durations of the kernels of the LAST inverse in a rocprofv3 kernel trace (csv), in launch order."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                         int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 0)) or 0), int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)))
rows.sort()
# the last 24 super-block kernels delimit the last inverse
idx = [i for i, r in enumerate(rows) if 'inverse_superblock_kernel' in r[2]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
first = idx[-n]
# walk back to the GEMMs that precede the first super-block kernel
start = first
while start > 0 and ('gemm_kernel' in rows[start - 1][2] or 'split_columns' in rows[start - 1][2]):
    start -= 1
t_prev = rows[start][0]
for s, e, name, wg, grid in rows[start:idx[-1] + 1]:
    short = name.replace('void tfep::', '').split('(')[0]
    print(f'{(s - t_prev) / 1e3:8.1f} gap  {(e - s) / 1e3:8.1f} us  grid {grid // max(wg, 1):5d} x {wg:4d}  {short}')
    t_prev = e
print('span of the last inverse (first GEMM to last chain kernel): %.2f ms' % ((rows[idx[-1]][1] - rows[start][0]) / 1e6))
