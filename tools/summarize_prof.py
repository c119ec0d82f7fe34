#!/usr/bin/env python
"""Condense rocprofv3 CSV output into the small summaries committed under profiles/.

usage: summarize_prof.py <rocprof -d dir> <out prefix>
  * <prefix>_kernel_stats.csv   : the --stats per-kernel table (top 25 rows by total time)
  * <prefix>_pmc.csv            : per kernel name: launches, and per counter the per-launch mean / total
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    return name if len(name) < 160 else name[:157] + '...'


def main(src, prefix):
    os.makedirs(os.path.dirname(prefix) or '.', exist_ok=True)
    for f in glob.glob(os.path.join(src, '**', '*_kernel_stats.csv'), recursive=True):
        rows = list(csv.reader(open(f)))
        with open(prefix + '_kernel_stats.csv', 'w', newline='') as out:
            w = csv.writer(out)
            for r in rows[:26]:
                w.writerow([short(c) for c in r])
    for f in glob.glob(os.path.join(src, '**', '*_counter_collection.csv'), recursive=True):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        disp = defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            agg[k][r['Counter_Name']][0] += 1
            agg[k][r['Counter_Name']][1] += float(r['Counter_Value'])
            disp[k].add(r['Dispatch_Id'])
        with open(prefix + '_pmc.csv', 'w', newline='') as out:
            w = csv.writer(out)
            w.writerow(['kernel', 'launches', 'counter', 'mean_per_launch', 'total'])
            items = sorted(agg.items(), key=lambda kv: -max(v[1] for v in kv[1].values()))
            for k, counters in items[:25]:
                for c, (n, tot) in counters.items():
                    nl = len(disp[k])
                    w.writerow([short(k), nl, c, tot / max(nl, 1), tot])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
