#!/usr/bin/env python
"""Per-dispatch values of one counter for kernels matching a substring, from a rocprofv3 --pmc csv directory."""
import csv
import glob
import sys

d, sub = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r['Kernel_Name']:
            rows.append((int(r['Dispatch_Id']), r['Counter_Name'], float(r['Counter_Value']), int(r['Grid_Size'])))
rows.sort()
for did, name, val, grid in rows[-int(sys.argv[3]) if len(sys.argv) > 3 else 0:]:
    print(did, name, f'{val:.0f}', 'grid', grid)
