"""Summarise a rocprofv3 kernel-trace csv of tools/probe/inv_trace.py: per-kernel totals of the LAST inverse and how
much GEMM time overlaps the block kernel."""
import csv
import glob
import sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
rows.sort()
last = max(i for i, r in enumerate(rows) if "weight_prepare_kernel" in r[2] and i < len(rows) - 1)
half = rows[last - 8:]                       # the last inverse (from its weight packing on)
t0, t1 = half[0][0], max(r[1] for r in half)
print('span ms', (t1 - t0) / 1e6, 'kernels', len(half))
tot = {}
for s, e, n, q in half:
    k = n.split('(')[0][:60]
    a = tot.setdefault(k, [0, 0.0, set()])
    a[0] += 1; a[1] += (e - s) / 1e6; a[2].add(q)
for k, (c, ms, q) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f'{ms:9.2f} ms {c:6d} calls  queues {sorted(q)}  {k}')
blk = [(s, e) for s, e, n, q in half if 'inverse_block' in n]
ov = 0.0
for s, e, n, q in half:
    if 'gemm' in n.lower():
        for bs, be in blk:
            if be > s and bs < e:
                ov += (min(e, be) - max(s, bs)) / 1e6
print('block kernel ms', sum(e - s for s, e in blk) / 1e6, ' GEMM time overlapping a block kernel ms', ov)
# a short window of the timeline
mid = blk[len(blk) // 2][0] - 300_000
for s, e, n, q in half:
    if mid <= s < mid + 2_000_000 and 'tfep' in n:
        print(f'{(s - mid) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  q{q}  {n.split("(")[0][:50]}')
