#!/usr/bin/env python
"""profiles/rNN_pmc_traffic.json (what bench.py reports as roofline.traffic) from the two counter passes of
tools/profile_roundN.sh:  pmc_traffic.py <fetch_pmc.csv> <write_pmc.csv> <out.json> <round>
HBM bytes per launch of the fused split kernel = 2 x FETCH_SIZE KB + WRITE_SIZE KB (gfx950: FETCH_SIZE counts half of the
16-byte-per-lane streaming reads, MI355X_MICROARCH.md, HBM section)."""
import csv
import json
import sys


def per_launch(path, counter):
    for r in csv.DictReader(open(path)):
        if 'split_gemm_kernel<25, 3, 25, 8' in r['kernel'] and r['counter'] == counter:
            return float(r['mean_per_launch']), r['kernel']
    raise SystemExit(f'{counter}: fused kernel not found in {path}')


fetch, kernel = per_launch(sys.argv[1], 'FETCH_SIZE')
write, _ = per_launch(sys.argv[2], 'WRITE_SIZE')
rnd = sys.argv[4]
json.dump({'kernel': kernel, 'FETCH_SIZE_KB_per_launch': fetch, 'WRITE_SIZE_KB_per_launch': write,
           'correction': 'gfx950: FETCH_SIZE counts half of 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM) -> x2',
           'hbm_bytes_per_launch': (2 * fetch + write) * 1024.0,
           'source': f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 1 --warmup 1 '
                     f'--no-cpu-baseline --no-extra-arms` (tools/profile_round{rnd}.sh), round {rnd}: '
                     f'profiles/r0{rnd}_bench_{{fetch,write}}_pmc.csv'}, open(sys.argv[3], 'w'), indent=1)
