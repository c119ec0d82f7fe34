#!/usr/bin/env python
"""Blocked inverse of one cfg2 layer (D = 3000, RQ-8, default width): the one-launch-per-super-block kernel against the
block-by-block launches, same box, alternating.  B from the environment (default 8192)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

dev = torch.device('cuda')
D, B = int(os.environ.get('D', 3000)), int(os.environ.get('B', 8192))
flow = bench.build_flow(D, 1, 8, dev)
layer = flow[0]
x = torch.randn(B, D, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)).clamp_(-4.9, 4.9)
res = {'B': B, 'D': D}
with torch.no_grad():
    y, l = layer(x)

    def clock(n=3):
        layer.inverse(y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = layer.inverse(y)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, out

    ref = None
    for rep in range(int(os.environ.get('REPS', 2))):
        for name, on in (('block_by_block', False), ('super_kernel', True)):
            layer.inverse_super_kernel = on
            dt, (xi, li) = clock()
            res.setdefault(name, []).append(round(1e3 * dt, 2))
            if not on:
                ref = (xi, li)
            else:
                res['max_dx_vs_block_by_block'] = float((xi - ref[0]).abs().max())
                res['max_dl_vs_block_by_block'] = float((li - ref[1]).abs().max())
                res['roundtrip_max_abs'] = float((xi - x).abs().max())
                res['ldj_cancel_max_abs'] = float((li + l).abs().max())
    layer._conditioner.cache_packed_weights = True
    dt, _ = clock()
    res['super_kernel_cached_packs'] = round(1e3 * dt, 2)
print(json.dumps(res))
