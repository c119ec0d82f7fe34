#!/usr/bin/env python
"""cfg4-ii forward (4-layer MAF + Moebius(2, unit sphere), 1024 features, B = 131 072): the layer kernel
(tfep_maf_layer_forward_split) against the launch-by-launch split path, same box, alternating."""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.nn.conditioners import generate_degrees  # noqa: E402
from tfep_amd.nn.flows import MAF, SequentialFlow  # noqa: E402
from tfep_amd.nn.transformers import MoebiusTransformer  # noqa: E402

dev = torch.device('cuda')
D, B = 512, int(os.environ.get('B', 131072))
torch.manual_seed(0)
with torch.device(dev):
    flow = SequentialFlow(*[MAF(generate_degrees(2 * D, 'ascending' if i % 2 == 0 else 'descending', repeats=2),
                                transformer=MoebiusTransformer(dimension=2, unit_sphere=True), initialize_identity=False)
                            for i in range(4)])
ang = torch.rand(B, D, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 2 * math.pi
x = torch.stack([torch.cos(ang), torch.sin(ang)], dim=2).reshape(B, 2 * D)


def clock(n=5):
    with torch.no_grad():
        flow(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            y, l = flow(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, y, l


res = {}
for rep in range(int(os.environ.get('REPS', 2))):
    for name, on, cache in (('launch_by_launch', False, False), ('layer_kernel', True, False), ('layer_kernel_cached_packs', True, True)):
        for layer in flow:
            layer.layer_kernel = bool(on)
            layer._conditioner.cache_packed_weights = cache
        dt, y, l = clock()
        res.setdefault(name, []).append(round(1e3 * dt, 3))
        if name == 'launch_by_launch':
            y0, l0 = y, l
        else:
            res[name + '_max_dy'] = float((y - y0).abs().max())
            res[name + '_max_dl'] = float((l - l0).abs().max())
alg = B * 4 * (2 * 4 * 2 * D + 4.0)
best = min(res['layer_kernel'])
nnz = sum(float(torch.count_nonzero(lin.mask)) for layer in flow for lin in layer._conditioner.layers[::2])
res['hbm_frac_algorithmic'] = alg / (best * 1e-3) / 8e12
res['mfma_frac'] = 2 * nnz * B / (best * 1e-3) / 1e12 / (2516.6 / 3)
if os.environ.get('TFEP_DIAG_MAF_LAYER'):
    import ctypes
    from tfep_amd import _lib
    buf = (ctypes.c_ulonglong * 5)()
    _lib.call('tfep_diag_maf_layer_cycles', buf)              # clear
    for layer in flow:
        layer.layer_kernel = True
    with torch.no_grad():
        flow(x)
    torch.cuda.synchronize()
    _lib.call('tfep_diag_maf_layer_cycles', buf)
    n = max(1, buf[4])
    res['cycles_per_workgroup'] = dict(k_loops=buf[0] / n, hidden_epilogues=buf[1] / n, moebius_epilogues=buf[2] / n, kernel=buf[3] / n,
                                       workgroups=int(buf[4]))
print(json.dumps(res))
