"""Wide (256-column) against narrow (32-column) tile of the exact-fp32 masked-linear GEMM on mid-sized dense products:
where is the crossover in workgroups?  (probe for ops.few_wide_tiles)"""
import json
import time

import torch

from tfep_amd import ops
from tfep_amd._lib import call, ptr, stream_of

dev = torch.device('cuda', 0)
tm, tn, tk = ops.tile_sizes()
nn = ops.narrow_tile_n()


def run(x, w, b, N, tile_n):
    out = torch.empty(x.shape[0], N, dtype=torch.float32, device=dev)
    call('tfep_masked_linear_forward', ptr(x), x.shape[1], ptr(w), w.shape[1], ptr(b), None, None, None, ptr(out), out.shape[1],
         x.shape[0], N, w.shape[0], w.shape[1], 1, tile_n, stream_of(x))
    return out


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for B, K, N in ((1024, 256, 800), (1024, 800, 800), (1024, 800, 3200), (4096, 800, 800), (4096, 800, 3200), (1024, 2048, 2048),
                (8192, 800, 800), (16384, 800, 3200), (2048, 1500, 1500)):
    Kp, Np = ops.round_up(K, tk), ops.round_up(N, tn)
    x = torch.randn(B, Kp, device=dev)
    w = torch.randn(Np, Kp, device=dev) / K ** 0.5
    b = torch.zeros(Np, device=dev)
    tw = timeit(lambda: run(x, w, b, N, 0))
    tnr = timeit(lambda: run(x, w, b, N, nn))
    wide_wgs = ((B + tm - 1) // tm) * ((N + tn - 1) // tn)
    print(json.dumps(dict(B=B, K=K, N=N, wide_tiles=wide_wgs, wide_us=round(tw, 1), narrow_us=round(tnr, 1))), flush=True)
