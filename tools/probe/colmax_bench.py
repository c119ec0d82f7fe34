"""Bandwidth of tfep_column_absmax against torch's reduction (TFEP_COLMAX_WGS: workgroups of the launch)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd import ops
for (R, C) in [(131072, 1024), (65536, 3000), (16384, 3000)]:
    x = torch.randn(R, C, device='cuda')
    res = {}
    for wgs in (256, 512, 1024, 2048, 4096, 8192):
        os.environ['TFEP_COLMAX_WGS'] = str(wgs)
        ops.column_absmax(x); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ops.column_absmax(x)
        torch.cuda.synchronize()
        res[wgs] = round((time.perf_counter() - t0) / 10 * 1e6, 1)
    torch.linalg.vector_norm(x, ord=float('inf'), dim=0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        torch.linalg.vector_norm(x, ord=float('inf'), dim=0)
    torch.cuda.synchronize()
    res['torch'] = round((time.perf_counter() - t0) / 10 * 1e6, 1)
    print(json.dumps({'rows': R, 'cols': C, 'MB': R * C * 4 // 2**20, 'us': res}), flush=True)
