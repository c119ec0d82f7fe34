"""A/B of the split GEMM's plain-linear tile width on a dense product (TFEP_SPLIT_WIDE_TILE=0/1 read per launch)."""
import os, sys, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd import ops

dev = 'cuda'
torch.manual_seed(0)
for (B, N, K) in [(16384, 76800, 9024), (16384, 9600, 9024), (76800, 9600, 16384), (65536, 76800, 9024)]:
    if B * N * 4 > 24e9:
        continue
    x = torch.randn(B, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.01
    xs, xi = ops.split_rows(x, K)
    ws, wi = ops.split_rows(w, K, per_tensor=True)
    wi4 = torch.zeros(4, device=dev); wi4[:2] = wi[:2]
    del x
    out = torch.empty(B, N, device=dev)
    res = {}
    for wide in (0, 1, 0, 1):
        os.environ['TFEP_SPLIT_WIDE_TILE'] = str(wide)
        ops.masked_linear_split(xs, xi, ws, wi4, None, N, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ops.masked_linear_split(xs, xi, ws, wi4, None, N, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        res.setdefault(wide, []).append(round(dt * 1e3, 3))
        if wide == 0:
            ref = out[:256, :1024].clone()
        else:
            res['max_abs_diff'] = float((out[:256, :1024] - ref).abs().max())
    tf = 2.0 * B * N * K / 1e12
    print(json.dumps({'B': B, 'N': N, 'K': K, 'ms_256': res[0], 'ms_400': res[1], 'TF_256': round(tf / min(res[0]) * 1e3, 1),
                      'TF_400': round(tf / min(res[1]) * 1e3, 1), 'max_abs_diff': res['max_abs_diff']}), flush=True)
    del xs, ws, out, w
