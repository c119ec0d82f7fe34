"""A/B of the hidden-layer kernel (ELU, split rows out) on a 256- vs a 384-column tile, dense product (TFEP_SPLIT_ELU_WIDE)."""
import os, sys, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd import ops
dev = 'cuda'
torch.manual_seed(0)
for (B, N, K) in [(65536, 15360, 15008), (65536, 15360, 3008), (16384, 15360, 15008)]:
    x = torch.randn(B, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.01
    xs, xi = ops.split_rows(x, K)
    ws, wi = ops.split_rows(w, K, per_tensor=True)
    wi4 = torch.zeros(4, device=dev); wi4[:2] = wi[:2]
    wi4[2] = float(w.abs().sum(dim=1).max())
    bias = torch.zeros(N, device=dev)
    del x
    res = {}
    outs = {}
    for wide in (0, 1, 0, 1):
        os.environ['TFEP_SPLIT_ELU_WIDE'] = str(wide)
        o, oi = ops.masked_linear_split(xs, xi, ws, wi4, bias, N, act=1, split_out=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            o, oi = ops.masked_linear_split(xs, xi, ws, wi4, bias, N, act=1, split_out=True)
        torch.cuda.synchronize()
        res.setdefault(wide, []).append(round((time.perf_counter() - t0) / 3 * 1e3, 3))
        outs[wide] = (o[:64].clone(), oi[:64].clone())
        del o, oi
    same = torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32)) and torch.equal(outs[0][1], outs[1][1])
    tf = 2.0 * B * N * K / 1e12
    print(json.dumps({'B': B, 'N': N, 'K': K, 'ms_256': res[0], 'ms_384': res[1], 'TF_256': round(tf / min(res[0]) * 1e3, 1),
                      'TF_384': round(tf / min(res[1]) * 1e3, 1), 'same_bits': same}), flush=True)
    del xs, ws, w
