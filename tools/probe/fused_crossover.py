"""Exact-fp32 path: fused output-GEMM + spline kernel against the generic path over layer and batch sizes -- where does
fusing start to pay?  (the threshold of AutoregressiveFlow._fused_pays)"""
import json
import time

import torch

from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF
from tfep_amd.nn.transformers import NeuralSplineTransformer

dev = torch.device('cuda', 0)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for D, K in ((128, 8), (512, 8), (512, 5), (1000, 8)):
    torch.manual_seed(0)
    maf = MAF(generate_degrees(D, 'ascending'), transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), K),
              initialize_identity=False).to(dev)
    maf.split_gemm = False
    for B in (1024, 4096, 16384, 65536):
        x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
        with torch.no_grad():
            maf.fused = True
            tf = timeit(lambda: maf(x))
            maf.fused = False
            tg = timeit(lambda: maf(x))
        wgs = ((B + 127) // 128) * ((D + 15) // 16)
        print(json.dumps(dict(D=D, K=K, B=B, fused_workgroups=wgs, fused_ms=round(tf, 3), generic_ms=round(tg, 3))), flush=True)
