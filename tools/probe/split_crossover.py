"""Split-f16 against exact-fp32 kernels over batch sizes for conditioners above 4 M weights (MADE.split_worthwhile takes the
split path for those at every batch size): where does the split path start to pay?  (probe)"""
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


for D, K in ((300, 8), (512, 8), (1000, 8), (3000, 8)):
    torch.manual_seed(0)
    with torch.device(dev):
        maf = MAF(generate_degrees(D, 'ascending'), transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), K),
                  initialize_identity=False)
    n_w = sum(l.mask.numel() for l in maf._conditioner._linears())
    for B in (256, 1024, 4096, 16384):
        x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
        maf.split_gemm = None
        res = dict(D=D, K=K, B=B, weights_M=round(n_w / 1e6, 1), default_is_split=bool(maf._use_split_gemm(B)))
        with torch.no_grad():
            for name, split in (('split_ms', True), ('fp32_ms', False)):
                maf.split_gemm, maf.fused = split, None
                res[name] = round(timeit(lambda: maf(x)), 3)
        print(json.dumps(res), flush=True)
    del maf
