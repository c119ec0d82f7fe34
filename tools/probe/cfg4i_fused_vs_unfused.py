"""cfg4-i forward (4 layers, 512 torsions, circular RQ-8, K = 2558 hidden units): fused output GEMM + spline epilogue (one wave per
SIMD evaluates the splines) against the un-fused kernels (GEMM writes the parameters, the spline kernel runs at full occupancy)."""
import os, sys, time, json, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.embeddings import PeriodicEmbedding
from tfep_amd.nn.flows import MAF, SequentialFlow
from tfep_amd.nn.transformers import NeuralSplineTransformer
D = 512
dev = 'cuda'
torch.manual_seed(0)
with torch.device(dev):
    flow = SequentialFlow(*[MAF(generate_degrees(D, 'ascending' if i % 2 == 0 else 'descending'),
                                embedding=PeriodicEmbedding(D, limits=[-math.pi, math.pi], periodic_indices=list(range(D))),
                                transformer=NeuralSplineTransformer(torch.full((D,), -math.pi), torch.full((D,), math.pi), 8, circular=True),
                                hidden_layers=2, initialize_identity=False) for i in range(4)])
for B in (131072, 32768):
    x = (torch.rand(B, D, device=dev) * 2 - 1) * 3.1
    res = {'B': B}
    outs = {}
    with torch.no_grad():
        for fused in (True, False, True, False):
            for l in flow:
                l.fused = fused
            outs[fused] = flow(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                flow(x)
            torch.cuda.synchronize()
            res.setdefault('fused_ms' if fused else 'unfused_ms', []).append(round((time.perf_counter() - t0) / 3 * 1e3, 2))
    res['max_abs_dldj'] = float((outs[True][1] - outs[False][1]).abs().max())
    print(json.dumps(res), flush=True)
