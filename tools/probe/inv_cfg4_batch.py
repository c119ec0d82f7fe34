"""cfg4-i inverse (4 layers x 512 torsions, circular RQ-8 + periodic embedding) against the batch size: is one pass over 16 384
rows cheaper than two over 8192 (which get the paired 16-row block kernel)?"""
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
for B in (4096, 8192, 16384):
    y = (torch.rand(B, D, device=dev) * 2 - 1) * 3.1
    with torch.no_grad():
        flow.inverse(y); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            x, _ = flow.inverse(y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        yy, _ = flow(x)
    print(json.dumps({'B': B, 'ms': round(dt * 1e3, 2), 'us_per_sample': round(dt * 1e6 / B, 3),
                      'roundtrip_circle_max': float(torch.remainder(yy - y + math.pi, 2 * math.pi).sub(math.pi).abs().max())}), flush=True)
