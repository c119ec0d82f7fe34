"""One blocked inverse of a cfg2 layer (default) or the cfg4-i flow (INV_CFG=cfg4i) for rocprofv3 counter passes."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.embeddings import PeriodicEmbedding
from tfep_amd.nn.flows import MAF, SequentialFlow
from tfep_amd.nn.transformers import NeuralSplineTransformer

dev = torch.device('cuda:0')
torch.manual_seed(0)
B = int(os.environ.get('INV_BATCH', 16384))
if os.environ.get('INV_CFG', 'cfg2') == 'cfg4i':
    D = 512
    with torch.device(dev):
        flow = SequentialFlow(MAF(generate_degrees(D, 'ascending'),
                                  transformer=NeuralSplineTransformer(torch.zeros(D), torch.ones(D), 8, circular=True),
                                  embedding=PeriodicEmbedding(D, limits=[0.0, 1.0]), initialize_identity=False))
    y = torch.rand(B, D, device=dev)
else:
    D = 3000
    with torch.device(dev):
        flow = SequentialFlow(MAF(generate_degrees(D, 'ascending'),
                                  transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                                  initialize_identity=False))
    y = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
with torch.no_grad():
    flow.inverse(y)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    flow.inverse(y)
    torch.cuda.synchronize()
    print('inverse ms', (time.perf_counter() - t0) * 1e3)
