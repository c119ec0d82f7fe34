"""cfg4-i / cfg4-ii forward passes for a rocprofv3 --kernel-trace --stats breakdown."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.embeddings import PeriodicEmbedding
from tfep_amd.nn.flows import MAF, SequentialFlow
from tfep_amd.nn.transformers import MoebiusTransformer, NeuralSplineTransformer
dev = torch.device('cuda:0')
D, B = 512, 131072
which = os.environ.get('CFG4', 'i')
order = lambda i: 'ascending' if i % 2 == 0 else 'descending'
with torch.device(dev):
    if which == 'i':
        flow = SequentialFlow(*[MAF(generate_degrees(D, order(i)), transformer=NeuralSplineTransformer(torch.zeros(D), torch.ones(D), 8, circular=True),
                                    embedding=PeriodicEmbedding(D, limits=[0.0, 1.0]), initialize_identity=False) for i in range(4)])
        x = torch.rand(B, D, device=dev)
    else:
        flow = SequentialFlow(*[MAF(generate_degrees(2 * D, order(i), repeats=2), transformer=MoebiusTransformer(dimension=2, unit_sphere=True),
                                    initialize_identity=False) for i in range(4)])
        th = torch.rand(B, D, device=dev) * 6.283
        x = torch.stack([torch.cos(th), torch.sin(th)], dim=-1).reshape(B, 2 * D)
with torch.no_grad():
    flow(x); torch.cuda.synchronize()
    for _ in range(3):
        flow(x)
    torch.cuda.synchronize()
