import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF, SequentialFlow
from tfep_amd.nn.transformers import NeuralSplineTransformer
from tfep_amd.graphs import GraphedFlow
dev = torch.device('cuda'); torch.manual_seed(0)
D, B = 3000, int(os.environ.get('B', 8192))
with torch.device(dev):
    flow = SequentialFlow(MAF(generate_degrees(D, 'ascending'), transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8), initialize_identity=False))
x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
with torch.no_grad():
    y, _ = flow(x)
    ref, lref = flow.inverse(y)
    ref = ref.clone()
    bad = 0
    for it in range(int(os.environ.get('N', 25))):
        xi, li = flow.inverse(y)
        d = (xi - ref).abs()
        if float(d.max()) > 0 or not torch.equal(li, lref):
            bad += 1
            rows = torch.nonzero((d > 0).any(1)).flatten(); cols = torch.nonzero((d > 0).any(0)).flatten()
            print('eager iter', it, 'max', float(d.max()), 'rows', len(rows), rows[:6].tolist(), 'cols', len(cols), cols[:6].tolist(), flush=True)
    gi = GraphedFlow(flow, B, D, inverse=True, warmup=1)
    for it in range(int(os.environ.get('N', 25))):
        xg, lg = gi(y)
        d = (xg - ref).abs()
        if float(d.max()) > 0:
            bad += 1
            rows = torch.nonzero((d > 0).any(1)).flatten(); cols = torch.nonzero((d > 0).any(0)).flatten()
            print('graph iter', it, 'max', float(d.max()), 'rows', len(rows), rows[:6].tolist(), 'cols', len(cols), cols[:6].tolist(), flush=True)
print('mismatches', bad)
