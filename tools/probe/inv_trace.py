"""One cfg2-layer blocked inverse (batch 8192) for a rocprofv3 --kernel-trace timeline (look-ahead overlap check)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF, SequentialFlow
from tfep_amd.nn.transformers import NeuralSplineTransformer

D, B = 3000, int(os.environ.get('INV_BATCH', 8192))
dev = torch.device('cuda:0')
torch.manual_seed(0)
with torch.device(dev):
    flow = SequentialFlow(MAF(generate_degrees(D, 'ascending'),
                              transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                              initialize_identity=False))
y = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
with torch.no_grad():
    flow.inverse(y)
    torch.cuda.synchronize()
    flow.inverse(y)
    torch.cuda.synchronize()
