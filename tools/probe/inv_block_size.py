"""Degrees per block of the blocked inverse (AutoregressiveFlow.inverse_block) against the batch size: one cfg2 layer."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF
from tfep_amd.nn.transformers import NeuralSplineTransformer
D = 3000
dev = torch.device('cuda:0')
torch.manual_seed(0)
with torch.device(dev):
    maf = MAF(generate_degrees(D, 'ascending'), transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
              initialize_identity=False)
for B in [int(b) for b in os.environ.get('BATCHES', '8192,32768,65536').split(',')]:
    x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
    with torch.no_grad():
        y, _ = maf(x)
        for G in (16, 24, 32, 48):
            maf.inverse_block = G
            maf.inverse(y); torch.cuda.synchronize()
            t0 = time.perf_counter()
            xi, _ = maf.inverse(y); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            bp = maf._blocked_plan(dev)
            print(f'B={B} inverse_block={G}: {dt * 1e3:.1f} ms, blocks {len(bp["blocks"])}, fused {bp["fused"] is not None}, '
                  f'round trip {float((xi - x).abs().max()):.1e}', flush=True)
    del x, y, xi
