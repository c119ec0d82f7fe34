"""Blocked inverse of one cfg2 layer (B = 8192) against the number of degrees per block (layer.inverse_block)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF
from tfep_amd.nn.transformers import NeuralSplineTransformer
D, B = 3000, int(os.environ.get('INV_B', 8192))
dev = 'cuda'
torch.manual_seed(0)
with torch.device(dev):
    layer = MAF(generate_degrees(D, 'ascending'), transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                initialize_identity=False)
x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
with torch.no_grad():
    y, _ = layer(x)
    for G in [int(v) for v in os.environ.get('INV_BLOCKS', '16 32 24 48 64').split()]:
        layer.inverse_block = G
        try:
            xi, _ = layer.inverse(y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                xi, _ = layer.inverse(y)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 2
            bp = layer._blocked_plan(x.device)
            print(json.dumps({'inverse_block': G, 'ms': round(dt * 1e3, 2), 'fused': bp['fused'] is not None, 'blocks': len(bp['blocks']),
                              'paired': getattr(layer, 'last_inverse_paired', None),
                              'roundtrip_max_abs': float((xi - x).abs().max())}), flush=True)
        except Exception as e:
            print(json.dumps({'inverse_block': G, 'error': f'{type(e).__name__}: {e}'[:300]}), flush=True)
