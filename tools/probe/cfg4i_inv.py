#!/usr/bin/env python
"""cfg4-i blocked inverse (4-layer MAF + circular RQ-8 + periodic embedding, 512 torsions) at several batch sizes."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.nn.conditioners import generate_degrees  # noqa: E402
from tfep_amd.nn.embeddings import PeriodicEmbedding  # noqa: E402
from tfep_amd.nn.flows import MAF, SequentialFlow  # noqa: E402
from tfep_amd.nn.transformers import NeuralSplineTransformer  # noqa: E402

dev = torch.device('cuda')
D = 512
torch.manual_seed(0)
with torch.device(dev):
    flow = SequentialFlow(*[MAF(generate_degrees(D, 'ascending' if i % 2 == 0 else 'descending'),
                                transformer=NeuralSplineTransformer(torch.zeros(D), torch.ones(D), 8, circular=True),
                                embedding=PeriodicEmbedding(D, limits=[0.0, 1.0]), initialize_identity=False) for i in range(4)])
x = torch.rand(16384, D, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
with torch.no_grad():
    y, _ = flow(x)
    for B in [int(b) for b in os.environ.get('BS', '4096,8192,16384').split(',')]:
        flow.inverse(y[:B])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            xi, _ = flow.inverse(y[:B])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        d = (xi - x[:B]).abs()
        print(json.dumps(dict(B=B, ms=round(1e3 * dt, 2), us_per_row=round(1e6 * dt / B, 3), schedule=[l.last_inverse_schedule for l in flow],
                              roundtrip=float(torch.minimum(d, 1 - d).max()))))
from tfep_amd import _lib
lib = _lib.load()
for i, layer in enumerate(flow):
    bp = layer._blocked_plan(dev, batch=8192)
    f = bp['fused']
    print(i, 'blocks', len(bp['blocks']), 'L', bp['L'], f, 'paired lds', lib.tfep_inverse_block_lds_bytes_paired(bp['L'], f['cache_len'], f['max_feats']),
          'supers', None if not bp.get('supers') else len(bp['supers']))
