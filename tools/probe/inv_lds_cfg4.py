"""LDS footprint of the block kernel's layouts for a cfg4-i layer (512 torsions, circular RQ-8, periodic embedding)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd import _lib
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.embeddings import PeriodicEmbedding
from tfep_amd.nn.flows import MAF
from tfep_amd.nn.transformers import NeuralSplineTransformer
import math
D = 512
dev = 'cuda'
with torch.device(dev):
    emb = PeriodicEmbedding(D, limits=[-math.pi, math.pi], periodic_indices=list(range(D)))
    layer = MAF(generate_degrees(D, 'ascending'), embedding=emb, initialize_identity=False,
                transformer=NeuralSplineTransformer(torch.full((D,), -math.pi), torch.full((D,), math.pi), 8, circular=True))
lib = _lib.load()
for G in (16, 8):
    layer.inverse_block = G
    bp = layer._blocked_plan(torch.device(dev, 0))
    f = bp['fused']
    L = bp['L']
    print(json.dumps({'inverse_block': G, 'fused': f, 'L': L,
                      'lds64': lib.tfep_inverse_block_lds_bytes(L, f['cache_len'], f['max_feats']) if f else None,
                      'lds16': lib.tfep_inverse_block_lds_bytes_rows(L, f['cache_len'], f['max_feats'], 16) if f else None,
                      'lds_paired': lib.tfep_inverse_block_lds_bytes_paired(L, f['cache_len'], f['max_feats']) if f else None,
                      'k_pad': list(layer._conditioner.plan(torch.device(dev, 0))['k_pad'])}))
