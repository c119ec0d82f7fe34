"""Randomised check of the mixed-transformer paths: fused forward against generic, blocked inverse (block kernel / per-step
launches) against the pass-per-degree algorithm, over random sizes, member layouts, degree orders and batches.  (probe)"""
import random
import sys

import torch

from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.embeddings import PeriodicEmbedding
from tfep_amd.nn.flows import MAF
from tfep_amd.nn.transformers import (AffineTransformer, MixedTransformer, NeuralSplineTransformer,
                                      VolumePreservingShiftTransformer)

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(7)
worst = dict(fwd=0.0, fwd_l=0.0, inv=0.0, inv_l=0.0)
for case in range(n_cases):
    torch.manual_seed(case)
    D = rng.randint(12, 140)
    n_cond = rng.choice([0, 0, 3])
    perm = torch.randperm(D)
    cond, mapped = perm[:n_cond].sort().values, perm[n_cond:].sort().values
    n_members = rng.randint(2, 5)
    cuts = sorted(rng.sample(range(1, len(mapped)), n_members - 1))
    pos = torch.randperm(len(mapped))
    idx = [pos[a:b].sort().values for a, b in zip([0] + cuts, cuts + [len(mapped)])]
    members, periodic_cols = [], []
    for ind in idx:
        n = len(ind)
        kind = rng.choice(['plain', 'circular', 'ident', 'ident_up', 'ident_both', 'learn_lo', 'affine', 'shift'])
        K = rng.choice([4, 5, 8])
        lo, hi = torch.full((n,), -1.5), torch.full((n,), 2.0)
        if kind == 'plain':
            members.append(NeuralSplineTransformer(lo, hi, K))
        elif kind == 'circular':
            members.append(NeuralSplineTransformer(torch.zeros(n), torch.ones(n), K, circular=True))
            periodic_cols += mapped[ind].tolist()
        elif kind == 'ident':
            members.append(NeuralSplineTransformer(lo, hi, K, identity_boundary_slopes=True))
        elif kind == 'ident_up':
            members.append(NeuralSplineTransformer(lo, hi, K, identity_boundary_slopes=True, learn_upper_bound=True))
        elif kind == 'ident_both':
            members.append(NeuralSplineTransformer(lo, hi, K, identity_boundary_slopes=True, learn_lower_bound=True,
                                                   learn_upper_bound=True))
        elif kind == 'learn_lo':
            members.append(NeuralSplineTransformer(lo, hi, min(K, 5), learn_lower_bound=True))
        elif kind == 'affine':
            members.append(AffineTransformer())
        else:
            members.append(VolumePreservingShiftTransformer())
    emb = PeriodicEmbedding(D, limits=[0.0, 1.0], periodic_indices=sorted(periodic_cols)) if periodic_cols and rng.random() < 0.7 else None
    H = D + 2 * len(periodic_cols) + rng.choice([8, 40, 96])       # (a hidden layer must hold one unit per input degree)
    maf = MAF(generate_degrees(D, rng.choice(['ascending', 'descending', 'random']), conditioning_indices=cond.tolist()),
              transformer=MixedTransformer(members, idx), hidden_layers=[H, H + 8], embedding=emb, initialize_identity=False).cuda()
    B = rng.choice([1, 17, 64, 300, 700])
    x = torch.rand(B, D, device='cuda') * 1.2 - 0.1
    has_fused = maf._fused_kind() is not None
    with torch.no_grad():
        maf.fused = False
        yg, lg = maf(x)
        if has_fused:
            for split in (False, True):
                maf.fused, maf.split_gemm = True, split
                y, l = maf(x)
                worst['fwd'] = max(worst['fwd'], float((y - yg).abs().max()))
                worst['fwd_l'] = max(worst['fwd_l'], float((l - lg).abs().max()))
            maf.fused, maf.split_gemm = None, None
        maf.blocked_inverse = False
        xr, lr = maf.inverse(yg)
        maf.blocked_inverse = True
        if maf._blocked_ok():
            for rows in (None, 64):
                maf.inverse_rows_per_wave = rows
                xb, lb = maf.inverse(yg)
                worst['inv'] = max(worst['inv'], float((xb - xr).abs().max()))
                worst['inv_l'] = max(worst['inv_l'], float((lb - lr).abs().max()))
    print(case, D, B, [type(m).__name__[:6] + str(getattr(m, 'n_parameters_per_feature', '')) for m in members],
          'fused' if has_fused else 'generic',
          'block-kernel' if maf._blocked_ok() and maf._blocked_plan(x.device)['fused'] is not None else
          ('per-step' if maf._blocked_ok() else 'pass-per-degree'), {k: f'{v:.1e}' for k, v in worst.items()}, flush=True)
print('worst', worst)
assert worst['fwd'] < 5e-5 and worst['fwd_l'] < 1e-3 and worst['inv'] < 2e-4 and worst['inv_l'] < 2e-3
print('ok')
