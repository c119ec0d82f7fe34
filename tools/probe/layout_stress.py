"""Randomised check of the fused spline layouts on plain MAF layers: random bins / flags / width / hidden width / degree
order / batch (ragged tiles, single rows, conditioning DOFs), fused (both GEMM kernels) against generic and the blocked
inverse against pass-per-degree.  (probe)"""
import random
import sys

import torch

from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF
from tfep_amd.nn.transformers import NeuralSplineTransformer

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(11)
worst = dict(fwd=0.0, fwd_l=0.0, inv=0.0, inv_l=0.0)
n_fused = 0
for case in range(n_cases):
    torch.manual_seed(1000 + case)
    D = rng.choice([2, 3, 15, 16, 17, 31, 33, 64, 100, 129, 200])       # (one feature: no hidden degree, the reference fails too)
    K = rng.choice([4, 5, 8])
    circular = rng.random() < 0.25
    ident = rng.random() < 0.5
    ll, lu = (False, False) if circular else (rng.random() < 0.4, rng.random() < 0.4)
    n_cond = rng.choice([0, 0, 2]) if D > 4 else 0
    lo, hi = (0.0, 1.0) if circular else (-2.0, 1.5)
    n_map = D - n_cond
    tr = NeuralSplineTransformer(torch.full((n_map,), lo), torch.full((n_map,), hi), K, circular=circular,
                                 identity_boundary_slopes=ident, learn_lower_bound=ll, learn_upper_bound=lu)
    cond = sorted(rng.sample(range(D), n_cond))
    maf = MAF(generate_degrees(D, rng.choice(['ascending', 'descending', 'random']), conditioning_indices=cond), transformer=tr,
              hidden_layers=[D + rng.choice([3, 50]), D + 20], initialize_identity=False).cuda()
    has_fused = maf._fused_kind() is not None
    n_fused += has_fused
    B = rng.choice([1, 2, 63, 128, 129, 257, 1000])
    x = (torch.rand(B, D, device='cuda') - 0.5) * (hi - lo) * (1.0 if circular else 1.5) + 0.5 * (hi + lo)
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
        for rows in (None, 64):
            maf.inverse_rows_per_wave = rows
            xb, lb = maf.inverse(yg)
            worst['inv'] = max(worst['inv'], float((xb - xr).abs().max()))
            worst['inv_l'] = max(worst['inv_l'], float((lb - lr).abs().max()))
    print(case, f'D={D} K={K} P={tr.n_parameters_per_feature} circ={int(circular)} ident={int(ident)} ll={int(ll)} lu={int(lu)} B={B}',
          'fused' if has_fused else 'generic', {k: f'{v:.1e}' for k, v in worst.items()}, flush=True)
print('worst', worst, 'fused cases', n_fused)
assert worst['fwd'] < 5e-5 and worst['fwd_l'] < 1e-3 and worst['inv'] < 5e-4 and worst['inv_l'] < 5e-3
print('ok')
