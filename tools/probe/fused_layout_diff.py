"""Where do the fused and the un-fused spline paths of one layout differ?  (probe)"""
import sys
import torch
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF
from tfep_amd.nn.transformers import NeuralSplineTransformer

K, circ, ident, ll, lu = [int(a) for a in sys.argv[1:6]]
torch.manual_seed(K)
D, B = 37, 301
lo, hi = (0.0, 2.0) if circ else (-3.0, 3.0)
tr = NeuralSplineTransformer(torch.full((D,), lo), torch.full((D,), hi), K, circular=bool(circ), identity_boundary_slopes=bool(ident),
                             learn_lower_bound=bool(ll), learn_upper_bound=bool(lu))
maf = MAF(generate_degrees(D, 'ascending'), transformer=tr, hidden_layers=[90, 110], initialize_identity=False).cuda()
x = (torch.rand(B, D, device='cuda') - 0.5) * (hi - lo) * (1.0 if circ else 1.4) + 0.5 * (hi + lo)
with torch.no_grad():
    for split in (False, True):
        maf.split_gemm = split
        maf.fused = True
        y, l = maf(x)
        maf.fused = False
        yg, lg = maf(x)
        d = (y - yg).abs()
        i = int(d.argmax())
        r, c = divmod(i, D)
        print(f'split={split}: max dy {float(d.max()):.3e} at row {r} feature {c} x={float(x[r, c]):.4f} fused={float(y[r, c]):.5f} '
              f'generic={float(yg[r, c]):.5f}; bad entries {int((d > 1e-4).sum())} of {d.numel()}; max dl {float((l - lg).abs().max()):.3e}')
        bad = (d > 1e-4).nonzero()
        print('  bad features:', sorted(set(bad[:, 1].tolist()))[:40], 'bad rows (first):', sorted(set(bad[:, 0].tolist()))[:10])
        params = maf._conditioner(x)
        P = tr.n_parameters_per_feature
        pr = params.view(B, P, D)
        print('  last (log scale) range', float(pr[:, P - 1].min()), float(pr[:, P - 1].max()), 'last2', float(pr[:, P - 2].min()),
              float(pr[:, P - 2].max()))
