import sys, torch
sys.path.insert(0, '.')
from tfep_amd.nn.dynamics import EGNNDynamics
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[3]) if len(sys.argv) > 3 else 4
gen = torch.Generator(device='cuda').manual_seed(7)
x = torch.randn(B, 3 * n, device='cuda', generator=gen) * 0.6
torch.manual_seed(0)
dyn = EGNNDynamics(node_types=[i % 4 for i in range(n)], r_cutoff=40.0, n_layers=L, initialize_identity=False).cuda()
eps = torch.randn(B, 3 * n, device='cuda', generator=gen)
dyn.split_gemm = True
with torch.no_grad():
    outs = [dyn.jvp(0.3, x, eps)[1] for _ in range(4)]
ref = outs[0]
for o in outs[1:]:
    d = (o - ref).abs()
    bad = d > 0
    print(f'B={B} n={n} L={L}: differing elements {int(bad.sum())} of {bad.numel()}, samples {int(bad.any(1).sum())}, '
          f'max {float(d.max()):.3e}; differing atoms of sample with most: {sorted(set((bad[bad.sum(1).argmax()].nonzero().flatten() // 3).tolist()))[:20]}')
