import sys, torch
sys.path.insert(0, '.')
from tfep_amd.nn.dynamics import EGNNDynamics
B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 256
gen = torch.Generator(device='cuda').manual_seed(7)
side, a = 7, 0.215
grid = torch.stack(torch.meshgrid(*[torch.arange(side, dtype=torch.float32)] * 3, indexing='ij'), -1).reshape(-1, 3)[:n]
x = (grid.cuda()[None] * a + (torch.rand(B, n, 3, device='cuda', generator=gen) - 0.5) * 0.3 * a).reshape(B, 3 * n)
torch.manual_seed(0)
dyn = EGNNDynamics(node_types=[i % 4 for i in range(n)], r_cutoff=4.0, initialize_identity=False).cuda()
eps = torch.randn(B, 3 * n, device='cuda', generator=gen)
for split in (True, False):
    dyn.split_gemm = split
    with torch.no_grad():
        v1, j1 = dyn.jvp(0.3, x, eps)
        v2, j2 = dyn.jvp(0.3, x, eps)
        vs, js = dyn.jvp(0.3, x[:48].clone(), eps[:48].clone())
        vm, jm = dyn.jvp(0.3, x[100:148].clone(), eps[100:148].clone())
    print('split', split, 'repeat equal: vel', torch.equal(v1, v2), 'jvp', torch.equal(j1, j2),
          '| slice[0:48] equal: vel', torch.equal(vs, v1[:48]), 'jvp', torch.equal(js, j1[:48]), float((js - j1[:48]).abs().max()),
          '| slice[100:148]: vel', torch.equal(vm, v1[100:148]), 'jvp', torch.equal(jm, j1[100:148]), float((jm - j1[100:148]).abs().max()),
          'jvp scale', float(j1.abs().max()))
