import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_randomized as t
seed = int(sys.argv[1])
spec, B, kind = t.random_case(seed)
torch.manual_seed(seed)
maf = t.build(spec)
with torch.no_grad():
    for n, p in maf.named_parameters():
        if n.endswith('weight_g'):
            p.mul_(torch.rand_like(p) + 0.5)
D = len(spec['degrees_in'])
gen = torch.Generator().manual_seed(seed + 1000)
x = torch.randn(B, D, generator=gen) * 1.3
maf = maf.cuda(); xg = x.cuda()
import os
os.environ['TFEP_DBG_INV'] = '1'
dbg = {}
with torch.no_grad():
    y, l = maf(xg)
    for name, kw in (('fused', dict(blocked_inverse=True, fused_inverse=True)), ('stepwise', dict(blocked_inverse=True, fused_inverse=False)),
                     ('perdegree', dict(blocked_inverse=False, fused_inverse=False))):
        for k, v in kw.items():
            setattr(maf, k, v)
        maf._dev.clear()
        xi, li = maf.inverse(y)
        if name != 'perdegree':
            dbg[name] = maf._dbg
        print(name, 'round trip max err %.3e' % (xi - xg).abs().max().item(), 'cols with err', ((xi - xg).abs().amax(0) > 1e-3).nonzero().flatten().tolist())

hf, hs = dbg['fused']['h'][0], dbg['stepwise']['h'][0]
d = (hf - hs).abs().amax(0)
print('h0 units with err', (d > 1e-4).nonzero().flatten().tolist(), 'of', hf.shape[1])
made = maf._conditioner
import torch as T
hid = T.sort(made._degrees[1].cpu()).values
print('sorted hidden degrees', hid.tolist())
