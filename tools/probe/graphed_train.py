"""cfg1-sized training step: eager against HIP-graph replay (tfep_amd.graphs.GraphedTrainingStep)."""
import copy, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.graphs import GraphedTrainingStep
from tfep_amd.loss import BoltzmannKLDivLoss
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF, SequentialFlow
dev = torch.device('cuda:0')
D, B = 66, 1024
torch.manual_seed(0)
with torch.device(dev):
    flow = SequentialFlow(*[MAF(generate_degrees(D, o), initialize_identity=False) for o in ('ascending', 'descending')])
flow2 = copy.deepcopy(flow)
xs = [torch.randn(B, D, device=dev) for _ in range(8)]
loss_mod = BoltzmannKLDivLoss()
loss_fn = lambda y, l: loss_mod((y ** 2).sum(dim=1), l)
opt = torch.optim.SGD(flow.parameters(), lr=1e-3)
opt2 = torch.optim.SGD(flow2.parameters(), lr=1e-3)
g = GraphedTrainingStep(flow2, loss_fn, opt2, B, D)
flow2.load_state_dict(flow.state_dict())       # the warm-up took optimiser steps: start both from the same point
le, lg = [], []
for x in xs:
    opt.zero_grad(set_to_none=True)
    y, l = flow(x); loss = loss_fn(y, l); loss.backward(); opt.step(); le.append(float(loss))
    lg.append(float(g(x)))
print('eager losses  ', [round(v, 5) for v in le])
print('graphed losses', [round(v, 5) for v in lg])
dmax = max(float((a - b).abs().max()) for a, b in zip(flow.state_dict().values(), flow2.state_dict().values()) if a.is_floating_point())
print('max parameter difference after 8 steps', dmax)
for name, fn in (('eager', lambda x: (opt.zero_grad(set_to_none=True), loss_fn(*flow(x)).backward(), opt.step())), ('graph replay', g)):
    for _ in range(5):
        fn(xs[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        fn(xs[0])
    torch.cuda.synchronize()
    print(f'{name}: {(time.perf_counter() - t0) * 10:.3f} ms per step')
