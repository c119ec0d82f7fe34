import faulthandler, os, sys, torch
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.graphs import GraphedTrainingStep, GraphedFlow
from tfep_amd.loss import BoltzmannKLDivLoss
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF, SequentialFlow
dev = torch.device('cuda:0')
D, B = 66, 1024
with torch.device(dev):
    flow = SequentialFlow(*[MAF(generate_degrees(D, o), initialize_identity=False) for o in ('ascending', 'descending')])
x = torch.randn(B, D, device=dev)
variant = os.environ.get('VARIANT', 'both')
if variant in ('both', 'graphflow'):
    with torch.no_grad():
        gf = GraphedFlow(flow, B, D)
        gf(x)
        gi = GraphedFlow(flow, B, D, inverse=True)
        gi(x)
    print('graphed flows ok', flush=True)
if variant in ('inv', 'all'):
    xi, _ = flow.inverse(x)                  # under grad mode: the not-differentiable node
    print('inverse under grad ok', flush=True)
if variant in ('refinv', 'all'):
    for l in flow:
        l.blocked_inverse = False
    flow.inverse(x)
    for l in flow:
        l.blocked_inverse = True
    print('reference inverse ok', flush=True)
if variant in ('many', 'all'):
    with torch.no_grad():
        gf = GraphedFlow(flow, B, D)
        for _ in range(60):
            gf(x)
loss_mod = BoltzmannKLDivLoss()
loss_fn = lambda yy, ll: loss_mod((yy ** 2).sum(dim=1), ll)
opt = torch.optim.SGD(flow.parameters(), lr=1e-6)
if variant in ('both', 'eager', 'all', 'many'):
    for _ in range(55 if variant in ('all', 'many') else 3):
        opt.zero_grad(set_to_none=True)
        loss_fn(*flow(x)).backward()
        opt.step()
    torch.cuda.synchronize()
    print('eager steps ok', flush=True)
g = GraphedTrainingStep(flow, loss_fn, opt, B, D)
print('captured', flush=True)
for _ in range(3):
    print(float(g(x)), flush=True)
