"""Device memory across repeated calls of every path (garbage collector off): must be flat after the first call."""
import gc, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.dynamics import EGNNDynamics
from tfep_amd.nn.flows import MAF, SequentialFlow, ContinuousFlow
from tfep_amd.nn.transformers import NeuralSplineTransformer
from tfep_amd.analysis import fep_estimator
dev = torch.device('cuda:0')
D, B = 600, 4096
with torch.device(dev):
    flow = SequentialFlow(*[MAF(generate_degrees(D, o), transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                                hidden_layers=[4500, 4500], initialize_identity=False) for o in ('ascending', 'descending')])
x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
gc.collect(); gc.disable()


def track(name, fn, n=5):
    vals = []
    for _ in range(n):
        fn()
        torch.cuda.synchronize()
        vals.append(torch.cuda.memory_allocated() / 2**20)
    print(f'{name}: MiB after each call {[round(v) for v in vals]}', flush=True)


with torch.no_grad():
    track('forward (no grad)', lambda: flow(x))
    y, _ = flow(x)
    track('inverse (no grad)', lambda: flow.inverse(y))
    track('fep_estimator', lambda: fep_estimator(torch.randn(B, device=dev)))
track('forward under grad mode, outputs dropped', lambda: flow(x))
dyn = EGNNDynamics(node_types=[0, 1] * 8, r_cutoff=2.0, time_feat_dim=4, node_feat_dim=16, distance_feat_dim=8, n_layers=2,
                   initialize_identity=False).to(dev)
cf = ContinuousFlow(dyn, solver='rk4', solver_options={'step_size': 0.25})
xs = torch.randn(256, 48, device=dev)
with torch.no_grad():
    track('continuous flow', lambda: cf(xs))
