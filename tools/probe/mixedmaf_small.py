"""A small flow built like the reference's MixedMAFMap (app/mixedmaf.py:330-360, 770-821): eager / HIP-graph time per call
of forward and inverse, fused against generic.  (probe)"""
import json
import sys
import time

import torch

from tfep_amd.graphs import GraphedFlow
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.embeddings import PeriodicEmbedding
from tfep_amd.nn.flows import MAF, SequentialFlow
from tfep_amd.nn.transformers import MixedTransformer, NeuralSplineTransformer, VolumePreservingShiftTransformer

D, B, n_layers = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 1024, 6
torch.manual_seed(5)
perm = torch.randperm(D)
n_cond = 5
cond, mapped = perm[:n_cond].sort().values, perm[n_cond:].sort().values
pos = torch.randperm(len(mapped))
m = len(mapped) - 6
sizes = dict(distances=m // 4, angles=m // 4, torsions=m // 3, reference=6)
sizes['cartesians'] = len(mapped) - sum(sizes.values())
idx, start = {}, 0
for k in ('distances', 'angles', 'torsions', 'cartesians', 'reference'):
    idx[k] = pos[start:start + sizes[k]].sort().values
    start += sizes[k]


def layer(order):
    n = sizes
    members = [
        NeuralSplineTransformer(torch.full((n['distances'],), 0.8), torch.full((n['distances'],), 2.5), 5,
                                identity_boundary_slopes=True, learn_upper_bound=True),
        NeuralSplineTransformer(torch.zeros(n['angles']), torch.ones(n['angles']), 5),
        NeuralSplineTransformer(torch.zeros(n['torsions']), torch.ones(n['torsions']), 5, circular=True),
        NeuralSplineTransformer(torch.full((n['cartesians'],), -1.5), torch.full((n['cartesians'],), 1.5), 5,
                                identity_boundary_slopes=True, learn_lower_bound=True, learn_upper_bound=True),
        VolumePreservingShiftTransformer(),
    ]
    return MAF(generate_degrees(D, order, conditioning_indices=cond.tolist()),
               transformer=MixedTransformer(members, [idx[k] for k in ('distances', 'angles', 'torsions', 'cartesians', 'reference')]),
               embedding=PeriodicEmbedding(D, limits=[0.0, 1.0], periodic_indices=mapped[idx['torsions']].tolist()),
               initialize_identity=False)


flow = SequentialFlow(*[layer('ascending' if i % 2 == 0 else 'descending') for i in range(n_layers)]).cuda()
x = torch.rand(B, D, device='cuda')
x[:, mapped[idx['distances']]] = x[:, mapped[idx['distances']]] * 1.5 + 0.9
x[:, mapped[idx['reference']]] = 0.0


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


if len(sys.argv) > 2 and sys.argv[2] == 'inverse-blocks':         # degrees per block of the blocked inverse
    with torch.no_grad():
        y, _ = flow(x)
        for G in (4, 8, 16, 32, 64):
            for lay in flow:
                lay.inverse_block = G
            print(G, round(timeit(lambda: flow.inverse(y), 5)[0], 2), 'ms', flush=True)
    sys.exit(0)

if len(sys.argv) > 2 and sys.argv[2] == 'train':                  # a training step, eager and as one HIP graph
    from tfep_amd.graphs import GraphedTrainingStep
    from tfep_amd.loss import BoltzmannKLDivLoss
    c = torch.rand(D, device='cuda') * 0.3
    opt = torch.optim.SGD(flow.parameters(), lr=1e-6)
    loss_fn = BoltzmannKLDivLoss()

    def step():
        opt.zero_grad(set_to_none=True)
        y, l = flow(x)
        loss = loss_fn((c * y ** 2).sum(dim=1), l)
        loss.backward()
        opt.step()
        return loss
    t_eager, loss = timeit(step, 10)
    n_grads = sum(p.grad is not None for p in flow.parameters())
    res = dict(D=D, batch=B, layers=n_layers, train_eager_ms=t_eager, loss=float(loss), params_with_grad=n_grads,
               n_params=sum(1 for _ in flow.parameters()))
    del loss
    opt.zero_grad(set_to_none=True)
    try:
        g = GraphedTrainingStep(flow, lambda y, l: loss_fn((c * y ** 2).sum(dim=1), l), opt, B, D)
        res['train_graph_ms'], lg = timeit(lambda: g(x), 10)
        res['loss_graph'] = float(lg)
    except Exception as e:
        res['train_graph_failed'] = f'{type(e).__name__}: {str(e)[:200]}'
    print(json.dumps(res))
    sys.exit(0)

if len(sys.argv) > 2 and sys.argv[2] == 'cprofile':               # host time of the eager forward
    import cProfile
    import pstats
    with torch.no_grad():
        timeit(lambda: flow(x))
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            flow(x)
        torch.cuda.synchronize()
        pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
    sys.exit(0)

if len(sys.argv) > 2 and sys.argv[2] == 'forward-only':           # under rocprofv3: 20 fused forward passes, nothing else
    with torch.no_grad():
        print(timeit(lambda: flow(x))[0])
    sys.exit(0)

with torch.no_grad():
    res = dict(D=D, batch=B, layers=n_layers)
    res['forward_ms'], (y, l) = timeit(lambda: flow(x))
    for lay in flow:
        lay.fused = False
    res['forward_generic_ms'], (yg, lg) = timeit(lambda: flow(x))
    for lay in flow:
        lay.fused = True
    res['forward_fused_ms'], _ = timeit(lambda: flow(x))
    for lay in flow:
        lay.fused = None                                # by size (the default)
    res['fused_vs_generic_max_abs'] = float((y - yg).abs().max())
    res['inverse_ms'], (xb, lb) = timeit(lambda: flow.inverse(y), 5)
    res['roundtrip_max_abs_nonperiodic'] = float((xb - x)[:, mapped[idx['distances']]].abs().max())
    for lay in flow:
        lay.blocked_inverse = False
    res['inverse_pass_per_degree_ms'], _ = timeit(lambda: flow.inverse(y), 2)
    for lay in flow:
        lay.blocked_inverse = True
    g = GraphedFlow(flow, B, D)
    res['forward_graph_ms'], _ = timeit(lambda: g(x))
    gi = GraphedFlow(flow, B, D, inverse=True)
    res['inverse_graph_ms'], _ = timeit(lambda: gi(y), 5)
print(json.dumps(res))
