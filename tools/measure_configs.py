#!/usr/bin/env python
"""Throughput of the other BASELINE.json configurations (parity-test cases, not the bench line) and of
the inverse, for DESIGN.md.  Prints one JSON line per measurement.  Run on an MI355X."""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfep_amd.nn.conditioners import generate_degrees  # noqa: E402
from tfep_amd.nn.embeddings import PeriodicEmbedding  # noqa: E402
from tfep_amd.nn.flows import MAF, SequentialFlow  # noqa: E402
from tfep_amd.nn.transformers import MoebiusTransformer, NeuralSplineTransformer  # noqa: E402

dev = torch.device('cuda')


def timeit(fn, warmup=1, steps=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def report(name, B, dt, **kw):
    print(json.dumps(dict(config=name, batch=B, ms=round(dt * 1e3, 3), samples_per_s=round(B / dt, 1), **kw)), flush=True)


PEAK_F32, PEAK_SPLIT = 157.3, 2516.6 / 3.0          # dense fp32 MFMA; dense fp16 MFMA / 3 MFMAs per fp32 product


def mfma_roofline(flow, B, dt, passes=1.0, note=None):
    """Mask-aware algorithmic flops of the conditioner GEMMs (2 nnz(mask) per sample and linear) times
    ``passes`` (1 = one forward's worth: forward, or the blocked inverse; a training step = forward + recompute +
    grad_input + grad_weight = 4) over the measured time, against the bound of the arithmetic the layers run on."""
    nnz = sum(float(torch.count_nonzero(lin.mask)) for layer in flow for lin in layer._conditioner.layers[::2])
    split = all(layer._use_split_gemm(B) for layer in flow)
    peak = PEAK_SPLIT if split else PEAK_F32
    tf = 2.0 * nnz * B * passes / dt / 1e12
    out = dict(bound='mfma', achieved=round(tf, 2), peak=round(peak, 1), unit='TFLOP/s', frac=round(tf / peak, 4),
               flops=2.0 * nnz * B * passes, arithmetic='split-f16 (fp16 MFMA / 3)' if split else 'fp32 MFMA')
    if note:
        out['note'] = note
    return out


def order(i):
    return 'ascending' if i % 2 == 0 else 'descending'


which = set(sys.argv[1:]) or {'cfg1', 'cfg2inv', 'cfg4'}
torch.manual_seed(0)

if 'cfg1' in which:
    D, B = 66, 1024
    with torch.device(dev):
        flow = SequentialFlow(*[MAF(generate_degrees(D, order(i)), initialize_identity=False) for i in range(2)])
    x = torch.randn(B, D, device=dev)
    with torch.no_grad():
        dt, (y, _) = timeit(lambda: flow(x), 3, 20)
    report('cfg1 forward: 2-layer MAF + affine, D=66', B, dt,
           roofline=mfma_roofline(flow, B, dt, note='27 k weights: launch / latency bound, absolute number only'))
    with torch.no_grad():
        dt, (xi, _) = timeit(lambda: flow.inverse(y), 1, 5)
    report('cfg1 inverse (blocked)', B, dt, roundtrip_max_abs=float((xi - x).detach().abs().max()))
    from tfep_amd.graphs import GraphedFlow
    with torch.no_grad():
        gf, gi = GraphedFlow(flow, B, D), GraphedFlow(flow, B, D, inverse=True)
        dt, _ = timeit(lambda: gf(x), 3, 50)
        report('cfg1 forward, HIP-graph replay', B, dt)
        dt, _ = timeit(lambda: gi(y), 3, 20)
        report('cfg1 inverse (blocked), HIP-graph replay', B, dt)
    for l in flow:
        l.blocked_inverse = False
    with torch.no_grad():
        dt, _ = timeit(lambda: flow.inverse(y), 1, 3)
    report('cfg1 inverse (one full pass per degree, reference algorithm)', B, dt)
    for l in flow:
        l.blocked_inverse = True
    # training step (forward, TFEP loss, backward, SGD update): eager, and captured into one HIP graph
    from tfep_amd.graphs import GraphedTrainingStep
    from tfep_amd.loss import BoltzmannKLDivLoss
    loss_mod = BoltzmannKLDivLoss()
    loss_fn = lambda yy, ll: loss_mod((yy ** 2).sum(dim=1), ll)
    opt = torch.optim.SGD(flow.parameters(), lr=1e-6)

    def eager_step():
        opt.zero_grad(set_to_none=True)
        loss_fn(*flow(x)).backward()
        opt.step()
    dt, _ = timeit(eager_step, 5, 50)
    report('cfg1 training step (forward + loss + backward + SGD), eager', B, dt)
    gstep = GraphedTrainingStep(flow, loss_fn, opt, B, D)
    dt, _ = timeit(lambda: gstep(x), 5, 50)
    report('cfg1 training step, HIP-graph replay (GraphedTrainingStep)', B, dt)

if 'cfg2inv' in which:
    D = 3000
    B = int(os.environ.get('INV_BATCH', 8192))
    with torch.device(dev):
        flow = SequentialFlow(MAF(generate_degrees(D, 'ascending'),
                                  transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                                  initialize_identity=False))
    x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
    with torch.no_grad():
        dt, (y, lf) = timeit(lambda: flow(x), 1, 2)
        report('cfg2 ONE layer forward', B, dt, roofline=mfma_roofline(flow, B, dt))
        dt, (xi, li) = timeit(lambda: flow.inverse(y), 1, 1)
    report('cfg2 ONE layer inverse (blocked, 3000 degrees)', B, dt,
           roundtrip_rel_l2=float((xi - x).detach().norm() / x.norm()), ldj_cancel_max_abs=float((lf + li).detach().abs().max()),
           roofline=mfma_roofline(flow, B, dt, note='one forward of flops; the chain of 3000 degree steps is sequential: '
                                                    'issue-rate bound block kernel (DESIGN.md section 7)'))
    from tfep_amd.graphs import GraphedFlow
    with torch.no_grad():
        gi = GraphedFlow(flow, B, D, inverse=True, warmup=1)
        dt, (xg, lg) = timeit(lambda: gi(y), 1, 2)
    dg = (xg - xi).detach().abs()
    extra = {} if float(dg.max()) == 0 else dict(max_abs_diff=float(dg.max()), rows_differing=int((dg > 0).any(1).sum()),
                                                  first_cols=torch.nonzero((dg > 0).any(0)).flatten()[:6].tolist(),
                                                  ldj_equal=bool(torch.equal(lg, li)))
    report('cfg2 ONE layer inverse (blocked), HIP-graph replay', B, dt, equals_eager=bool(torch.equal(xg, xi)),
           peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), **extra)

if 'cfg3shard' in which:
    # BASELINE cfg3 per-GPU share: the 4-layer cfg2 flow on 8192 rows (65536 / 8), weights re-packed every step
    D, B = 3000, 8192
    with torch.device(dev):
        flow = SequentialFlow(*[MAF(generate_degrees(D, order(i)),
                                    transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                                    initialize_identity=False) for i in range(4)])
    x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
    with torch.no_grad():
        dt, _ = timeit(lambda: flow(x), 2, 10)
        report('cfg3 per-GPU shard: 4-layer cfg2 flow on 8192 rows, re-pack every step', B, dt, roofline=mfma_roofline(flow, B, dt))
        for l in flow:
            l._conditioner.cache_packed_weights = True
        dt, _ = timeit(lambda: flow(x), 2, 10)
        report('cfg3 per-GPU shard: same, packed weights cached', B, dt, roofline=mfma_roofline(flow, B, dt))
    del flow

if 'variants' in which:
    # the spline layouts on the fused epilogue against the generic path (split GEMMs, the (B, P D)
    # parameters through HBM, stand-alone spline kernel).  One cfg2-sized layer (D = 3000, H = 14998) at B = 32768: K = 8 /
    # 5 / 4 plain splines and the identity-slope / learnable-bound splines of MixedMAFMap (app/mixedmaf.py:770-811).
    D, B = 3000, 32768
    x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
    ident = dict(identity_boundary_slopes=True)
    both = dict(learn_lower_bound=True, learn_upper_bound=True)
    for name, kw in (('K=8 plain', dict(n_bins=8)),
                     ('K=8 plain, fused=False', dict(n_bins=8)),
                     ('K=5 plain', dict(n_bins=5)),
                     ('K=5 plain, fused=False', dict(n_bins=5)),
                     ('K=5 circular', dict(n_bins=5, circular=True)),
                     ('K=4 plain', dict(n_bins=4)),
                     ('K=8 identity slopes', dict(n_bins=8, **ident)),
                     ('K=8 identity slopes, fused=False', dict(n_bins=8, **ident)),
                     ('K=5 identity slopes', dict(n_bins=5, **ident)),
                     ('K=5 identity slopes, fused=False', dict(n_bins=5, **ident)),
                     ('K=5 learnable bounds', dict(n_bins=5, **both)),
                     ('K=5 learnable bounds, fused=False', dict(n_bins=5, **both)),
                     ('K=5 identity slopes + both bounds learnable (16 parameters like the plain layout: its own epilogue kind)',
                      dict(n_bins=5, **ident, **both))):
        torch.manual_seed(0)
        with torch.device(dev):
            flow = SequentialFlow(MAF(generate_degrees(D, 'ascending'),
                                      transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), **kw),
                                      initialize_identity=False))
        if 'fused=False' in name:
            flow[0].fused = False
        with torch.no_grad():
            dt, _ = timeit(lambda: flow(x), 1, 3)
        P = flow[0]._transformer.n_parameters_per_feature
        report(f'one cfg2-sized layer, {name}', B, dt, params_per_feature=P, fused=flow[0]._fused_kind() is not None,
               param_tensor_gb=round(B * P * D * 4 / 1e9, 2), roofline=mfma_roofline(flow, B, dt))
        del flow

if 'variants' in which or 'mixed' in which:
    # the transformer of the reference's MixedMAFMap (app/mixedmaf.py:770-811) on one cfg2-sized layer: four 5-bin
    # spline groups (distances: identity slopes + learnable upper bound; angles: plain; torsions: circular; cartesians:
    # identity slopes + both bounds learnable) -- one fused launch per group against the generic path
    from tfep_amd.nn.transformers import MixedTransformer
    D, B = 3000, 32768
    torch.manual_seed(0)
    perm = torch.randperm(D)
    sizes = [999, 999, 900, 102]
    idx = [perm[sum(sizes[:i]):sum(sizes[:i + 1])].sort().values for i in range(4)]
    x = torch.rand(B, D, device=dev)
    for name in ('MixedMAFMap transformer (4 spline groups, one fused launch each)', 'MixedMAFMap transformer, fused=False'):
        torch.manual_seed(0)
        members = [NeuralSplineTransformer(torch.zeros(sizes[0]), torch.ones(sizes[0]), 5, identity_boundary_slopes=True,
                                           learn_upper_bound=True),
                   NeuralSplineTransformer(torch.zeros(sizes[1]), torch.ones(sizes[1]), 5),
                   NeuralSplineTransformer(torch.zeros(sizes[2]), torch.ones(sizes[2]), 5, circular=True),
                   NeuralSplineTransformer(torch.zeros(sizes[3]), torch.ones(sizes[3]), 5, identity_boundary_slopes=True,
                                           learn_lower_bound=True, learn_upper_bound=True)]
        with torch.device(dev):
            flow = SequentialFlow(MAF(generate_degrees(D, 'ascending'), transformer=MixedTransformer(members, idx),
                                      initialize_identity=False))
        if 'fused=False' in name:
            flow[0].fused = False
        with torch.no_grad():
            dt, _ = timeit(lambda: flow(x), 1, 3)
        report(f'one cfg2-sized layer, {name}', B, dt, fused=flow[0]._fused_kind() is not None,
               out_features=flow[0]._conditioner.layers[-1].out_features, roofline=mfma_roofline(flow, B, dt))
        if 'fused=False' not in name:
            # its inverse: the blocked substitution with one step per degree and member (block kernel kind 3, fp32 block
            # GEMMs; the pass-per-degree algorithm of the reference is 3000 conditioner passes, ~0.1 s each at this size)
            Bi = 8192
            with torch.no_grad():
                yi, _ = flow(x[:Bi])
                dti, (xi, _) = timeit(lambda: flow.inverse(yi), 1, 2)
            roof = mfma_roofline(flow, Bi, dti)
            roof.update(peak=round(PEAK_F32, 1), frac=round(roof['achieved'] / PEAK_F32, 4), arithmetic='fp32 MFMA',
                        note='sequential in the degree: bound by the block kernel chain, see DESIGN 7')
            report('one cfg2-sized layer, MixedMAFMap transformer: inverse (blocked, 3000 degrees, block kernel kind 3)', Bi, dti,
                   roundtrip_max_abs=float((xi - x[:Bi]).abs().max()), roofline=roof)
        del flow

if 'train' in which:
    from tfep_amd.loss import BoltzmannKLDivLoss
    D = 3000
    B = int(os.environ.get('TRAIN_BATCH', 16384))
    with torch.device(dev):
        flow = SequentialFlow(MAF(generate_degrees(D, 'ascending'),
                                  transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                                  initialize_identity=False))
    x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
    c = torch.rand(D, device=dev) * 0.3

    opt = torch.optim.SGD(flow.parameters(), lr=1e-7)

    def train_step(update=False):
        for p in flow.parameters():
            p.grad = None
        y, ldj = flow(x)
        loss = BoltzmannKLDivLoss()((c * y ** 2).sum(dim=1), ldj)
        loss.backward()
        if update:
            opt.step()                 # parameters change: the next step packs its weights again, as a real loop does
        return loss.detach()
    with torch.no_grad():
        dtf, _ = timeit(lambda: flow(x), 1, 2)
    dt, loss = timeit(train_step, 1, 2)
    dtu, _ = timeit(lambda: train_step(True), 1, 3)
    report('cfg2 ONE layer forward (no grad)', B, dtf, roofline=mfma_roofline(flow, B, dtf))
    report('cfg2 ONE layer training step (forward + backward of all parameters)', B, dt, loss=float(loss),
           peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
           roofline=mfma_roofline(flow, B, dt, passes=3.0, note='forward + grad_input + grad_weight (the forward keeps its '
                                                                'activations: no recompute; TFEP_SAVE_ACTIVATIONS_GIB)'))
    report('cfg2 ONE layer training step + SGD update (weights re-packed every step)', B, dtu,
           roofline=mfma_roofline(flow, B, dtu, passes=3.0, note='as above, parameters updated between the steps'))

if 'train4' in which:
    # the whole cfg2 flow (4 layers) in a training step: forward, TFEP loss, backward of every parameter, SGD update
    from tfep_amd.loss import BoltzmannKLDivLoss
    from tfep_amd.nn.flows import _backward
    D = 3000
    B = int(os.environ.get('TRAIN_BATCH', 16384))
    with torch.device(dev):
        flow = SequentialFlow(*[MAF(generate_degrees(D, 'ascending' if i % 2 == 0 else 'descending'),
                                    transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                                    initialize_identity=False) for i in range(4)])
    x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
    c = torch.rand(D, device=dev) * 0.3
    opt = torch.optim.SGD(flow.parameters(), lr=1e-7)

    def train_step(update=True):
        for p in flow.parameters():
            p.grad = None
        y, ldj = flow(x)
        loss = BoltzmannKLDivLoss()((c * y ** 2).sum(dim=1), ldj)
        loss.backward()
        if update:
            opt.step()
        return loss.detach()
    dtu, loss = timeit(train_step, 1, 2)
    dtn, _ = timeit(lambda: train_step(False), 1, 2)
    kept = [bool(_backward.saves_activations_at(l, B)) for l in flow]
    fl = sum(2.0 * sum(float(torch.count_nonzero(lin.mask)) for lin in l._conditioner._linears()) for l in flow)
    report('cfg2 FOUR-layer flow training step + SGD update (weights re-packed every step)', B, dtu, loss=float(loss),
           peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), activations_kept=kept,
           roofline={'bound': 'mfma', 'achieved': round(3.0 * fl * B / dtu / 1e12, 1), 'peak': 838.9, 'unit': 'TFLOP/s',
                     'frac': round(3.0 * fl * B / dtu / 1e12 / 838.9, 4), 'note': 'forward + grad_input + grad_weight of 12 masked linears'})
    report('cfg2 FOUR-layer flow training step without the update', B, dtn,
           roofline={'bound': 'mfma', 'achieved': round(3.0 * fl * B / dtn / 1e12, 1), 'peak': 838.9, 'unit': 'TFLOP/s',
                     'frac': round(3.0 * fl * B / dtn / 1e12 / 838.9, 4)})
    del flow, opt

if 'cfg4' in which:
    D, B = 512, 131072
    with torch.device(dev):
        flow = SequentialFlow(*[MAF(generate_degrees(D, order(i)),
                                    transformer=NeuralSplineTransformer(torch.zeros(D), torch.ones(D), 8, circular=True),
                                    embedding=PeriodicEmbedding(D, limits=[0.0, 1.0]), initialize_identity=False)
                                for i in range(4)])
    x = torch.rand(B, D, device=dev)
    with torch.no_grad():
        dt, (y, _) = timeit(lambda: flow(x), 1, 3)
    report('cfg4-i forward: 4-layer MAF + circular RQ-8 + periodic embedding, 512 torsions', B, dt,
           y_in_domain=bool(((y >= 0) & (y <= 1)).all()), roofline=mfma_roofline(flow, B, dt))
    Bi = 16384
    with torch.no_grad():
        dti, (xi, _) = timeit(lambda: flow.inverse(y[:Bi]), 1, 2)
    dcirc = (xi - x[:Bi]).abs()
    report('cfg4-i inverse (blocked, fused block kernel; 4 layers x 512 degrees)', Bi, dti,
           roundtrip_circle_max=float(torch.minimum(dcirc, 1 - dcirc).max()), roofline=mfma_roofline(flow, Bi, dti))
    with torch.device(dev):
        flow = SequentialFlow(*[MAF(generate_degrees(2 * D, order(i), repeats=2),
                                    transformer=MoebiusTransformer(dimension=2, unit_sphere=True),
                                    initialize_identity=False) for i in range(4)])
    ang = torch.rand(B, D, device=dev) * 2 * math.pi
    x = torch.stack([torch.cos(ang), torch.sin(ang)], dim=2).reshape(B, 2 * D)
    with torch.no_grad():
        dt, (y, _) = timeit(lambda: flow(x), 1, 3)
    # generic path: per layer x (4 KB) in, 3 activations / parameter rows of 1024 floats written and re-read, y out
    hbm = B * 4 * 1024 * 4 * (2 + 2 * 3) / dt / 1e9
    report('cfg4-ii forward: 4-layer MAF + Moebius(d=2, unit sphere), 512 torsions as 1024 features (split-f16 GEMMs by batch size)', B, dt,
           max_norm_error=float((y.reshape(B, D, 2).norm(dim=2) - 1).abs().max()),
           roofline=dict(bound='hbm', achieved=round(hbm, 1), peak=8000.0, unit='GB/s', frac=round(hbm / 8000.0, 4),
                         note='unfused path: activations and the (B, P D) parameters round-trip HBM; 3.1 M weights'),
           mfma=mfma_roofline(flow, B, dt))
    for l in flow:                                   # one arithmetic per conditioner: exact-fp32 kernels at every batch size
        l._conditioner.split_by_batch = False
    with torch.no_grad():
        dt2, _ = timeit(lambda: flow(x), 1, 3)
    report('cfg4-ii forward with split_by_batch = False (exact-fp32 GEMMs for this 3 M-weight conditioner)', B, dt2,
           mfma=mfma_roofline(flow, B, dt2))
    for l in flow:
        l._conditioner.split_by_batch = True
    with torch.no_grad():
        dti, (xi, _) = timeit(lambda: flow.inverse(y[:Bi]), 1, 2)
    report('cfg4-ii inverse (blocked; 4 layers x 512 degrees)', Bi, dti,
           roundtrip_max_abs=float((xi - x[:Bi]).abs().max()), roofline=mfma_roofline(flow, Bi, dti))

if 'hbm' in which:
    # HBM-bound kernels in isolation: algorithmic bytes / time vs the 8 TB/s HBM3E peak.
    from tfep_amd import ops
    D, B, K = 3000, 16384, 8
    P = 3 * K + 1
    x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
    par = torch.randn(B, P * D, device=dev)
    cfg = ops.SplineConfig(torch.full((D,), -5.0, device=dev), torch.full((D,), 5.0, device=dev),
                           torch.full((D,), -5.0, device=dev), torch.full((D,), 5.0, device=dev), K)
    for inverse in (False, True):
        dt, _ = timeit(lambda: ops.spline(x, par, cfg, inverse=inverse), 2, 10)
        nbytes = B * (4 * (P * D + 2 * D) + 4)
        report(f'standalone RQ-8 spline kernel ({"inverse" if inverse else "forward"}), D=3000', B, dt,
               algorithmic_GBps=round(nbytes / dt / 1e9, 1), frac_of_8TBps=round(nbytes / dt / 8e12, 3))
    par2 = torch.randn(B, 2 * D, device=dev) * 0.3
    dt, _ = timeit(lambda: ops.affine(x, par2), 2, 20)
    nbytes = B * (4 * (2 * D + 2 * D) + 4)
    report('standalone affine kernel, D=3000', B, dt, algorithmic_GBps=round(nbytes / dt / 1e9, 1),
           frac_of_8TBps=round(nbytes / dt / 8e12, 3))
    # weight re-pack of the cfg2 output layer: read v + mask, write packed (12 B per weight) + memset
    N, Kd = 75000, 14998
    v = torch.randn(N, Kd, device=dev)
    g = torch.rand(N, 1, device=dev) + 0.5
    mask = (torch.rand(N, Kd, device=dev) > 0.5).float()
    out = torch.empty(N, 15008, device=dev)
    dt, _ = timeit(lambda: ops.masked_weight_prepare(v, g, mask, out=out, n_rows_padded=N, k_padded=15008), 1, 5)
    nbytes = N * Kd * 12.0
    report('weight_prepare (weight-norm + mask + pack) 75000x14998', 1, dt, algorithmic_GBps=round(nbytes / dt / 1e9, 1),
           frac_of_8TBps=round(nbytes / dt / 8e12, 3))
