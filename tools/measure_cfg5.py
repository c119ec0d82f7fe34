#!/usr/bin/env python
"""BASELINE config 5 at size: continuous flow over EGNN dynamics, 3 x 256 atoms, batch 16384, 10 ODE steps (rk4 on a
fixed grid of 10 steps = 40 dynamics evaluations), Hutchinson trace.  Prints one JSON line with its own roofline.

    python tools/measure_cfg5.py [--batch 16384] [--atoms 256] [--steps 10] [--solver rk4] [--evals-only N]

Algorithmic work: per live edge and layer three F x F products for the value and three for the tangent
= 6 * 2 * F^2 flop (F = 64: 49 152); with ``--regularization`` the reverse pass runs instead (15 products: the
Hutchinson-Frobenius estimate needs e^T J itself); the node-level products are < 1 % and not counted.  ``--evals-only N`` times N single
dynamics + JVP evaluations instead of the whole flow (a quick look at the edge kernel).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PEAK_FP32_MFMA = 157.3                 # exact-fp32 chain (v_mfma_f32_16x16x4_f32)
PEAK_SPLIT = 2516.6 / 3.0              # split-f16 chain: dense fp16 MFMA peak / 3 MFMAs per fp32 product (bench.py's basis)


def peak_of_chain():
    return PEAK_SPLIT if os.environ.get('TFEP_EGNN_SPLIT', '1') != '0' else PEAK_FP32_MFMA


def synthetic_positions(B, n, density, gen, device):
    """Atoms on a jittered cubic lattice at the given number density (per nm^3): liquid-like, no overlaps."""
    side = int(round(n ** (1 / 3) + 0.499))
    a = (1.0 / density) ** (1 / 3)
    g = torch.stack(torch.meshgrid(*[torch.arange(side, dtype=torch.float32)] * 3, indexing='ij'), -1).reshape(-1, 3)[:n]
    x = g.to(device)[None] * a + (torch.rand(B, n, 3, device=device, generator=gen) - 0.5) * 0.3 * a
    return x.reshape(B, 3 * n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16384)
    ap.add_argument('--atoms', type=int, default=256)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--solver', default='rk4')
    ap.add_argument('--cutoff', type=float, default=None, help='nm; default: every pair inside the cutoff')
    ap.add_argument('--density', type=float, default=100.0, help='atoms per nm^3 of the synthetic configuration')
    ap.add_argument('--evals-only', type=int, default=0)
    ap.add_argument('--regularization', action='store_true')
    args = ap.parse_args()
    from tfep_amd.nn.dynamics import EGNNDynamics
    from tfep_amd.nn.flows import ContinuousFlow
    dev = torch.device('cuda')
    B, n = args.batch, args.atoms
    gen = torch.Generator(device=dev).manual_seed(1234)
    x = synthetic_positions(B, n, args.density, gen, dev)
    box = (x.reshape(B, n, 3).amax(1) - x.reshape(B, n, 3).amin(1)).max().item()
    r_cutoff = args.cutoff if args.cutoff is not None else 2.0 * box          # all pairs
    torch.manual_seed(0)
    dyn = EGNNDynamics(node_types=[i % 4 for i in range(n)], r_cutoff=r_cutoff, initialize_identity=False).to(dev)
    pos = x.reshape(B, n, 3)[:64]
    d = (pos[:, :, None] - pos[:, None]).norm(dim=-1)
    live = float(((d <= r_cutoff) & (d > 0)).float().sum(dim=(1, 2)).mean())       # live directed edges per sample
    F, L = 64, 4
    # per live edge and layer: forward-mode evaluation = 3 products for the value + 3 for the tangent; with the regulariser
    # the reverse pass runs instead: 3 (forward, kept) + 2 passes x (3 recomputed + 3 reverse) = 15 products
    units = 15 if (args.regularization and not args.evals_only) else 6
    flop_eval = live * L * units * 2 * F * F * B
    res = dict(config=dict(workload=f'cfg5: continuous flow, EGNN dynamics (4 layers, 64 features, 64 radial basis '
                                    f'functions), 3x{n} atoms, batch {B}, {args.solver} with {args.steps} steps, Hutchinson trace',
                           r_cutoff=r_cutoff, live_edges_per_sample=live, all_pairs=n * (n - 1)),
               dtype='f32', data='synthetic')
    eps = torch.randn(1, B, 3 * n, device=dev, generator=gen)
    with torch.no_grad():
        dyn.jvp(0.5, x[:256], eps[0, :256])                                        # warm-up (kernel attributes, allocator)
        torch.cuda.synchronize()
        if args.evals_only:
            times = []
            for _ in range(args.evals_only):
                t0 = time.perf_counter()
                dyn.jvp(0.5, x, eps[0], need_jvp=False)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            dt = sorted(times)[len(times) // 2]                              # median (the first one warms the clocks)
            res.update(metric='s per dynamics + JVP evaluation', value=dt, unit='s', each=[round(t, 4) for t in times],
                       roofline=dict(bound='mfma', achieved=flop_eval / dt / 1e12, peak=peak_of_chain(), unit='TFLOP/s',
                                     frac=flop_eval / dt / 1e12 / peak_of_chain(), flops_per_evaluation=flop_eval,
                                     vs_fp32_mfma_peak=flop_eval / dt / 1e12 / PEAK_FP32_MFMA))
        else:
            flow = ContinuousFlow(dyn, solver=args.solver, solver_options={'step_size': 1.0 / args.steps},
                                  regularization=args.regularization)
            flow.ode_func.fixed_noise = eps
            t0 = time.perf_counter()
            out = flow(x)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            n_eval = flow.last_solver_stats['n_evaluations']
            res.update(metric='samples/s (fwd + Hutchinson log-det trace) continuous flow', value=B / dt, unit='samples/s',
                       seconds=dt, n_evaluations=n_eval, trace_mean=float(out[1].mean()),
                       displacement_rms=float((out[0] - x).pow(2).mean().sqrt()),
                       roofline=dict(bound='mfma', achieved=flop_eval * n_eval / dt / 1e12, peak=peak_of_chain(),
                                     unit='TFLOP/s', frac=flop_eval * n_eval / dt / 1e12 / peak_of_chain(),
                                     vs_fp32_mfma_peak=flop_eval * n_eval / dt / 1e12 / PEAK_FP32_MFMA,
                                     flops_per_evaluation=flop_eval, products_per_edge_and_layer=units,
                                     kernel=('egnn_edge_kernel<4,false,S> + 2 x egnn_edge_bwd_kernel<4,*,S> (reverse pass: e^T J)'
                                             if units == 15 else 'egnn_edge_kernel<4,true,S> (value + forward-mode tangent)'),
                                     arithmetic='split-f16 (3 x v_mfma_f32_16x16x32_f16 per fp32 product)'
                                     if os.environ.get('TFEP_EGNN_SPLIT', '1') != '0' else 'v_mfma_f32_16x16x4_f32'))
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
