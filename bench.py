#!/usr/bin/env python
"""Headline benchmark: samples/s of forward + log|det J| for the cfg2 flow of BASELINE.json
(4-layer MAF + RQ-spline(8 bins), 3x1000 atoms = 3000 features, batch 65536, fp32) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: for each of the 4 MAF layers the masked
weight-norm re-pack (the reference recomputes it on every forward) and its conversion to split-f16
rows, the conversion of each layer's activations, the two hidden GEMMs (+ELU) and the fused
output-GEMM + spline + log-det kernel (split-f16 MFMA by default, exact-fp32 MFMA with
TFEP_SPLIT_GEMM=0), then the TFEP free-energy
estimator over the batch's log-weights (sufficient statistics + one RCCL all-gather of 9 scalars
per rank when N > 1).  Inputs are resident in HBM before the timed region.  Weak scaling: every
rank processes its own full batch with a full weight replica; there is no collective on the data
path.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, chip table (dense fp32 matrix)
PEAK_F16_MFMA_TFLOPS = 2516.6          # dense fp16 matrix: 256 CUs x 4096 flop/clk x 2.4 GHz ("~2.5 PF")


def build_flow(D, n_layers, n_bins, device, seed=0):
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    torch.manual_seed(seed)
    layers = []
    with torch.device(device):          # build weights / masks directly in HBM (5.6 GB per layer each)
        for i in range(n_layers):
            layers.append(MAF(
                degrees_in=generate_degrees(D, 'ascending' if i % 2 == 0 else 'descending'),
                transformer=NeuralSplineTransformer(x0=torch.full((D,), -5.0), xf=torch.full((D,), 5.0),
                                                    n_bins=n_bins),
                hidden_layers=2, weight_norm=True, initialize_identity=False))
    return SequentialFlow(*layers).to(device)


def cpu_baseline(flow, D, n_bins, chunk, budget_s=30.0):
    """Time the numpy oracle (fp32) for ONE MAF layer on `chunk` samples and scale to all layers.
    kind 'port': the CPU restatement of the reference algorithm, not the reference itself."""
    from oracle import flows as oflows, made as omade
    import threadpoolctl
    cores = os.cpu_count() or 1
    layer0 = flow[0]
    sd = {k: v.detach().cpu().numpy() for k, v in layer0._conditioner.state_dict().items()}
    layer = dict(degrees_in=omade.generate_degrees(D, 'ascending'),
                 transformer=dict(type='spline', x0=np.full(D, -5.0, np.float32), xf=np.full(D, 5.0, np.float32),
                                  n_bins=n_bins),
                 embedding=None, made=omade.made_layers_from_state(sd))
    x = np.random.default_rng(1234).standard_normal((chunk, D)).astype(np.float32).clip(-4.9, 4.9)
    t0 = time.perf_counter()
    y, ldj = oflows.maf_forward(x, layer)
    dt = time.perf_counter() - t0
    n_layers = len(flow)
    try:
        threads = max(i.get('num_threads', 1) for i in threadpoolctl.threadpool_info()) or cores
    except Exception:
        threads = cores
    return {
        'value': chunk / (dt * n_layers), 'unit': 'samples/s', 'cores': int(threads), 'kind': 'port',
        'sample': f'numpy fp32 oracle, 1 of {n_layers} MAF layers (weight-norm + 3 masked linears + RQ spline) on a '
                  f'{chunk}-sample chunk in {dt:.1f} s, scaled by 1/{n_layers}; host has {cores} logical cores',
    }, (x, y, ldj)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--features', type=int, default=3000)
    ap.add_argument('--batch', type=int, default=65536, help='samples per GPU')
    ap.add_argument('--layers', type=int, default=4)
    ap.add_argument('--bins', type=int, default=8)
    ap.add_argument('--cpu-chunk', type=int, default=2048)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        if rank == 0 and world == 1 and args.gpus > 1:
            sys.exit(f'--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`')
    import torch.distributed as dist
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    use_dist = world > 1 or 'RANK' in os.environ          # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', device_id=device)

    from tfep_amd.analysis import fep_estimator
    D, B = args.features, args.batch
    flow = build_flow(D, args.layers, args.bins, device)
    # Synthetic inputs: x ~ N(0,1) clipped to the spline domain, a
    # different stream per rank; u_B, u_A synthetic reduced potentials for the log-weights.
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    x = torch.randn(B, D, device=device, generator=gen).clamp_(-4.9, 4.9)
    u_B = torch.randn(B, device=device, generator=gen)
    u_A = torch.randn(B, device=device, generator=gen)

    fused_layers = [l for l in flow if l._fused_kind() is not None]
    assert len(fused_layers) == len(flow), 'bench expects the fused HIP path on every layer'
    # algorithmic flops of the fused output kernel: 2 * nnz(mask_out) per sample
    nnz_out = [float(torch.count_nonzero(l._conditioner.layers[-1].mask)) for l in flow]
    nnz_all = [sum(float(torch.count_nonzero(m.mask)) for m in l._conditioner.layers[::2]) for l in flow]

    @torch.no_grad()                                 # the metric is the forward + log|det J| pass
    def step():
        y, ldj = flow(x)
        work = u_B - ldj - u_A                       # reduced work of the mapped samples
        return y, ldj, fep_estimator(work, distributed=use_dist)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    sync()
    for l in flow:
        l._profile_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)

    # dominant kernel: fused output GEMM + spline (one launch per layer per step)
    kern_ms = [ev0.elapsed_time(ev1) for l in flow for ev0, ev1 in l._profile_events]
    kern_flops = [2.0 * nnz_out[i] * B for i, l in enumerate(flow) for _ in l._profile_events]
    achieved = sum(kern_flops) / (sum(kern_ms) * 1e-3) / 1e12

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the
    # figure comes from the committed rocprofv3 --pmc passes of this same command (profiles/README.md).
    split = all(l._use_split_gemm() for l in flow)
    if split:
        # three fp16 MFMAs per fp32 product: the matrix-pipe bound for fp32-equivalent flops is a third of the
        # dense fp16 peak
        peak = PEAK_F16_MFMA_TFLOPS / 3.0
        kernel = 'split_gemm_kernel<25,EPI_SPLINE> (fused MADE output layer + RQ spline + log-det; 3 x v_mfma_f32_16x16x32_f16 per fp32 product)'
        tname = 'r01_final_pmc_traffic.json'             # counters of the final build of round 1 (profiles/README.md)
    else:
        peak = PEAK_FP32_MFMA_TFLOPS
        kernel = 'gemm_kernel<2,25,EPI_SPLINE> (fused MADE output layer + RQ spline + log-det; v_mfma_f32_16x16x4_f32)'
        tname = 'r01_pmc_traffic.json'
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', tname)
    if D == 3000 and B == 65536 and args.layers == 4 and os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath))['hbm_bytes_per_launch']
        except Exception:
            traffic = None

    if rank == 0:
        res = {
            'metric': 'samples/s (fwd+log|detJ|) MAF+RQ-spline, 3N=3000, batch 64k',
            'value': world * B * args.steps / elapsed,
            'unit': 'samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'gemm_arithmetic': ('fp32 operands as fp16 hi+lo halves, 3 fp16 MFMAs per product, fp32 accumulate '
                                '(fp32-equivalent; spline / log-det in fp64)' if split else
                                'fp32 MFMA, fp32 accumulate (spline / log-det in fp64)'),
            'config': {'workload': f'cfg2: {args.layers}-layer MAF + RQ neural-spline ({args.bins} bins), '
                                   f'{D} features (3x{D // 3} atoms), batch {B} per GPU, fp32, '
                                   'forward + log|det J| + TFEP estimator',
                       'global_batch': world * B, 'features': D, 'layers': args.layers,
                       'hidden_width': int(flow[0]._conditioner.dimensions_hidden[0]),
                       'parallelism': f'dp{world} (batch-sharded replicas, 9-scalar RCCL all-gather)'},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic, 'kernel': kernel,
                         'peak_basis': ('fp32-equivalent flops; dense fp16 MFMA peak 2516.6 TFLOP/s / 3 MFMAs per product'
                                        if split else 'dense fp32 MFMA peak'),
                         'vs_fp32_mfma_peak': achieved / PEAK_FP32_MFMA_TFLOPS,
                         # what the matrix pipes sustain on this board with nothing else in flight (pure-register
                         # 16x16x32 f16 MFMA loop on toggling operands: 1914 TFLOP/s at the power limit, / 3)
                         'vs_power_limited_mfma_ceiling': (achieved / (1914.0 / 3.0)) if split else None,
                         'power_limited_ceiling_source': 'profiles/r01_mfma_shape_probe.txt' if split else None,
                         'flops_per_launch': kern_flops[0], 'avg_launch_ms': float(np.mean(kern_ms)),
                         # the same launches counted as dense GEMMs (no credit for skipping the masked half)
                         'dense_equivalent_tflops': achieved * sum(float(l._conditioner.layers[-1].mask.numel()) for l in flow)
                         / sum(nnz_out),
                         'whole_step_tflops': 2.0 * sum(nnz_all) * B * args.steps / elapsed / 1e12},
            'delta_f_estimate': float(out[2]),
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                base, (xs, ys, ls) = cpu_baseline(flow, D, args.bins, args.cpu_chunk)
                res['cpu_baseline'] = base
                # the same chunk through layer 0 on the GPU, checked against the CPU result
                with torch.no_grad():
                    yg, lg = flow[0](torch.from_numpy(xs).to(device))
                res['cpu_check'] = {'rel_l2_y': float(np.linalg.norm(yg.cpu().numpy() - ys) / np.linalg.norm(ys)),
                                    'max_abs_ldj': float(np.abs(lg.cpu().numpy() - ls).max())}
            except Exception as e:                              # the headline number must still print
                res['cpu_baseline'] = {'value': None, 'unit': 'samples/s', 'cores': os.cpu_count(), 'kind': 'port',
                                       'sample': f'failed: {type(e).__name__}: {e}'}
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
