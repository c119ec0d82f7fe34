#!/usr/bin/env python
"""Headline benchmark: samples/s of forward + log|det J| for the cfg2 / cfg3 flow of BASELINE.json
(4-layer MAF + RQ-spline(8 bins), 3x1000 atoms = 3000 features, batch 65536, fp32) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over the 65536-sample batch: for each of the 4 MAF layers the masked
weight-norm re-pack (the reference recomputes it on every forward) and its conversion to split-f16
rows, the conversion of each layer's activations, the two hidden GEMMs (+ELU) and the fused
output-GEMM + spline + log-det kernel (split-f16 MFMA by default, exact-fp32 MFMA with
TFEP_SPLIT_GEMM=0), then the TFEP free-energy estimator over the batch's log-weights (sufficient statistics + one
RCCL all-gather of 9 scalars per rank when N > 1).  Inputs are resident in HBM before the timed region.

N > 1 is BASELINE config 3: the SAME 65536-sample batch sharded by rows over the ranks (8192 per GPU at N = 8,
``tfep_amd.distributed.shard_rows``), a full weight replica per rank, no collective on the data path: STRONG scaling
(``--scaling weak`` gives every rank its own 65536 rows instead).  Rank 0 prints ONE JSON line.  After the timed
headline region the same process also times (a) the step with the packed weights cached across forwards
(``cached_repack``; the headline re-packs every step), (b) the step on the exact-fp32 MFMA GEMMs (``exact_fp32``),
(c) at N = 1 one layer's blocked inverse (8192 rows) and training step (16 384 rows) with their rooflines
(``other_paths``) and (d), on rank 0 at N = 1, the CPU restatement of the path on the host cores (``cpu_baseline``).
(``TFEP_BENCH_ARMS=cached,exact,inverse,train,cfg45`` -- a subset -- selects which of the extra arms run, for debugging
one of them; the default is all of them.)  ``other_paths`` also carries BASELINE configs 4 and 5 at size: the forward of
the 4-layer circular-spline and Moebius flows on 512 torsions (batch 131 072) and one dynamics + Jacobian-vector-product
evaluation of the EGNN continuous flow (3 x 256 atoms, batch 16 384), each with its roofline.
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


def cpu_baseline(flow, D, n_bins, chunks):
    """Time the torch-CPU restatement of the path (``oracle/torch_cpu.py``: fp32, no autograd, every host core through
    torch's intra-op pool / MKL) for ONE MAF layer on a chunk of samples and scale to all layers,
    once per chunk size in ``chunks`` (default 1024 and 4096): 1024 is BASELINE.md's protocol; a larger chunk amortises the per-call weight passes
    (weight-norm + mask over 5.6 GB per layer, ~2 s whatever the chunk) and is the CPU's best case.  ``value`` is the
    BEST of the runs.  kind 'port': the CPU restatement of the reference algorithm, not the reference itself."""
    from oracle import torch_cpu
    cores = os.cpu_count() or 1
    threads = torch.get_num_threads()
    sd = flow[0]._conditioner.state_dict()
    made = [{k: sd[f'layers.{2 * i}.{k}'].detach().cpu() for k in ('bias', 'mask', 'weight_g', 'weight_v')}
            for i in range(3)]
    x0, xf = torch.full((D,), -5.0), torch.full((D,), 5.0)
    n_layers = len(flow)
    xall = torch.randn(max(chunks), D, generator=torch.Generator().manual_seed(1234)).clamp_(-4.9, 4.9)
    torch_cpu.maf_forward(xall[:16], made, x0, xf, n_bins)          # page the weights in, spin the thread pool up
    runs, check = [], None
    for chunk in chunks:
        x = xall[:chunk]
        t0 = time.perf_counter()
        y, ldj = torch_cpu.maf_forward(x, made, x0, xf, n_bins)
        dt = time.perf_counter() - t0
        runs.append({'chunk': chunk, 'seconds_one_layer': round(dt, 2), 'samples_per_s': chunk / (dt * n_layers)})
        if check is None:
            check = (x.numpy(), y.numpy(), ldj.numpy())
        del y, ldj
    best = max(runs, key=lambda r: r['samples_per_s'])
    try:
        vs_fp64 = fp64_check(torch_cpu, flow, made, D, n_bins, next(flow.parameters()).device)
    except Exception as e:                                  # (the baseline number must still be reported)
        vs_fp64 = {'failed': f'{type(e).__name__}: {e}'}
    return {
        'value': best['samples_per_s'], 'unit': 'samples/s', 'cores': int(threads), 'kind': 'port', 'runs': runs,
        'sample': f'torch-CPU fp32 restatement (oracle/torch_cpu.py), 1 of {n_layers} MAF layers (weight-norm + 3 masked '
                  f'linears + RQ spline) on chunks of {", ".join(str(c) for c in chunks)} samples, scaled by 1/{n_layers}; '
                  f'best: chunk {best["chunk"]} in {best["seconds_one_layer"]} s; {threads} torch threads on {cores} logical cores',
        'note': 'BASELINE.md section 2 has the reference itself at 37 samples/s (4 layers) on 8 vCPU, measured with '
                'weight_norm=False and chunk 1024; this port runs weight_norm=True as cfg2 specifies (the norm, scale and '
                'mask passes over 5.6 GB per layer are memory-bound and do not scale with the core count)',
        'vs_fp64': vs_fp64,
    }, check


def fp64_check(torch_cpu, flow, made, D, n_bins, device, rows=256):
    """The asserted comparison of tests/test_gpu_parity.py::test_cfg2_layer_vs_fp64_oracle, reported in the bench line: layer
    0 on ``rows`` samples through the default (split-f16) and the exact-fp32 GEMMs against the torch-CPU restatement in
    FLOAT64 on the same float32 weights and inputs (row chunks of the weight matrices: 8192 output units at a time), with
    the float32 restatement's own distance from it as the noise floor.  ``torch_cpu``: the oracle module, handed in by
    ``cpu_baseline`` -- the one place of this file that imports it."""
    x = torch.randn(rows, D, generator=torch.Generator().manual_seed(4321)).clamp_(-4.9, 4.9)

    def cpu(dtype):
        h = x.to(dtype)
        for i, layer in enumerate(made):
            outs = []
            for r0 in range(0, layer['bias'].numel(), 8192):
                part = {k: v[r0:r0 + 8192].to(dtype) for k, v in layer.items()}
                outs.append(torch.nn.functional.linear(h, torch_cpu.effective_weight(part), part['bias']))
            h = torch.cat(outs, dim=1)
            if i + 1 < len(made):
                h = torch.nn.functional.elu(h)
        return torch_cpu.spline_forward(x.to(dtype), h, torch.full((D,), -5.0, dtype=dtype), torch.full((D,), 5.0, dtype=dtype), n_bins)
    with torch.no_grad():
        y64, l64 = cpu(torch.float64)
        y32, l32 = cpu(torch.float32)
        out = {'rows': rows, 'reference': 'oracle/torch_cpu.py in float64 on the float32 weights and inputs',
               'fp32_cpu': {'rel_l2_y': float((y32.double() - y64).norm() / y64.norm()), 'max_abs_ldj': float((l32.double() - l64).abs().max())},
               'ldj_abs_median': float(l64.abs().median())}
        layer = flow[0]
        for name, split in (('split_f16_default', None), ('exact_fp32', False)):
            layer.split_gemm = split
            yg, lg = layer(x.to(device))
            out[name] = {'rel_l2_y': float((yg.cpu().double() - y64).norm() / y64.norm()),
                         'max_abs_ldj': float((lg.cpu().double() - l64).abs().max())}
        layer.split_gemm = None
    out['tolerance'] = 'north star: <= 1e-5 rel on y and log|det J| (asserted in tests/test_gpu_parity.py at 512 rows)'
    return out


def _clock(fn, n, device):
    fn()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(device)
    return (time.perf_counter() - t0) / n


def _mfma_roofline(flow, B, dt, note):
    """Mask-aware flops of the conditioner GEMMs (2 nnz(mask) per sample and linear) over the time, against
    the bound of the arithmetic the layers run on at this batch size."""
    nnz = sum(float(torch.count_nonzero(lin.mask)) for layer in flow for lin in layer._conditioner.layers[::2])
    split = all(layer._use_split_gemm(B) for layer in flow)
    peak = PEAK_F16_MFMA_TFLOPS / 3.0 if split else PEAK_FP32_MFMA_TFLOPS
    tf = 2.0 * nnz * B / dt / 1e12
    return {'bound': 'mfma', 'achieved': tf, 'peak': peak, 'unit': 'TFLOP/s', 'frac': tf / peak,
            'arithmetic': 'split-f16 (fp16 MFMA / 3)' if split else 'fp32 MFMA', 'note': note}


def cfg4_i_arm(device, D=512, B=131072, n_layers=4):
    """BASELINE config 4, variant (i) of SURVEY 8d: 4-layer MAF, circular RQ-8 spline + periodic embedding, 512 torsions."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.embeddings import PeriodicEmbedding
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    torch.manual_seed(0)
    with torch.device(device):
        flow = SequentialFlow(*[MAF(generate_degrees(D, 'ascending' if i % 2 == 0 else 'descending'),
                                    transformer=NeuralSplineTransformer(torch.zeros(D), torch.ones(D), 8, circular=True),
                                    embedding=PeriodicEmbedding(D, limits=[0.0, 1.0]), initialize_identity=False)
                                for i in range(n_layers)])
    x = torch.rand(B, D, device=device, generator=torch.Generator(device=device).manual_seed(4))
    with torch.no_grad():
        dt = _clock(lambda: flow(x), 3, device)
        y, _ = flow(x)
        Bi = 16384                                                  # the blocked inverse of the same flow (4 x 512 degrees)
        dti = _clock(lambda: flow.inverse(y[:Bi]), 2, device)
        xi, _ = flow.inverse(y[:Bi])
        sched = flow[0].last_inverse_schedule
        dti8 = _clock(lambda: flow.inverse(y[:8192]), 2, device)      # ... and at 8192 rows, where every pair of waves is resident
        sched8 = flow[0].last_inverse_schedule
    dcirc = (xi - x[:Bi]).abs()
    return {'workload': f'cfg4-i: {n_layers}-layer MAF + circular RQ-8 + periodic embedding, {D} torsions, batch {B}, forward + log|det J|',
            'rows': B, 'ms': 1e3 * dt, 'samples_per_s': B / dt, 'y_in_domain': bool(((y >= 0) & (y <= 1)).all()),
            'roofline': _mfma_roofline(flow, B, dt, 'fused output GEMM + circular spline epilogue'),
            'inverse': {'rows': Bi, 'ms': 1e3 * dti, 'samples_per_s': Bi / dti, 'schedule': sched,
                        'roundtrip_circle_max': float(torch.minimum(dcirc, 1 - dcirc).max()),
                        'roofline': _mfma_roofline(flow, Bi, dti, 'blocked inverse: one forward of flops; 4 x 512 sequential degrees'),
                        'at_8192_rows': {'ms': 1e3 * dti8, 'samples_per_s': 8192 / dti8, 'schedule': sched8,
                                         'roofline': _mfma_roofline(flow, 8192, dti8, 'one launch per super-block (paired 16-row kernel)')}}}


def cfg4_ii_arm(device, D=512, B=131072, n_layers=4):
    """BASELINE config 4, variant (ii): 4-layer MAF + Moebius(d = 2, unit sphere) on the torsions as unit 2-vectors (1024
    features).  3.1 M weights per layer: bound by HBM traffic (x in, y out, log-det: 8196 B per sample and layer when fused)."""
    import math
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import MoebiusTransformer
    torch.manual_seed(0)
    with torch.device(device):
        flow = SequentialFlow(*[MAF(generate_degrees(2 * D, 'ascending' if i % 2 == 0 else 'descending', repeats=2),
                                    transformer=MoebiusTransformer(dimension=2, unit_sphere=True),
                                    initialize_identity=False) for i in range(n_layers)])
    ang = torch.rand(B, D, device=device, generator=torch.Generator(device=device).manual_seed(5)) * 2 * math.pi
    x = torch.stack([torch.cos(ang), torch.sin(ang)], dim=2).reshape(B, 2 * D)
    with torch.no_grad():
        dt = _clock(lambda: flow(x), 5, device)
        y, _ = flow(x)
    alg = B * n_layers * (2 * 4 * 2 * D + 4.0)                      # x in + y out + log-det per layer
    return {'workload': f'cfg4-ii: {n_layers}-layer MAF + Moebius(d=2, unit sphere), {D} torsions as {2 * D} features, batch {B}, forward + log|det J|',
            'rows': B, 'ms': 1e3 * dt, 'samples_per_s': B / dt,
            'max_norm_error': float((y.reshape(B, D, 2).norm(dim=2) - 1).abs().max()),
            'roofline': {'bound': 'hbm', 'achieved': alg / dt / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': alg / dt / 8e12,
                         'algorithmic_bytes': alg, 'note': 'algorithmic bytes = x in + y out + log-det per layer (fused)'},
            'mfma': _mfma_roofline(flow, B, dt, 'the same time against the GEMM bound')}


def cfg5_arm(device, n=256, B=16384, n_evals=3):
    """BASELINE config 5: ONE evaluation of the EGNN dynamics with its Jacobian-vector product (the integrand of the
    continuous flow with a Hutchinson trace; the full flow is 10 rk4 steps = 40 of these) at 3 x 256 atoms, batch 16 384."""
    from tfep_amd.nn.dynamics import EGNNDynamics
    gen = torch.Generator(device=device).manual_seed(1234)
    side, density = 7, 100.0                                        # jittered cubic lattice at 100 atoms / nm^3 (tools/measure_cfg5.py)
    a = (1.0 / density) ** (1 / 3)
    g = torch.stack(torch.meshgrid(*[torch.arange(side, dtype=torch.float32)] * 3, indexing='ij'), -1).reshape(-1, 3)[:n]
    x = (g.to(device)[None] * a + (torch.rand(B, n, 3, device=device, generator=gen) - 0.5) * 0.3 * a).reshape(B, 3 * n)
    torch.manual_seed(0)
    dyn = EGNNDynamics(node_types=[i % 4 for i in range(n)], r_cutoff=2.0 * side * a, initialize_identity=False).to(device)
    eps = torch.randn(B, 3 * n, device=device, generator=gen)
    with torch.no_grad():
        dyn.jvp(0.5, x[:256], eps[:256])
        dt = _clock(lambda: dyn.jvp(0.5, x, eps, need_jvp=False), n_evals, device)
    F, L, live = 64, 4, n * (n - 1)                                 # every pair inside the cutoff
    split = os.environ.get('TFEP_EGNN_SPLIT', '1') != '0' and dyn.split_gemm is not False
    flops = float(live) * L * 6 * 2 * F * F * B                     # 3 F x F products for the value + 3 for the tangent per edge and layer
    peak = PEAK_F16_MFMA_TFLOPS / 3.0 if split else PEAK_FP32_MFMA_TFLOPS
    return {'workload': f'cfg5: EGNN dynamics (4 layers, 64 features) + JVP, 3x{n} atoms, batch {B}: one integrand evaluation '
                        f'of the continuous flow (Hutchinson trace)', 'rows': B, 'ms': 1e3 * dt, 'evaluations_per_s': 1.0 / dt,
            'flow_samples_per_s_at_40_evaluations': B / (40 * dt),
            'roofline': {'bound': 'mfma', 'achieved': flops / dt / 1e12, 'peak': peak, 'unit': 'TFLOP/s', 'frac': flops / dt / 1e12 / peak,
                         'flops_per_evaluation': flops, 'kernel': 'egnn_edge_kernel<4,true,%s>' % ('true' if split else 'false'),
                         'arithmetic': 'split-f16 (fp16 MFMA / 3)' if split else 'fp32 MFMA'}}


def spawn_ranks(n, argv=None):
    """``python bench.py --gpus N`` without an outer launcher: start ``python -m torch.distributed.run --nproc-per-node N
    bench.py ...`` as a CHILD process and hand its output and exit code through.  The parent has made no HIP call
    (importing torch and parsing arguments initialise nothing) and never replaces itself (no exec): it only waits."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)]
    cmd += list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: what RCCL needs on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    return subprocess.call(cmd, env=env)                        # stdout / stderr inherited: rank 0's JSON line passes through


def _rehearsal_stats(work):
    """TFEP_BENCH_BACKEND=gloo only (no GPU, no kernels): the 9 statistics of ``tfep_tfep_reduce`` for the plain estimator,
    in torch on the CPU, so that the launch / sharding / all-gather plumbing of the N-rank bench can be rehearsed."""
    r = work.double()
    e = -r
    m = e.max()
    ninf = float('-inf')
    return torch.stack([torch.tensor(float(r.numel()), dtype=torch.float64), r.sum(), torch.tensor(ninf, dtype=torch.float64),
                        torch.tensor(0.0, dtype=torch.float64), torch.tensor(0.0, dtype=torch.float64), m,
                        torch.exp(e - m).sum(), torch.tensor(ninf, dtype=torch.float64), torch.tensor(0.0, dtype=torch.float64)])


def rows_for_rank(batch, rank, world, scaling='strong'):
    """Rows ``[row0, row1)`` of the global batch that ``rank`` maps, and the global batch size.  strong (BASELINE
    cfg3): ``batch`` rows sharded contiguously (8192 per GPU at N = 8); weak: ``batch`` rows on every rank."""
    from tfep_amd.distributed import shard_rows
    if scaling == 'strong':
        row0, row1 = shard_rows(batch, rank, world)
        return row0, row1, batch
    return rank * batch, (rank + 1) * batch, world * batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--features', type=int, default=3000)
    ap.add_argument('--batch', type=int, default=65536, help='GLOBAL batch (row-sharded over the ranks)')
    ap.add_argument('--scaling', choices=['strong', 'weak'], default='strong',
                    help='strong (BASELINE cfg3): the batch is sharded over the ranks; weak: --batch rows per rank')
    ap.add_argument('--layers', type=int, default=4)
    ap.add_argument('--bins', type=int, default=8)
    ap.add_argument('--cpu-chunks', type=int, nargs='+', default=[1024, 4096])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-arms', action='store_true',
                    help='skip the cached-repack and exact-fp32 arms and the one-layer inverse / training-step lines')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # not under a launcher: the ranks are CHILD processes of this one, which has not touched the GPU and never will
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus and rank == 0:
        print(f'bench.py: --gpus {args.gpus} but the launcher started {world} ranks; reporting n_gpus = {world}', file=sys.stderr)
    import torch.distributed as dist
    # TFEP_BENCH_BACKEND=gloo: a REHEARSAL of the N-rank launch on a box without GPUs (tests/test_distributed_cpu.py) --
    # the spawn, the rendezvous, the row sharding and the statistics all-gather run; no kernel does, the flow is the
    # identity, and the line says so ("data": "rehearsal ..."): never a measurement.
    backend = os.environ.get('TFEP_BENCH_BACKEND', 'nccl')
    rehearsal = backend != 'nccl'
    use_dist = world > 1 or 'RANK' in os.environ          # launched by torch.distributed.run
    if rehearsal:
        device = torch.device('cpu')
        args.no_extra_arms = args.no_cpu_baseline = True
        if use_dist:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            dist.init_process_group(backend)
    else:
        device = torch.device('cuda', local_rank)
        torch.cuda.set_device(device)
        if use_dist:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            dist.init_process_group('nccl', device_id=device)

    from tfep_amd.analysis import fep_estimator
    D = args.features
    row0, row1, global_batch = rows_for_rank(args.batch, rank, world, args.scaling)
    B = row1 - row0
    flow = build_flow(D, args.layers, args.bins, device)
    # Synthetic inputs: x ~ N(0,1) clipped to the spline domain, a
    # different stream per rank; u_B, u_A synthetic reduced potentials for the log-weights.
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    x = torch.randn(B, D, device=device, generator=gen).clamp_(-4.9, 4.9)
    u_B = torch.randn(B, device=device, generator=gen)
    u_A = torch.randn(B, device=device, generator=gen)

    fused_layers = [l for l in flow if l._fused_kind() is not None]
    assert rehearsal or len(fused_layers) == len(flow), 'bench expects the fused HIP path on every layer'
    # algorithmic flops of the fused output kernel: 2 * nnz(mask_out) per sample
    nnz_out = [float(torch.count_nonzero(l._conditioner.layers[-1].mask)) for l in flow]
    nnz_all = [sum(float(torch.count_nonzero(m.mask)) for m in l._conditioner.layers[::2]) for l in flow]

    @torch.no_grad()                                 # the metric is the forward + log|det J| pass
    def step():
        y, ldj = flow(x)
        work = u_B - ldj - u_A                       # reduced work of the mapped samples
        return y, ldj, fep_estimator(work, distributed=use_dist)

    if rehearsal:
        from tfep_amd.distributed import allreduce_stats

        def step():                                  # noqa: F811  (no kernels: identity map, statistics in torch)
            st = _rehearsal_stats(u_B - u_A)
            st = allreduce_stats(st) if use_dist else st
            return x, torch.zeros(B), -(st[5] + torch.log(st[6]) - torch.log(st[0]))

    def sync():
        if use_dist:
            dist.barrier()
        if not rehearsal:
            torch.cuda.synchronize(device)

    def timed(n_warmup, n_steps):
        """W untimed steps, then exactly K steps between barrier + synchronize; MAX over ranks.  Returns
        (seconds, last output, HIP-event ms of every fused-kernel launch, its flops per launch)."""
        for _ in range(n_warmup):
            step()
        sync()
        for l in flow:
            l._profile_events = []
        t0 = time.perf_counter()
        for _ in range(n_steps):
            out = step()
        sync()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = [ev0.elapsed_time(ev1) for l in flow for ev0, ev1 in l._profile_events]
        fl = [2.0 * nnz_out[i] * B for i, l in enumerate(flow) for _ in l._profile_events]
        for l in flow:
            l._profile_events = None
        return float(t), out, ms, fl

    # ---------------------------------------------------------------- headline: weights re-packed every step
    elapsed, out, kern_ms, kern_flops = timed(args.warmup, args.steps)
    achieved = sum(kern_flops) / (sum(kern_ms) * 1e-3) / 1e12 if kern_ms else 0.0
    split = rehearsal or all(l._use_split_gemm() for l in flow)
    # every rank's fused-kernel rate (HIP events on its own stream): rank 0 reports its own as `achieved`, all of them
    # and their sum beside it
    per_rank = torch.tensor([achieved, float(np.mean(kern_ms)) if kern_ms else 0.0], dtype=torch.float64, device=device)
    if use_dist:
        gathered = [torch.empty_like(per_rank) for _ in range(world)]
        dist.all_gather(gathered, per_rank)
        per_rank_tf = [float(g[0]) for g in gathered]
        per_rank_ms = [float(g[1]) for g in gathered]
    else:
        per_rank_tf, per_rank_ms = [achieved], [float(per_rank[1])]

    # ---------------------------------------------------------------- extra arms (same process, same device)
    extra = {}
    arms = set(os.environ.get('TFEP_BENCH_ARMS', 'cached,exact,inverse,train,cfg45').split(','))     # (debugging: a subset)
    if not args.no_extra_arms and 'cached' in arms:
        for l in flow:
            l._conditioner.cache_packed_weights = True
        t_c, _, ms_c, fl_c = timed(1, args.steps)
        extra['cached_repack'] = {
            'value': global_batch * args.steps / t_c, 'ms_per_step': 1e3 * t_c / args.steps,
            'note': 'packed split-f16 weights kept across forwards while no parameter / mask version changed '
                    '(MADE.cache_packed_weights; SURVEY.md 8b); the headline value re-packs on every step'}
        for l in flow:
            l._conditioner.cache_packed_weights = False
            l._conditioner.invalidate_plan()            # frees the cached packs
    if not args.no_extra_arms and 'exact' in arms:
        for l in flow:
            l.split_gemm = False
        n_exact = min(3, args.steps)
        t_e, _, ms_e, fl_e = timed(1, n_exact)
        tf_e = sum(fl_e) / (sum(ms_e) * 1e-3) / 1e12
        extra['exact_fp32'] = {
            'value': global_batch * n_exact / t_e, 'ms_per_step': 1e3 * t_e / n_exact, 'steps': n_exact,
            'fused_kernel_tflops': tf_e, 'frac_of_157.3': tf_e / PEAK_FP32_MFMA_TFLOPS,
            'whole_step_tflops': 2.0 * sum(nnz_all) * B * world * n_exact / t_e / 1e12 / world,
            'kernel': 'gemm_kernel<2,25,EPI_SPLINE> (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate)'}
        for l in flow:
            l.split_gemm = None

    # ---------------------------------------------------------------- the rows either side of the headline
    # one layer of the same flow: blocked inverse on 8192 rows and a training step (forward + backward of every
    # parameter) on 16 384 rows, each with the mask-aware MFMA roofline of the arithmetic it runs on
    if not args.no_extra_arms and world == 1 and B >= 16384:
        try:
            from tfep_amd.loss import BoltzmannKLDivLoss
            layer, other = flow[0], {}
            for l in flow:
                l._conditioner.invalidate_plan()                # packs of the arms above: this arm times one layer on its own
            torch.cuda.empty_cache()
            flops_layer = 2.0 * nnz_all[0]
            peak_other = PEAK_F16_MFMA_TFLOPS / 3.0 if split else PEAK_FP32_MFMA_TFLOPS

            def clock(fn, n):
                fn()
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize(device)
                return (time.perf_counter() - t0) / n
            with torch.no_grad():
                y8, _ = layer(x[:8192])
                t_i = clock(lambda: layer.inverse(y8), 2) if 'inverse' in arms else 1.0
                xi, _ = layer.inverse(y8) if 'inverse' in arms else (x[:8192], None)
                # a sampling loop on fixed weights: the packs of the inverse kept across calls (MADE.cache_packed_weights)
                layer._conditioner.cache_packed_weights = True
                t_ic = clock(lambda: layer.inverse(y8), 2) if 'inverse' in arms else 1.0
                layer._conditioner.cache_packed_weights = False
                layer._conditioner.invalidate_plan()
            other['inverse_one_layer'] = {
                'rows': 8192, 'ms': 1e3 * t_i, 'samples_per_s': 8192 / t_i, 'ms_cached_packs': 1e3 * t_ic,
                'roundtrip_max_abs': float((xi - x[:8192]).abs().max()), 'schedule': layer.last_inverse_schedule,
                'roofline': {'bound': 'mfma', 'achieved': flops_layer * 8192 / t_i / 1e12, 'peak': peak_other, 'unit': 'TFLOP/s',
                             'frac': flops_layer * 8192 / t_i / 1e12 / peak_other,
                             'note': 'one forward of flops; 3000 sequential degree steps (blocked forward substitution)'}}
            del y8, xi
            if 'inverse' in arms:
                # the same layer's inverse at the FULL batch of the headline (65 536 rows: one sample row per lane of the
                # block kernel, block-by-block launches -- the 16-row pairs of the 8192-row call would not be resident)
                with torch.no_grad():
                    yf, _ = layer(x)
                    t_if = clock(lambda: layer.inverse(yf), 1)
                    xf_, _ = layer.inverse(yf)
                other['inverse_full_batch'] = {
                    'rows': B, 'ms': 1e3 * t_if, 'samples_per_s': B / t_if, 'schedule': layer.last_inverse_schedule,
                    'roundtrip_max_abs': float((xf_ - x).abs().max()),
                    'roofline': {'bound': 'mfma', 'achieved': flops_layer * B / t_if / 1e12, 'peak': peak_other, 'unit': 'TFLOP/s',
                                 'frac': flops_layer * B / t_if / 1e12 / peak_other, 'note': 'one forward of flops'}}
                del yf, xf_
                torch.cuda.empty_cache()
            c = torch.rand(D, device=device, generator=gen) * 0.3
            x16 = x[:16384]

            opt = torch.optim.SGD(layer.parameters(), lr=1e-7)

            def train_step():
                for prm in layer.parameters():
                    prm.grad = None
                yt, lt = layer(x16)
                BoltzmannKLDivLoss()((c * yt ** 2).sum(dim=1), lt).backward()
                opt.step()          # parameters change: every step packs its weights again, as a real loop does
            t_t = clock(train_step, 2)

            def train_step_no_update():                # the same step without the optimiser: packed weights are re-used
                for prm in layer.parameters():
                    prm.grad = None
                yt, lt = layer(x16)
                BoltzmannKLDivLoss()((c * yt ** 2).sum(dim=1), lt).backward()
            t_n = clock(train_step_no_update, 2)
            from tfep_amd.nn.flows import _backward
            free_b, total_b = torch.cuda.mem_get_info(device)
            other['training_step_one_layer'] = {
                'rows': 16384, 'ms': 1e3 * t_t, 'samples_per_s': 16384 / t_t,
                'ms_without_update': 1e3 * t_n, 'frac_without_update': 3.0 * flops_layer * 16384 / t_n / 1e12 / peak_other,
                'activations_kept': bool(_backward.saves_activations_at(layer, 16384)),
                'free_memory_gib': round(free_b / 2 ** 30, 1),
                'roofline': {'bound': 'mfma', 'achieved': 3.0 * flops_layer * 16384 / t_t / 1e12, 'peak': peak_other,
                             'unit': 'TFLOP/s', 'frac': 3.0 * flops_layer * 16384 / t_t / 1e12 / peak_other,
                             'note': 'forward + grad_input + grad_weight of every masked linear (activations kept, no recompute) + SGD '
                                     'update: the weights are packed again on every step'}}
            for prm in layer.parameters():
                prm.grad = None
            extra['other_paths'] = other
        except Exception as e:                                  # the headline number must still print
            extra['other_paths'] = {'failed': f'{type(e).__name__}: {e}'}
        # BASELINE configs 4 and 5: parity-test cases, timed here so that they have a driver-run number
        other = extra.setdefault('other_paths', {})
        for name, arm in (('cfg4_i_forward', cfg4_i_arm), ('cfg4_ii_forward', cfg4_ii_arm), ('cfg5_eval', cfg5_arm)):
            if name.split('_')[0] not in arms and 'cfg45' not in arms:
                continue
            try:
                torch.cuda.empty_cache()
                other[name] = arm(device)
            except Exception as e:
                other[name] = {'failed': f'{type(e).__name__}: {e}'}

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the
    # figure comes from the committed rocprofv3 --pmc passes of this same command (profiles/README.md).
    if split:
        # three fp16 MFMAs per fp32 product: the matrix-pipe bound for fp32-equivalent flops is a third of the
        # dense fp16 peak
        peak = PEAK_F16_MFMA_TFLOPS / 3.0
        kernel = 'split_gemm_kernel<25,EPI_SPLINE> (fused MADE output layer + RQ spline + log-det; 3 x v_mfma_f32_16x16x32_f16 per fp32 product)'
        tnames = ('r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_final_pmc_traffic.json')
    else:
        peak = PEAK_FP32_MFMA_TFLOPS
        kernel = 'gemm_kernel<2,25,EPI_SPLINE> (fused MADE output layer + RQ spline + log-det; v_mfma_f32_16x16x4_f32)'
        tnames = ('r01_pmc_traffic.json',)
    traffic, traffic_src = None, None
    if D == 3000 and B == 65536 and args.layers == 4:
        for tname in tnames:
            tpath = os.path.join(ROOT, 'profiles', tname)
            try:
                traffic, traffic_src = json.load(open(tpath))['hbm_bytes_per_launch'], 'profiles/' + tname
                break
            except Exception:
                continue

    if rank == 0:
        res = {
            'metric': 'samples/s (fwd+log|detJ|) MAF+RQ-spline, 3N=3000, batch 64k',
            'value': None if rehearsal else global_batch * args.steps / elapsed,
            'unit': 'samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32', 'repack': 'every_step',
            'data': 'synthetic' if not rehearsal else
                    f'rehearsal on {backend} without a GPU: identity flow, no kernels -- NOT a measurement',
            'collective_world_size': dist.get_world_size() if use_dist else 1,
            'collective_backend': (dist.get_backend() if use_dist else None),
            'gemm_arithmetic': ('fp32 operands as fp16 hi+lo halves, 3 fp16 MFMAs per product, fp32 accumulate '
                                '(fp32-equivalent; spline / log-det in fp64)' if split else
                                'fp32 MFMA, fp32 accumulate (spline / log-det in fp64)'),
            'config': {'workload': f'cfg{2 if world == 1 else 3}: {args.layers}-layer MAF + RQ neural-spline ({args.bins} bins), '
                                   f'{D} features (3x{D // 3} atoms), global batch {global_batch} '
                                   f'({B} rows per GPU), fp32, forward + log|det J| + TFEP estimator',
                       'global_batch': global_batch, 'rows_per_gpu': B, 'features': D, 'layers': args.layers,
                       'hidden_width': int(flow[0]._conditioner.dimensions_hidden[0]),
                       'parallelism': f'dp{world} (row-sharded batch, weight replicas, 9-scalar RCCL all-gather)'},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic, 'traffic_source': traffic_src, 'kernel': kernel,
                         'peak_basis': ('fp32-equivalent flops; dense fp16 MFMA peak 2516.6 TFLOP/s / 3 MFMAs per product'
                                        if split else 'dense fp32 MFMA peak'),
                         'vs_fp32_mfma_peak': achieved / PEAK_FP32_MFMA_TFLOPS,
                         # what the matrix pipes sustain on this board with nothing else in flight (pure-register
                         # 16x16x32 f16 MFMA loop on toggling operands: 1914 TFLOP/s at the power limit, / 3)
                         'vs_power_limited_mfma_ceiling': (achieved / (1914.0 / 3.0)) if split else None,
                         'power_limited_ceiling_source': 'profiles/r01_mfma_shape_probe.txt' if split else None,
                         'flops_per_launch': kern_flops[0] if kern_flops else None,
                         'avg_launch_ms': float(np.mean(kern_ms)) if kern_ms else None,
                         # N > 1: `achieved` / `frac` are rank 0's GPU; every rank's rate and the whole job's beside them
                         'per_gpu_tflops': per_rank_tf, 'per_gpu_avg_launch_ms': per_rank_ms,
                         'aggregate_tflops': float(sum(per_rank_tf)), 'aggregate_peak': peak * world,
                         'aggregate_frac': float(sum(per_rank_tf)) / (peak * world),
                         # the same launches counted as dense GEMMs (no credit for skipping the masked half)
                         'dense_equivalent_tflops': achieved * sum(float(l._conditioner.layers[-1].mask.numel()) for l in flow)
                         / sum(nnz_out),
                         'whole_step_tflops': 2.0 * sum(nnz_all) * B * args.steps / elapsed / 1e12,
                         'whole_step_aggregate_tflops': 2.0 * sum(nnz_all) * global_batch * args.steps / elapsed / 1e12},
            'delta_f_estimate': float(out[2]),
        }
        res.update(extra)
        if not args.no_cpu_baseline and world == 1:
            try:
                base, (xs, ys, ls) = cpu_baseline(flow, D, args.bins, args.cpu_chunks)
                res['cpu_baseline'] = base
                # the same chunk through layer 0 on the GPU, checked against the CPU result
                with torch.no_grad():
                    yg, lg = flow[0](torch.from_numpy(xs).to(device))
                res['cpu_check'] = {'rel_l2_y': float(np.linalg.norm(yg.cpu().numpy() - ys) / np.linalg.norm(ys)),
                                    'max_abs_ldj': float(np.abs(lg.cpu().numpy() - ls).max()),
                                    'note': 'GPU layer 0 vs the fp32 CPU restatement on the CPU chunk; the asserted fp64 '
                                            'comparison at this width is tests/test_gpu_parity.py::test_cfg2_layer_vs_fp64_oracle'}
            except Exception as e:                              # the headline number must still print
                res['cpu_baseline'] = {'value': None, 'unit': 'samples/s', 'cores': os.cpu_count(), 'kind': 'port',
                                       'sample': f'failed: {type(e).__name__}: {e}'}
            if isinstance(res.get('cpu_baseline'), dict) and 'vs_fp64' in res['cpu_baseline']:
                res.setdefault('cpu_check', {})['vs_fp64'] = res['cpu_baseline'].pop('vs_fp64')
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
