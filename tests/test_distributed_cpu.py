"""world_size-2 gloo tests (CPU) of the multi-GPU reduction path: per-rank sufficient statistics
-> all-gather -> combine must reproduce the oracle's whole-batch loss / free-energy estimate."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as gu
from oracle import loss as oloss
from tfep_amd.distributed import OverlappedGradientSync, allreduce_gradients, allreduce_stats, combine_stats, shard_rows


def shard_stats(uB, ldj, uA, lw, bias, kT=1.0, ignore_nan=False):
    """numpy restatement of the 9 statistics tfep_tfep_reduce emits for one shard (tfep_hip.h)."""
    r = (uB - ldj - uA).astype(np.float32).astype(np.float64)
    isn = np.isnan(r)
    keep = ~isn if ignore_nan else np.ones_like(isn)
    out = np.zeros(9)
    out[0] = keep.sum()
    out[1] = r[keep].sum() if ignore_nan else r.sum()
    if lw is not None:
        m = lw.max()
        e = np.exp(lw - m)
        out[2], out[3] = m, e.sum()
        out[4] = (e * np.where(isn & ignore_nan, 0.0, r)).sum()
    else:
        out[2] = -np.inf
    e_arg = -r / kT + (bias / kT if bias is not None else 0.0)
    out[5] = e_arg.max()
    out[6] = np.exp(e_arg - out[5]).sum()
    if bias is not None:
        out[7] = (bias / kT).max()
        out[8] = np.exp(bias / kT - out[7]).sum()
    else:
        out[7] = -np.inf
    return out


def finalize(stats, weighted, biased, kT=1.0):
    loss = stats[4] / stats[3] if weighted else stats[1] / stats[0]
    lse = stats[5] + np.log(stats[6])
    lse -= (stats[7] + np.log(stats[8])) if biased else np.log(stats[0])
    return loss, -kT * lse


def test_shard_rows_partition():
    for n, w in [(10, 3), (65536, 8), (7, 8), (0, 2)]:
        spans = [shard_rows(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def test_combine_stats_matches_whole_batch_oracle():
    g = gu.load('loss.npz')
    uB, ldj, lw, uA = (g[k].astype(np.float64) for k in ('uB', 'ldj', 'lw', 'uA'))
    N = uB.size
    for world in (1, 2, 3, 8):
        shards = [shard_stats(*(a[b:e] for a in (uB, ldj, uA, lw, lw)))
                  for b, e in (shard_rows(N, r, world) for r in range(world))]
        comb = combine_stats(torch.tensor(np.stack(shards))).numpy()
        loss, df = finalize(comb, weighted=True, biased=True)
        np.testing.assert_allclose(loss, oloss.boltzmann_kl_div_loss(uB, ldj, lw, uA), rtol=1e-6)
        np.testing.assert_allclose(df, oloss.fep_estimator(np.stack([uB - ldj - uA, lw], axis=1)), rtol=1e-6)
    # an empty shard (max = -inf) must not poison the combination
    shards.append(np.array([0, 0, -np.inf, 0, 0, -np.inf, 0, -np.inf, 0]))
    comb2 = combine_stats(torch.tensor(np.stack(shards))).numpy()
    np.testing.assert_allclose(comb2, comb, rtol=1e-12)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = gu.load('loss.npz')
        uB, ldj, lw, uA = (g[k].astype(np.float64) for k in ('uB', 'ldj', 'lw', 'uA'))
        b, e = shard_rows(uB.size, rank, world)
        local = torch.tensor(shard_stats(uB[b:e], ldj[b:e], uA[b:e], None, None))
        glob = allreduce_stats(local)                       # all_gather + combine (the RCCL path, on gloo)
        loss, df = finalize(glob.numpy(), weighted=False, biased=False)
        # gradient averaging of a replicated module (two small buckets)
        lin = torch.nn.Linear(5, 3)
        for i, p in enumerate(lin.parameters()):
            p.grad = torch.full_like(p, float(rank + 1 + i))
        allreduce_gradients(lin, bucket_bytes=40, average=True)
        gavg = [float(p.grad.flatten()[0]) for p in lin.parameters()]
        q.put((rank, float(loss), float(df), gavg))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_allreduce_of_tfep_statistics():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = gu.load('loss.npz')
    uB, ldj, uA = (g[k].astype(np.float64) for k in ('uB', 'ldj', 'uA'))
    for _, loss, df, gavg in res:                           # every rank holds the GLOBAL answer
        np.testing.assert_allclose(loss, oloss.boltzmann_kl_div_loss(uB, ldj, None, uA), rtol=1e-6)
        np.testing.assert_allclose(df, oloss.fep_estimator(uB - ldj - uA), rtol=1e-6)
        assert gavg == [1.5, 2.5]                           # mean over ranks of (rank + 1 + i)
    assert res[0][1:] == res[1][1:]


def _stats_cpu(uB, ldj, uA, lw, bias, kT=1.0, ignore_nan=False):
    """CPU stand-in for ops.tfep_reduce in the loss (the HIP reduction cannot run here): same 9 statistics."""
    def a(t):
        return None if t is None else t.detach().double().numpy()
    uB_, ldj_, uA_ = a(uB), a(ldj), a(uA)
    z = np.zeros_like(uB_)
    return torch.tensor(shard_stats(uB_, z if ldj_ is None else ldj_, z if uA_ is None else uA_, a(lw), a(bias),
                                    kT=kT, ignore_nan=ignore_nan))


def _loss_grad_worker(rank, world, port, q, weighted):
    """Distributed loss + gradient sync: every rank must end with the gradient of the GLOBAL-batch loss."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import tfep_amd.loss as tl
        tl.reduce_stats = _stats_cpu                         # this process only (the HIP reduction cannot run here)
        torch.manual_seed(0)
        n = 11                                               # ragged shards (6 + 5)
        x = torch.randn(n, 4, dtype=torch.float64)
        lw = torch.randn(n, dtype=torch.float64) if weighted else None
        lin = torch.nn.Linear(4, 2).double()                 # same replica on every rank (seeded)
        b, e = shard_rows(n, rank, world)
        out = lin(x[b:e])
        loss = tl.BoltzmannKLDivLoss(distributed=True)(out[:, 0], out[:, 1], None if lw is None else lw[b:e])
        loss.backward()
        allreduce_gradients(lin, bucket_bytes=40)            # default: SUM
        q.put((rank, float(loss), [p.grad.flatten().tolist() for p in lin.parameters()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('weighted', [False, True])
def test_two_rank_gloo_loss_and_gradient_sync_match_single_process(weighted):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_loss_grad_worker, args=(r, world, port, q, weighted)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, full batch, plain torch (reference loss.py:125-140)
    torch.manual_seed(0)
    n = 11
    x = torch.randn(n, 4, dtype=torch.float64)
    lw = torch.randn(n, dtype=torch.float64) if weighted else None
    lin = torch.nn.Linear(4, 2).double()
    out = lin(x)
    r = out[:, 0] - out[:, 1]
    loss = (torch.softmax(lw, 0) * r).sum() if weighted else r.mean()
    loss.backward()
    for _, l, grads in res:
        np.testing.assert_allclose(l, float(loss), rtol=1e-6)      # the statistics round r to float32
        for g, p in zip(grads, lin.parameters()):
            np.testing.assert_allclose(g, p.grad.flatten().numpy(), rtol=1e-10, atol=1e-14)


def _log_worker(rank, world, port, q, log_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import warnings
        from tfep_amd.io import TFEPLogger, gather_to_rank0
        n, batch = 7, 7                                        # 7 rows over 2 ranks: ragged shards (4 + 3)
        b, e = shard_rows(n, rank, world)
        shard = {'dataset_sample_index': torch.arange(b, e), 'potential': torch.arange(b, e, dtype=torch.float32) * 0.5 - 1.0}
        full = gather_to_rank0(shard)
        if rank == 0:
            loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.arange(n)), batch_size=batch)
            with warnings.catch_warnings():
                warnings.simplefilter('error')
                TFEPLogger(save_dir_path=log_dir, data_loader=loader).save_eval_tensors(full, step_idx=0)
        q.put((rank, None if full is None else {k: v.tolist() for k, v in full.items()}))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gather_of_per_sample_logs(tmp_path):
    """Per-sample potentials of a sharded batch reach rank 0 in row order and land in one TFEPLogger file."""
    from tfep_amd.io import TFEPLogger
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_log_worker, args=(r, world, port, q, str(tmp_path / 'log'))) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None
    assert res[0]['dataset_sample_index'] == list(range(7))
    assert res[0]['potential'] == [i * 0.5 - 1.0 for i in range(7)]
    back = TFEPLogger(save_dir_path=str(tmp_path / 'log')).read_eval_tensors(step_idx=0)
    assert back['dataset_sample_index'].tolist() == list(range(7)) and back['potential'].tolist() == res[0]['potential']


def _bench_shard_worker(rank, world, port, q):
    """bench.py's cfg3 sharding on gloo: every rank reduces ITS rows of the one global batch; the all-gathered
    statistics give every rank the estimate of the whole 64k-row batch (the numbers, not the GPU kernels, are under test)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import bench
        n = 65536
        row0, row1, glob = bench.rows_for_rank(n, rank, world, 'strong')
        work = np.random.default_rng(7).standard_normal(n)             # the same global batch on every rank
        z = np.zeros(row1 - row0)
        local = torch.tensor(shard_stats(work[row0:row1], z, z, None, None))
        _, df = finalize(allreduce_stats(local).numpy(), weighted=False, biased=False)
        q.put((rank, row0, row1, glob, float(df)))
    finally:
        dist.destroy_process_group()


def test_bench_strong_scaling_shards_one_global_batch():
    import bench
    assert bench.rows_for_rank(65536, 3, 8, 'strong') == (3 * 8192, 4 * 8192, 65536)      # BASELINE cfg3
    assert bench.rows_for_rank(65536, 0, 1, 'strong') == (0, 65536, 65536)                # N = 1: cfg2 unchanged
    assert bench.rows_for_rank(65536, 3, 8, 'weak') == (3 * 65536, 4 * 65536, 8 * 65536)
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2], r[3]) for r in res] == [(0, 32768, 65536), (32768, 65536, 65536)]
    work = np.random.default_rng(7).standard_normal(65536)
    ref = oloss.fep_estimator(work.astype(np.float32).astype(np.float64))
    for r in res:
        np.testing.assert_allclose(r[4], ref, rtol=1e-6)


def test_bench_spawns_its_own_ranks_when_run_without_a_launcher():
    """``python bench.py --gpus 2`` (the form the driver uses) with no WORLD_SIZE in the environment: the parent starts
    ``torch.distributed.run`` as a child, never touches a GPU itself, and hands rank 0's JSON line and the exit code
    through.  Rehearsed on gloo (TFEP_BENCH_BACKEND): identity flow, no kernels -- the launch, the rendezvous, the row
    sharding and the 9-scalar all-gather are what is under test."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['TFEP_BENCH_BACKEND'] = 'gloo'
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--features', '66', '--batch', '1000',
           '--steps', '2', '--warmup', '1']
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout                           # ONE line, from rank 0
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['collective_world_size'] == 2 and res['collective_backend'] == 'gloo'
    assert res['config']['global_batch'] == 1000 and res['config']['rows_per_gpu'] == 500 and res['scaling'] == 'strong'
    assert res['value'] is None and 'rehearsal' in res['data']          # never mistaken for a measurement
    assert len(res['roofline']['per_gpu_tflops']) == 2 and 'aggregate_frac' in res['roofline']
    # the estimate of the WHOLE batch from the two shards' statistics: -log mean exp(-(u_B - u_A)) with the bench's own streams
    work = []
    for rank in range(2):
        gen = torch.Generator().manual_seed(1234 + rank)
        torch.randn(500, 66, generator=gen)
        u_b, u_a = torch.randn(500, generator=gen), torch.randn(500, generator=gen)
        work.append((u_b - u_a).double())
    w = torch.cat(work)
    ref = -(torch.logsumexp(-w, 0) - np.log(1000.0))
    np.testing.assert_allclose(res['delta_f_estimate'], float(ref), rtol=1e-6)
    # a failing child's exit code reaches the caller (the parent relays, it does not swallow)
    env['TFEP_BENCH_BACKEND'] = 'no-such-backend'               # accepted by the parent, fails in every rank
    bad = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith('{')]


def _overlap_worker(rank, world, port, q):
    """Two stacked linears with gradients above / below the in-place threshold: ``allreduce_gradients`` and the hook-driven
    ``OverlappedGradientSync`` must both give the sum over ranks WITHOUT a staging copy of the large gradients."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(64, 48), torch.nn.Tanh(), torch.nn.Linear(48, 32))    # weights 12 KB / 6 KB, biases < 200 B
        x = torch.randn(10, 64, generator=torch.Generator().manual_seed(100 + rank))
        small = 4096                                        # bytes: both weight gradients count as "large" here
        copied = []
        real_cat = torch.cat
        torch.cat = lambda ts, *a, **k: (copied.extend(int(t.numel()) for t in ts), real_cat(ts, *a, **k))[1]
        try:
            net(x).square().sum().backward()
            ptrs = [p.grad.data_ptr() for p in net.parameters()]
            allreduce_gradients(net, small_bytes=small)
            g_plain = [p.grad.clone() for p in net.parameters()]
            assert ptrs == [p.grad.data_ptr() for p in net.parameters()]          # reduced where they lie
            for p in net.parameters():
                p.grad = None
            sync = OverlappedGradientSync(net, small_bytes=small)
            net(x).square().sum().backward()
            launched = sync.launched_in_backward                                 # queued by the hooks, before wait()
            sync.wait()
            g_hook = [p.grad.clone() for p in net.parameters()]
            sync.remove()
        finally:
            torch.cat = real_cat
        q.put((rank, [g.flatten().tolist() for g in g_plain], [g.flatten().tolist() for g in g_hook], launched, max(copied) if copied else 0))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gradient_sync_in_place_and_overlapped_with_the_backward():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process: the sum of the two ranks' gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 48), torch.nn.Tanh(), torch.nn.Linear(48, 32))
    total = None
    for rank in range(world):
        for p in net.parameters():
            p.grad = None
        net(torch.randn(10, 64, generator=torch.Generator().manual_seed(100 + rank))).square().sum().backward()
        g = [p.grad.flatten().clone() for p in net.parameters()]
        total = g if total is None else [a + b for a, b in zip(total, g)]
    for _, g_plain, g_hook, launched, biggest_copy in res:
        for a, b, t in zip(g_plain, g_hook, total):
            np.testing.assert_allclose(a, t.numpy(), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(b, t.numpy(), rtol=1e-5, atol=1e-6)
        assert launched == 2                                 # both weight gradients left during the backward
        assert biggest_copy <= 48                            # only the biases went through a flat buffer
