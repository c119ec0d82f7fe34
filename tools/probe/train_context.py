"""Why is the one-layer training step 242 ms inside bench.py and 221 ms on its own?  Replays bench's sequence piece by
piece on one layer.  (probe)"""
import sys
import time

import torch

sys.path.insert(0, '.')
from bench import build_flow                      # noqa: E402
from tfep_amd.loss import BoltzmannKLDivLoss      # noqa: E402

dev = torch.device('cuda', 0)
D = 3000
flow = build_flow(D, 4 if 'four' in sys.argv else 1, 8, dev)
layer = flow[0]
x = torch.randn(65536, D, device=dev).clamp_(-4.9, 4.9)
c = torch.rand(D, device=dev) * 0.3
x16 = x[:16384]
opt = torch.optim.SGD(layer.parameters(), lr=1e-7)


def train_step():
    for prm in layer.parameters():
        prm.grad = None
    yt, lt = layer(x16)
    BoltzmannKLDivLoss()((c * yt ** 2).sum(dim=1), lt).backward()
    opt.step()


def clock(fn, n=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


what = sys.argv[1:]
print('train alone', round(clock(train_step), 1), flush=True)
if 'forward' in what:
    with torch.no_grad():
        for _ in range(2):
            flow(x)
        torch.cuda.synchronize()
    print('after a 65536-row no-grad forward:', round(clock(train_step), 1), flush=True)
if 'exact' in what:
    for l in flow:
        l.split_gemm = False
    with torch.no_grad():
        flow(x)
        torch.cuda.synchronize()
    for l in flow:
        l.split_gemm = None
    print('after an exact-fp32 forward:', round(clock(train_step), 1), flush=True)
if 'inverse' in what:
    with torch.no_grad():
        y8, _ = layer(x[:8192])
        layer.inverse(y8)
        torch.cuda.synchronize()
    print('after an 8192-row inverse:', round(clock(train_step), 1), flush=True)
if 'invalidate' in what:
    layer._conditioner.invalidate_plan()
    torch.cuda.empty_cache()
    print('after invalidate_plan + empty_cache:', round(clock(train_step), 1), flush=True)
    print('again:', round(clock(train_step), 1), flush=True)
if 'sustained' in what:
    # the step time over a long run: the board's power controller averages over seconds
    train_step()
    torch.cuda.synchronize()
    for blk in range(8):
        t0 = time.perf_counter()
        for _ in range(5):
            train_step()
        torch.cuda.synchronize()
        print(f'steps {5 * blk + 1}-{5 * blk + 5}: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms', flush=True)
