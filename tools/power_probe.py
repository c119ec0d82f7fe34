#!/usr/bin/env python
"""Is the split GEMM power-limited?  Runs the hidden-layer-sized dense split GEMM in a loop on random operands and on
all-zero operands while sampling `rocm-smi --showclocks --showpower`, and prints clock / power / throughput for both."""
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, '.')
from tfep_amd import ops  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(['/opt/rocm/bin/rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True,
                                 timeout=10).stdout
            sclk = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', txt)
            pw = re.search(r'Power \(W\): ([\d.]+)', txt)
            if sclk and pw:
                out.append((int(sclk.group(1)), float(pw.group(1))))
        except Exception:
            pass
        time.sleep(0.3)


def run(label, a, w):
    tk = ops.tile_sizes()[2]
    B, K = a.shape
    N = w.shape[0]
    bias = torch.zeros(N, device='cuda')
    out = torch.empty(B, N, device='cuda')
    as_, ainv = ops.split_rows(a, K)
    ws_, winv = ops.split_rows(w, K, per_tensor=True)
    f = lambda: ops.masked_linear_split(as_, ainv, ws_, winv, bias, N, act=1, out=out)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples))
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 6.0:
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        n += 10
    dt = (time.perf_counter() - t0) / n
    stop.set()
    th.join()
    samples = samples[1:] or samples
    sclk = sum(s for s, _ in samples) / max(len(samples), 1)
    pw = sum(p for _, p in samples) / max(len(samples), 1)
    print(f'{label}: {dt * 1e3:.1f} ms per GEMM = {2.0 * B * K * N / dt / 1e12:.0f} TFLOP/s fp32-equivalent; '
          f'rocm-smi mean over {len(samples)} samples: sclk {sclk:.0f} MHz, socket power {pw:.0f} W')


if __name__ == '__main__':
    torch.manual_seed(0)
    B, K, N = 65536, 15008, 15104
    a = torch.randn(B, K, device='cuda')
    w = torch.randn(N, K, device='cuda') / K ** 0.5
    run('random operands', a, w)
    a.zero_()
    w.zero_()
    run('all-zero operands', a, w)
