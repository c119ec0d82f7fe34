#!/usr/bin/env python
"""One schedule of the blocked inverse of a cfg2 layer, N calls (for a kernel trace): SUPER=1 / 0."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

dev = torch.device('cuda')
D, B, N = int(os.environ.get('D', 3000)), int(os.environ.get('B', 8192)), int(os.environ.get('N', 4))
layer = bench.build_flow(D, 1, 8, dev)[0]
layer.inverse_super_kernel = os.environ.get('SUPER', '1') != '0'
x = torch.randn(B, D, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)).clamp_(-4.9, 4.9)
with torch.no_grad():
    y, _ = layer(x)
    layer.inverse(y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        layer.inverse(y)
    torch.cuda.synchronize()
if os.environ.get('TFEP_DIAG_INVERSE'):
    import ctypes
    from tfep_amd import _lib
    buf = (ctypes.c_ulonglong * 9)()
    _lib.call('tfep_diag_inverse_cycles', buf)
    with torch.no_grad():
        layer.inverse(y)
    torch.cuda.synchronize()
    _lib.call('tfep_diag_inverse_cycles', buf)
    names = ['products', 'block_init', 'wait_hidden', 'hidden_dots', 'wait_output', 'parameter_dot', 'transformer_inverse_and_stores', 'kernel']
    n = max(1, buf[8])
    print({k: round(buf[i] / n * 24 / 1e6, 3) for i, k in enumerate(names)}, 'M cycles per pair and inverse (24 launches);', int(buf[8]), 'pair-launches')
print('ms per inverse', 1e3 * (time.perf_counter() - t0) / N, 'schedule', layer.last_inverse_schedule, 'calls', N + 1)
if os.environ.get('TFEP_DIAG') == '16':
    import ctypes
    from tfep_amd import _lib
    buf = (ctypes.c_ulonglong * 4)()
    _lib.call('tfep_diag_split_cycles', buf)
    with torch.no_grad():
        layer.inverse(y)
    torch.cuda.synchronize()
    _lib.call('tfep_diag_split_cycles', buf)
    print(f'split GEMMs of one inverse: {buf[2]} workgroups, k-loops {buf[0] / 256 / 1.0e6:.2f} M cycles per CU, epilogues {buf[1] / 256 / 1.0e6:.2f} M, '
          f'lifetimes {buf[3] / 256 / 100 / 1e3:.2f} ms per CU -> clock {(buf[0] + buf[1]) / max(1, buf[3]) * 100:.0f} MHz')
