#!/usr/bin/env python
"""Per-phase shader cycles of the split fused kernel on the cfg2 layer (run with TFEP_DIAG=16)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, '.')
os.environ['TFEP_DIAG'] = os.environ.get('TFEP_DIAG', '16')
import bench  # noqa: E402
from tfep_amd import _lib  # noqa: E402

flow = bench.build_flow(3000, 1, 8, 'cuda')
x = torch.randn(65536, 3000, device='cuda').clamp_(-4.9, 4.9)
buf = (ctypes.c_ulonglong * 4)()
with torch.no_grad():
    flow(x)
    torch.cuda.synchronize()
    _lib.call('tfep_diag_split_cycles', buf)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    flow[0]._profile_events = []
    flow(x)
    torch.cuda.synchronize()
    _lib.call('tfep_diag_split_cycles', buf)
    ms = [a.elapsed_time(b) for a, b in flow[0]._profile_events]
print('fused launch ms', ms, 'cycles: loop %.3e epilogue %.3e workgroups %d' % (buf[0], buf[1], buf[2]))
print('busy real time per CU: %.1f ms (sum of workgroup lifetimes / 256 CUs); shader clock while busy %.2f GHz' % (buf[3] / 100e6 / 256 * 1e3, (buf[0] + buf[1]) / (buf[3] / 100e6) / 1e9))
print('per workgroup: loop %.0f epilogue %.0f cycles; epilogue share %.1f%%' % (buf[0] / buf[2], buf[1] / buf[2], 100.0 * buf[1] / (buf[0] + buf[1])))
