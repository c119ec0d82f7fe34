#!/usr/bin/env python
"""Measure the fp32-MFMA ceiling of the attached MI355X with the GEMM's own instruction mix."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tfep_amd import _lib

blocks, iters = 256 * 4, 4000
scratch = torch.empty(blocks * 512, device='cuda')
s = _lib.stream_of(scratch)
_lib.call('tfep_diag_mfma_peak', _lib.ptr(scratch), blocks, 100, s)
torch.cuda.synchronize()
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call('tfep_diag_mfma_peak', _lib.ptr(scratch), blocks, iters, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    flops = blocks * 8 * iters * 200 * 2048.0
    print(f'register-only mfma_f32_16x16x4 loop: {ms:.1f} ms, {flops / ms / 1e9:.1f} TFLOP/s '
          f'({flops / ms / 1e9 / 157.3 * 100:.1f} % of 157.3)')
